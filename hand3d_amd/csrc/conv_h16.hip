// conv_h16.hip -- 3x3 (round 4: also 7x7 and 1x1) / stride-1 / SAME convolution in half precision (f16 operands, float32 accumulate) for the
// half-precision trunks of BASELINE config 5: bias + leaky-ReLU (+ 2x2 max-pool) fused, halves out.
//
// Same call sites as conv_mfma.hip's F16 instantiations (NetworkOps.conv_relu + max_pool, utils/general.py:36-65; trunk layer
// lists nets/ColorHandPose3DNetwork.py:144-157,183-199) and the same arithmetic (v_mfma_f32_32x32x16_f16 on the same packed
// weights), but built for the half-precision matrix pipe, which consumes operands 16x faster per FLOP than the f32 one.
// What the ablations of the general kernel showed (profiles/r02_tuning_notes.md): at 16 MFMAs x 32 cycles per K step its weight
// LDS-DMA pieces, fragment reads and per-step barrier cost more than the MFMAs.  Hence:
//   * 4 x NT register tiles per wave: 128 pixels x (32 NT) couts; workgroup = 2 x 2 waves = 16 x 16 output pixels x (64 NT)
//     couts; NT = 4 (Cout % 256 == 0 and Cin >= 512) fills all 256 AGPRs: ONE wave per SIMD, one workgroup per CU;
//   * weights go global -> VGPR straight in MFMA fragment order (the f16 blob's 1-KB pieces), ring of RING K-steps,
//     scalar offset per (tap, K-step): no LDS, no DMA issue slots, no barrier for B;
//   * the 18 x 18 x 64-half input patch (with halo) sits in LDS; the nine taps x four K-steps of a chunk are one straight-line
//     block (every fragment address is base + immediate, fragments prefetched one step ahead): ONE barrier per 64-channel
//     chunk (576 MFMAs per wave at NT = 4) instead of one per 16; the lane -> patch-pixel map makes every ds_read_b128
//     conflict-free for the hardware's real 16-lane groups;
//   * NT = 4: the patch is DOUBLE buffered (the next chunk streams in under the current one in three register groups);
//     NT <= 2: single buffer and two (NT = 1: three) workgroups per CU -- with 1-4 chunks per item a lone workgroup's load and
//     store phases are serial latency chains that only another workgroup's MFMAs can cover;
//   * MFMA issued as D[cout][pixel] (weights = A operand): a lane holds four consecutive couts of a pixel, the epilogue rounds
//     to half, transposes through per-wave LDS slabs and stores 16 bytes per lane with a pixel's couts contiguous (the pooled
//     form takes the max of the four pixels' half vectors on the way);
//   * persistent grid over (image, tile, cout block) items; edges and ragged tiles are out-of-range buffer offsets.
//   * round 4: the filter size is a template parameter KS -- PoseNet2D's 7x7 score-map stages (nets/ColorHandPose3DNetwork.py:206-215) and the 1x1
//     layers with >= 64 couts take the single-buffer forms with a patch of (16 + KS - 1)^2 pixels and 4 KS^2 K-steps per chunk.
//   * round 6: the fused first block (conv1_1 + conv1_2 + pool) has a second form, conv_h16_first_kernel below: one workgroup per CU, conv1_2's whole
//     filter resident in registers, the next tile's patch built between the MFMAs of the current one (3.89 -> 3.21 ms at the config-5 shape).
// Measured (profiles/r06_h16_counters.md): 0.43-0.48 of the 2.5 PF dense peak per 3x3 instantiation at the config-5 shape (the 7x7 form: 0.53); the
// matrix pipe is busy 0.67-0.73 of the time at the 1.5-1.6 GHz the chip sustains under this load (a register-only loop of the same MFMA: 2.05 GHz).
#include "hp3d_common.h"
#include <algorithm>

namespace {

constexpr int HT = 16;                    // output tile: 16 x 16 pixels
constexpr int HPW = HT + 2;               // patch width / height (halo 1) -- these are the 3x3 values; the kernel derives its own from KS
constexpr int HPITCH = 36;                // floats per patch pixel: 64 halves = 32 floats + 4 pad (144 B: conflict-free b128)
constexpr int HPATCH_FLOATS = HPW * HPW * HPITCH;        // 11664 floats = 46.7 KB per buffer
constexpr int HPGS = 4;                    // patch pieces per group (the kernel derives the group count HPG from its filter size)
#ifndef HP3D_H16_RING
#define HP3D_H16_RING 6
#endif
#ifndef HP3D_H16_RING2
#define HP3D_H16_RING2 6
#endif
#ifndef HP3D_H16_RING1
#define HP3D_H16_RING1 6
#endif
// K-steps of weight fragments in flight per wave, by cout blocks per wave NT (a step is 4 NT MFMAs x 32 cycles: the fewer
// blocks, the shorter the time a ring of a given depth covers)
constexpr int h16_ring(int nt) { return nt == 4 ? HP3D_H16_RING : nt == 2 ? HP3D_H16_RING2 : HP3D_H16_RING1; }

// WPS = workgroups per CU (waves per SIMD).  WPS == 1: the patch is double buffered (the next chunk streams in under the
// current one).  WPS > 1 (NT <= 2: few accumulators, few chunks, store-heavy): ONE patch buffer, and the load / store phases
// of one workgroup are covered by the MFMAs of the other(s) on the same CU.
// FUSE (NT = 1, POOL: the first block of both trunks, nets/ColorHandPose3DNetwork.py:144-145,183-184): `in` is the raw
// float32 image and the 18 x 18 x 64 patch of conv1_1's output is COMPUTED into LDS instead of loaded -- 11 row blocks of 32
// patch pixels, K = 27 + two bias rows in two v_mfma_f32_32x32x16_f16 steps with the operand order, rounding points and
// accumulation order of conv_first_kernel<true> (bit-identical halves), leaky-ReLU, zero outside the image (conv1_2's
// SAME padding pads conv1_1's OUTPUT) -- so that conv1_1's 64-channel activation (5 GB at 128 x 480 x 640) never goes to HBM.
// KS (round 4): the filter size -- 3 (the trunk layers), 7 (PoseNet2D's score-map stages, nets/ColorHandPose3DNetwork.py:206-215) or 1 (the
// 1x1 layers in front of the score-map heads, :160, :203, :213).  Only the patch extent (16 + KS - 1)^2, its halo and the number of taps
// change; 7x7 / 1x1 run the single-buffer form (NT <= 2), the conflict-free lane map of the A fragments is the 3x3 one (18-pixel rows).
template <int NT, bool POOL, int WPS, bool FUSE = false, int KS = 3>
HP3D_KERNEL2(256, WPS)
void conv_h16_kernel(const ConvParams p) {
    constexpr bool DB = WPS == 1 && KS == 3;
    constexpr int HRING = h16_ring(NT);
    static_assert(!FUSE || (NT == 1 && POOL && !DB && KS == 3), "fused first block: 64 -> 64 couts, pooled, single patch buffer");
    static_assert(KS == 1 || KS == 3 || KS == 7, "");
    static_assert(KS == 3 || (!POOL && NT <= 2), "7x7 / 1x1: plain epilogue, single patch buffer");
    // patch geometry of this filter size (the names shadow the 3x3 constants above on purpose)
    constexpr int HPW = HT + KS - 1;                         // patch width / height
    constexpr int HPATCH_FLOATS = HPW * HPW * HPITCH;
    constexpr int HPIECES = HPW * HPW * 8;
    constexpr int HPVEC = (HPIECES + 255) / 256;             // 16-byte pieces per thread: 8 (1x1), 11 (3x3), 16 (7x7)
    constexpr int HPG = (HPVEC + HPGS - 1) / HPGS;
    constexpr int NSTEP = 4 * KS * KS;                       // K-steps of a 64-half chunk: 4 per tap
    constexpr int HPSTEP = NSTEP / HPG;
    constexpr int PAD = (KS - 1) / 2;
    HP3D_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = HP3D_READFIRSTLANE(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;             // 2 x 2 waves: pixel half x cout half
    const int li = lane & 31, lh = lane >> 5;
    constexpr int BN = 2 * NT * 32;                      // couts per workgroup
    const int CO32 = p.Cout >> 5;
    const int nchunks = p.Cin >> 5;                      // p.Cin counts 4-byte units (f16 pairs): 32 units = 64 halves
    const int KB = nchunks * 4;                          // K-steps (16 halves) per tap
    const int tiles_x = (p.Wo + HT - 1) / HT, tiles_y = (p.Ho + HT - 1) / HT;
    const int ncb = p.Cout / BN;
    const int nitems = p.B * tiles_y * tiles_x * ncb;
    const int Hs = POOL ? (p.Ho >> 1) : p.Ho, Ws = POOL ? (p.Wo >> 1) : p.Wo;

    // A fragment rows of this lane: M-block mt of pixel half wm covers output rows 8 wm + 2 mt, +1 (32 pixels = 2 rows x 16).
    // MFMA column li -> pixel (dy, x) of the block.  ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27},
    // {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS table) -- and a group is conflict-free when its 16
    // lanes' 16-byte pieces fall into 16 different 4-bank columns: with the 144-byte pixel pitch that is "16 different patch
    // pixel indices mod 16".  Rows are 18 pixels apart, so the lanes of a group take x = 0..7 of row 0 with x = 6..13 of row 1,
    // or x = 8..15 of row 0 with x = 14, 15, 0..5 of row 1 (a plain 16-lanes-per-row map would be 2-way conflicted: 8 LDS
    // cycles per read instead of 4).  Tap and K-step offsets shift all lanes alike and keep the property.
    const int pdy = li >> 4;
    const int pj = li & 15;
    const int pdx = KS != 3 ? pj
                  : pdy == 0 ? (pj < 4 ? pj : pj < 12 ? pj + 4 : pj - 8)
                             : (pj < 2 ? pj + 14 : pj < 4 ? pj - 2 : pj < 12 ? pj + 2 : pj - 10);
    int abase[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int ly = (wm * 4 + mt) * 2 + pdy;
        abase[mt] = ((ly * HPW + pdx) * HPITCH + lh * 4) * 4;         // bytes
    }
    // fused first block: conv1_1's 32 x 64 filter matrix (K slot q = 8 j + i of step j is k = 16 j + 8 kh + i; k = (r*3+s)*3+c;
    // k = 27, 28 carry the float32 bias as hi + lo halves) in 16 registers for the whole kernel, as in conv_first.hip
    f32x4 bwh[2][2];
    if (FUSE) {
        const int m = li, kh = lh;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const float bias = p.bias1[nb * 32 + m], bias_hi = (float)(hp3d_f16)bias;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f16x8 hh;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = 16 * j + 8 * kh + i;
                    const float wv = k < 27 ? p.wpk1[(((k >> 3) * 2 + nb) * 2 + ((k >> 2) & 1)) * 128 + m * 4 + (k & 3)] : 0.f;
                    hh[i] = (hp3d_f16)(k < 27 ? wv : k == 27 ? bias_hi : k == 28 ? bias - bias_hi : 0.f);
                }
                bwh[nb][j] = __builtin_bit_cast(f32x4, hh);
            }
        }
    }
    // ... and where operand slot q sits in the staged image window relative to the pixel's (row - 1, col - 1) corner; the two
    // bias rows read the constant 1.0 and the K tail the constant 0.0 parked behind the window (negative = absolute index)
    int fkoff[16];
    if (FUSE) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int k = 16 * (q >> 3) + 8 * lh + (q & 7);
            const int r = k / 9, sx = (k / 3) % 3, c = k % 3;
            fkoff[q] = k < 27 ? (r * (HPW + 2) + sx) * 3 + c : k < 29 ? -((HPW + 2) * (HPW + 2) * 3) : -((HPW + 2) * (HPW + 2) * 3 + 1);
        }
    }
    const int wbase = (tid >> 3) * HPITCH + (tid & 7) * 4;     // LDS float index of this thread's patch piece 0 (piece v: + 32 v pixels)
    const hp3d_rsrc_t wrsrc = HP3D_MAKE_RSRC(p.wpk, (unsigned)(KS * KS * p.Cin) * (unsigned)p.Cout * 4u);
    const int tap_stride_b = KB * CO32 * 1024;           // bytes between taps of the packed weights

    for (int item = blockIdx.x; item < nitems; item += (int)gridDim.x) {
        // XCD-affine item order (round 6, as conv_wino4's item_decode): workgroup ids go round-robin over the 8 XCDs, so with the cout block
        // innermost the ncb items that read the SAME input patch landed on ncb different XCDs and the patch crossed the fabric once per cout
        // block (counter traffic 1.65x the algorithmic bytes, profiles/r04_h16_counters.md).  Now consecutive items of one XCD are the cout
        // blocks of one tile: the patch is fetched once per XCD and re-read from that XCD's L2.  (Whole groups of eight tiles in that
        // order; the up to seven tiles left over behind them, still cout block innermost.)
        int it, cb;
        {
            const int ntile = nitems / ncb, aff = (ntile >> 3) * 8 * ncb;
            if (item < aff) {
                const int xcd = item & 7, j = item >> 3, tq = j / ncb;
                cb = j - tq * ncb;
                it = tq * 8 + xcd;
            } else {
                const int q = item - aff, ti = q / ncb;
                cb = q - ti * ncb;
                it = (ntile & ~7) + ti;
            }
        }
        const int tx = it % tiles_x; it /= tiles_x;
        const int ty = it % tiles_y;
        [[maybe_unused]] const int b = it / tiles_y;          // (the host pass sees the buffer-descriptor stand-ins only)
        const int oy0 = ty * HT, ox0 = tx * HT;
        // this thread's patch pieces: piece idx -> (patch pixel, 16-byte slot) -> byte offset inside image b; SAME padding,
        // the image edge and idx past the patch get an out-of-range offset, which a buffer load answers with zeros.
        // Pieces move in HPG groups of HPGS (fetch -> registers -> LDS) so that only HPGS x 4 registers hold patch data.
        constexpr int OOR = (int)0x80000000;
        const hp3d_rsrc_t irsrc = FUSE ? HP3D_MAKE_RSRC(p.in + (size_t)b * p.H * p.W * 3, (unsigned)(p.H * p.W) * 12u)
                                       : HP3D_MAKE_RSRC(p.in + (size_t)b * p.H * p.W * p.in_cs, (unsigned)(p.H * p.W) * (unsigned)p.in_cs * 4u);
        int poff[HPVEC];
#pragma unroll
        for (int v = 0; v < HPVEC; ++v) {
            const int idx = tid + v * 256;
            const int pix = idx >> 3, c4 = idx & 7;
            const int py = pix / HPW, px = pix - py * HPW;
            const int gy = oy0 - PAD + py, gx = ox0 - PAD + px;
            const bool ok = idx < HPIECES && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            poff[v] = ok ? ((gy * p.W + gx) * p.in_cs + c4 * 4) * 4 : OOR;
        }
        f32x4 preg[HPGS];
        auto patch_fetch = [&](int chunk, int g) {
#pragma unroll
            for (int j = 0; j < HPGS; ++j)
                if (g * HPGS + j < HPVEC) preg[j] = HP3D_BUFFER_LOAD16(irsrc, poff[g * HPGS + j], chunk * 128);
        };
        auto patch_commit = [&](int buf, int g) {
#pragma unroll
            for (int j = 0; j < HPGS; ++j) {
                const int v = g * HPGS + j;                // piece tid + 256 v: patch pixel (tid >> 3) + 32 v, slot tid & 7
                if (v < HPVEC && tid + v * 256 < HPIECES) *(f32x4*)(smem + buf * HPATCH_FLOATS + wbase + v * (32 * HPITCH)) = preg[j];
            }
        };

        f32x16 acc[4][NT];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

        // weight fragments: global -> VGPR, lane-linear 1-KB pieces [tap][K-step][co32]; this wave's couts = wn half of the block
        const int wvoff = (cb * (BN / 32) + wn * NT) * 1024 + lane * 16;
        f32x4 fb[HRING][NT];
        auto b_fetch = [&](int slot, int tap, int kb) {
            const int soff = tap * tap_stride_b + kb * (CO32 * 1024);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fb[slot][nt] = HP3D_BUFFER_LOAD16(wrsrc, wvoff + nt * 1024, soff);
        };
        f32x4 fa[2][4];
        auto a_fetch = [&](int set, int buf, int toff_b, int ks) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                fa[set][mt] = *(const f32x4*)((const char*)smem + buf * (HPATCH_FLOATS * 4) + abase[mt] + toff_b + ks * 32);
        };

        // a whole chunk into buffer 0 with all pieces in flight at once (the fragment registers are free at that point), and
        // the first RING-1 K-steps of its weights: (tap, ks) = (s / 4, s % 4)
        auto load_chunk = [&](int chunk) {
            constexpr int GRP = HPVEC <= 11 ? HPVEC : 8;          // pieces in flight at once (7x7: 16 pieces in two rounds: registers)
#pragma unroll
            for (int v0 = 0; v0 < HPVEC; v0 += GRP) {
                f32x4 first[GRP];
#pragma unroll
                for (int v = 0; v < GRP; ++v)
                    if (v0 + v < HPVEC) first[v] = HP3D_BUFFER_LOAD16(irsrc, poff[v0 + v], chunk * 128);
                if (v0 == 0) {
#pragma unroll
                    for (int s = 0; s < HRING - 1; ++s) b_fetch(s, s >> 2, chunk * 4 + (s & 3));
                }
#pragma unroll
                for (int v = 0; v < GRP; ++v)
                    if (v0 + v < HPVEC && tid + (v0 + v) * 256 < HPIECES) *(f32x4*)(smem + wbase + (v0 + v) * (32 * HPITCH)) = first[v];
            }
        };
        // fused first block: the patch = conv1_1 over the image.  The 20 x 20 x 3 float32 image window (zero outside the image)
        // is staged in LDS behind the patch buffer, followed by the constants 1.0 (bias rows) and 0.0 (K tail); then, row block by
        // row block of 32 patch pixels (wave w: blocks w, w + 4, w + 8), lane (pixel, K half) reads its 16 operands at
        // window[pixel base + fkoff[q]] -- fkoff = (r * 20 + s) * 3 + c of k = (r*3+s)*3+c, set once per kernel
        constexpr int FWIN = (HPW + 2) * (HPW + 2) * 3;          // 1200 floats
        float* fwin = smem + HPATCH_FLOATS;
        auto build_patch = [&]() {
#pragma unroll
            for (int s = 0; s < HRING - 1; ++s) b_fetch(s, s >> 2, s & 3);
#pragma unroll
            for (int v = 0; v < (FWIN + 2 + 255) / 256; ++v) {
                const int f = tid + v * 256;
                const int row = f / ((HPW + 2) * 3), rem = f - row * ((HPW + 2) * 3), col = rem / 3, c = rem - col * 3;
                const int yy = oy0 - 2 + row, xx = ox0 - 2 + col;
                const bool ok = f < FWIN && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
                const float val = HP3D_BUFFER_LOAD4(irsrc, ok ? ((yy * p.W + xx) * 3 + c) * 4 : OOR, 0);
                if (f < FWIN + 2) fwin[f] = f == FWIN ? 1.0f : val;           // [FWIN] = 1.0, [FWIN + 1] = 0.0
            }
            __syncthreads();
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int blk = wave; blk < (HPW * HPW + 31) / 32; blk += 4) {
                const int pp = blk * 32 + li;                  // patch pixel of this lane
                const int ppc = pp < HPW * HPW ? pp : 0;
                const int py = ppc / HPW, px = ppc - py * HPW;
                const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;               // conv1_1 output position
                const bool inside = pp < HPW * HPW && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                const float* wb = fwin + (py * (HPW + 2) + px) * 3;           // window row py-1+1.., i.e. image (gy - 1, gx - 1)
                f32x4 ah[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f16x8 hh;
#pragma unroll
                    for (int i = 0; i < 8; ++i) hh[i] = (hp3d_f16)(fkoff[8 * j + i] >= 0 ? wb[fkoff[8 * j + i]] : fwin[-fkoff[8 * j + i]]);
                    ah[j] = __builtin_bit_cast(f32x4, hh);
                }
                f32x16 c0 = HP3D_MFMA_32x32x16_F16(bwh[0][0], ah[0], zero);
                f32x16 c1 = HP3D_MFMA_32x32x16_F16(bwh[1][0], ah[0], zero);
                c0 = HP3D_MFMA_32x32x16_F16(bwh[0][1], ah[1], c0);
                c1 = HP3D_MFMA_32x32x16_F16(bwh[1][1], ah[1], c1);
                // register 4a + e of half nb = cout 32 nb + 8 a + 4 lh + e of pixel pp: 8-byte pieces of the 144-byte patch row
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    f16x4 h0, h1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v0 = c0[4 * a + e], v1 = c1[4 * a + e];
                        v0 = fmaxf(v0, HP3D_LEAKY_SLOPE * v0); v1 = fmaxf(v1, HP3D_LEAKY_SLOPE * v1);
                        h0[e] = inside ? (hp3d_f16)v0 : (hp3d_f16)0.f; h1[e] = inside ? (hp3d_f16)v1 : (hp3d_f16)0.f;
                    }
                    if (pp < HPW * HPW) {
                        *(f16x4*)((char*)smem + pp * (HPITCH * 4) + (8 * a + 4 * lh) * 2) = h0;
                        *(f16x4*)((char*)smem + pp * (HPITCH * 4) + (32 + 8 * a + 4 * lh) * 2) = h1;
                    }
                }
            }
        };
        __syncthreads();                       // the previous item's waves are done with the patch buffers / slabs
        if constexpr (FUSE) build_patch(); else load_chunk(0);
        __syncthreads();
        int cur = 0;
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const bool has_next = DB && chunk + 1 < nchunks;
            if (!DB && chunk > 0) { __syncthreads(); load_chunk(chunk); __syncthreads(); }
            a_fetch(0, cur, 0, 0);
            // the 4 KS^2 K-steps of this chunk (36 for 3x3), straight line: step s = 4 tap + ks
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                const int tap = s >> 2, ks = s & 3;
                HP3D_SCHED_BARRIER();
                // next chunk's patch, group g: fetched at step g HPSTEP, committed to the other buffer HPSTEP - 1 steps later
                if (has_next) {
                    if (s % HPSTEP == 0) patch_fetch(chunk + 1, s / HPSTEP);
                    if (s % HPSTEP == HPSTEP - 1) patch_commit(cur ^ 1, s / HPSTEP);
                }
                // prefetch: A fragments of step s+1 (other register set), weight fragments of step s + RING - 1
                if (s + 1 < NSTEP) {
                    const int t1 = (s + 1) >> 2, r1 = t1 / KS, c1 = t1 - r1 * KS;
                    a_fetch((s + 1) & 1, cur, (r1 * HPW + c1) * HPITCH * 4, (s + 1) & 3);
                }
                {
                    const int s2 = s + HRING - 1;
                    if (s2 < NSTEP) b_fetch(s2 % HRING, s2 >> 2, chunk * 4 + (s2 & 3));
                    else if (has_next) b_fetch(s2 % HRING, (s2 - NSTEP) >> 2, (chunk + 1) * 4 + ((s2 - NSTEP) & 3));
                }
                HP3D_SCHED_BARRIER();          // the prefetches are issued BEFORE this step's MFMAs: a whole step of latency cover
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = HP3D_MFMA_32x32x16_F16(fb[s % HRING][nt], fa[s & 1][mt], acc[mt][nt]);   // D[cout][pixel]
                (void)tap; (void)ks;
            }
            HP3D_SCHED_BARRIER();
            if (DB) { __syncthreads(); cur ^= 1; }     // patch[cur ^ 1] complete, patch[cur] free
        }
        if (!DB) __syncthreads();              // all waves are done with the patch: the slabs below reuse its memory

        // ---- epilogue: bias + leaky-ReLU (+ 2x2 max-pool), halves out.  Accumulator (lane (li, lh), register 4 a + e) = pixel li
        //      of row block mt, cout 32 nt + 8 a + 4 lh + e: four consecutive couts per lane.  Each wave transposes through its own
        //      LDS slab (the patch buffers are free after the last chunk's barrier): 8-byte writes [pixel][cout], 16-byte reads, and
        //      every global store is 16 B per lane with one pixel's couts contiguous (the pooled form takes the max of the four
        //      pixels' half vectors on the way: rounding is monotonic, so max-then-round == round-then-max).
        //      Buffer stores with 32-bit offsets inside the image; an invalid lane gets an out-of-range offset and is dropped.
        const hp3d_rsrc_t orsrc = HP3D_MAKE_RSRC((hp3d_f16*)p.out + (size_t)b * Hs * Ws * p.out_cs,
                                                 (unsigned)(Hs * Ws) * (unsigned)p.out_cs * 2u);
        const int px_b = p.out_cs * 2;                                       // bytes per output pixel
        constexpr int G = 4 * NT;                                            // 16-byte groups per pixel of this wave's couts
        constexpr int SPITCH = 64 * NT + 16;                                 // slab bytes per pixel (pad: conflict-free)
        constexpr int SLAB = 32 * SPITCH;                                    // one row block (32 pixels)
        char* slab = (char*)smem + wave * (4 * SLAB);                        // this wave's four slabs (one per row block)
        f32x4 bias4[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int a = 0; a < 4; ++a)
                bias4[nt][a] = *(const f32x4*)(p.bias + cb * BN + (wn * NT + nt) * 32 + a * 8 + lh * 4);
        // slab pixel of MFMA column li (the lane -> pixel map of abase): sp = 16 dy + x
        const int sp_w = pdy * 16 + pdx;
        const int wr_off = sp_w * SPITCH + lh * 8;
        const int g = lane % G, sp0 = lane / G;                              // read side: lane -> (pixel sp0 + k PPR, group g)
        const int cbyte = (cb * BN + wn * (32 * NT)) * 2 + g * 16;
        const bool cok = cb * BN + wn * (32 * NT) + g * 8 < p.cout_store;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            char* sl = slab + mt * SLAB;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    f16x4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[mt][nt][a * 4 + e] + bias4[nt][a][e];
                        if (p.act) x = fmaxf(x, HP3D_LEAKY_SLOPE * x);
                        h[e] = (hp3d_f16)x;
                    }
                    *(f16x4*)(sl + wr_off + nt * 64 + a * 16) = h;
                }
        }
        HP3D_WAVE_LDS_SYNC();                  // one rendezvous for all four row blocks
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const char* sl = slab + mt * SLAB;
            const int yb = oy0 + (wm * 4 + mt) * 2;                          // conv-output row of slab row 0
            if (POOL) {
                constexpr int PPR = 64 / G;                                  // pooled pixels per round (16 when NT = 1: 8 are real)
                constexpr int ROUNDS = (8 + PPR - 1) / PPR;
#pragma unroll
                for (int r = 0; r < ROUNDS; ++r) {
                    const int pp = sp0 + r * PPR;                            // pooled column inside the tile, 0..7
                    const int ppc = pp < 8 ? pp : 0;
                    f16x8 m = *(const f16x8*)(sl + (2 * ppc) * SPITCH + g * 16);
#pragma unroll
                    for (int w = 1; w < 4; ++w) {
                        const f16x8 o = *(const f16x8*)(sl + ((w >> 1) * 16 + 2 * ppc + (w & 1)) * SPITCH + g * 16);
#pragma unroll
                        for (int e = 0; e < 8; ++e) m[e] = m[e] > o[e] ? m[e] : o[e];
                    }
                    const int yp = yb >> 1, xp = (ox0 >> 1) + pp;
                    const int off = (cok && pp < 8 && yp < Hs && xp < Ws) ? (yp * Ws + xp) * px_b + cbyte : OOR;
                    HP3D_BUFFER_STORE16(orsrc, __builtin_bit_cast(f32x4, m), off, 0);
                }
            } else {
                constexpr int PPR = 64 / G;                                  // pixels per round: 4 / 8 / 16
#pragma unroll
                for (int r = 0; r < 32 / PPR; ++r) {
                    const int sp = sp0 + r * PPR;
                    const f32x4 v = *(const f32x4*)(sl + sp * SPITCH + g * 16);
                    const int y = yb + ((r * PPR) >> 4), x = ox0 + sp0 + ((r * PPR) & 15);
                    const int off = (cok && y < Hs && x < Ws) ? (y * Ws + x) * px_b + cbyte : OOR;
                    HP3D_BUFFER_STORE16(orsrc, v, off, 0);
                }
            }
        }
    }
}

// ---- the fused first block with conv1_2's FILTERS RESIDENT IN REGISTERS (round 6) ----------------------------------------------------------
// conv_h16_kernel<1, true, 2, true> above measured 0.31 of the dense peak (matrix pipe busy 0.40, profiles/r06_h16_counters.md): with one cout
// block per wave every A fragment read from LDS feeds ONE MFMA -- four waves x 1 KB per 32-cycle MFMA = the CU's whole 128 B/clk of LDS -- and
// the filter ring asks the L1 for another 32 B/clk on top, while the layer's whole filter is 73.7 KB and the SAME for every one of the
// 153,600 items of the config-5 shape.  Here:
//   * one workgroup per CU, one wave per SIMD; a wave owns 4 output rows x 16 pixels x ALL 64 couts (2 row blocks x 2 cout blocks): an A
//     fragment feeds two MFMAs (LDS at half its rate), and the wave's 72 filter fragments (36 K-steps x 2 cout blocks, 288 registers) are
//     loaded ONCE per kernel -- no filter traffic at all inside the item loop;
//   * the patch is double buffered and the NEXT item's patch (conv1_1 over the staged image window, the same MFMA form, operand order and
//     rounding points as above) is built in slices between the K-steps of the current item; the window after that is fetched at the top of the
//     loop and staged at its end: ONE workgroup barrier per item;
//   * epilogue as above (bias, leaky-ReLU, halves, 2x2 max-pool through per-wave LDS slabs); a wave's slabs are the patch rows it builds
//     itself, so no second barrier is needed.
// Same results bit for bit as the form above (same products, same K order per accumulator).
HP3D_KERNEL2(256, 1)
void conv_h16_first_kernel(const ConvParams p) {
    constexpr int FW = HPW + 2;                              // staged image window: 20 x 20 x 3 float32
    constexpr int FWIN = FW * FW * 3;                        // 1200 floats; [FWIN] = 1.0 (bias rows), [FWIN + 1] = 0.0 (K tail)
    constexpr int FWIN_PITCH = FWIN + 4;
    constexpr int FWV = (FWIN + 255) / 256;                  // window floats per thread
    constexpr int NBLK = (HPW * HPW + 31) / 32;              // 11 row blocks of 32 patch pixels
    constexpr int OOR = (int)0x80000000;
    constexpr int WRES_AGPR_STEPS = 32;                      // filter fragments of K-steps 0..31 live in the 256 AGPRs, the last four steps' in VGPRs
    HP3D_DYN_SMEM(smem);
    float* const fwin0 = smem + 2 * HPATCH_FLOATS;
    float* const bias_l = fwin0 + 2 * FWIN_PITCH;            // conv1_2's 64 biases
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = HP3D_READFIRSTLANE(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int tiles_x = (p.Wo + HT - 1) / HT, tiles_y = (p.Ho + HT - 1) / HT;
    const int nitems = p.B * tiles_y * tiles_x;
    const int Hs = p.Ho >> 1, Ws = p.Wo >> 1;
    // lane -> patch pixel of an A fragment: the conflict-free map of conv_h16_kernel (KS = 3)
    const int pdy = li >> 4, pj = li & 15;
    const int pdx = pdy == 0 ? (pj < 4 ? pj : pj < 12 ? pj + 4 : pj - 8) : (pj < 2 ? pj + 14 : pj < 4 ? pj - 2 : pj < 12 ? pj + 2 : pj - 10);
    int abase[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) abase[mt] = ((((wave * 2 + mt) * 2 + pdy) * HPW + pdx) * HPITCH + lh * 4) * 4;
    // conv1_1's filter matrix and operand slots, as in conv_h16_kernel<.., FUSE = true>
    f32x4 bwh[2][2];
    {
        const int m = li, kh = lh;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const float bias = p.bias1[nb * 32 + m], bias_hi = (float)(hp3d_f16)bias;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f16x8 hh;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = 16 * j + 8 * kh + i;
                    const float wv = k < 27 ? p.wpk1[(((k >> 3) * 2 + nb) * 2 + ((k >> 2) & 1)) * 128 + m * 4 + (k & 3)] : 0.f;
                    hh[i] = (hp3d_f16)(k < 27 ? wv : k == 27 ? bias_hi : k == 28 ? bias - bias_hi : 0.f);
                }
                bwh[nb][j] = __builtin_bit_cast(f32x4, hh);
            }
        }
    }
    int fkoff[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int k = 16 * (q >> 3) + 8 * lh + (q & 7);
        const int r = k / 9, sx = (k / 3) % 3, c = k % 3;
        fkoff[q] = k < 27 ? (r * FW + sx) * 3 + c : k < 29 ? FWIN : FWIN + 1;      // float index relative to the pixel's window corner / absolute
    }
    // conv1_2's filters: fragment (K-step s = 4 tap + ks, cout block nt) is the lane-linear 1-KB piece (2 s + nt) of the packed blob
    const hp3d_rsrc_t wrsrc = HP3D_MAKE_RSRC(p.wpk, 9u * 64u * 32u * 4u);
    f32x4 wres[36][2];
#pragma unroll
    for (int s = 0; s < 36; ++s)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) wres[s][nt] = HP3D_BUFFER_LOAD16(wrsrc, lane * 16, (s * 2 + nt) * 1024);
    // this thread's floats of a window: f = tid + 256 v -> (row, col, channel)
    int wrc[FWV];
#pragma unroll
    for (int v = 0; v < FWV; ++v) {
        const int f = tid + v * 256;
        const int row = f / (FW * 3), rem = f - row * (FW * 3), col = rem / 3, c = rem - col * 3;
        wrc[v] = f < FWIN ? (row | (col << 8) | (c << 16)) : -1;
    }
    if (tid < 64) bias_l[tid] = p.bias[tid];
    if (tid < 2) { fwin0[FWIN + tid] = tid ? 0.f : 1.f; fwin0[FWIN_PITCH + FWIN + tid] = tid ? 0.f : 1.f; }

    struct Item { int b, oy0, ox0; };
    auto decode = [&](int item) {
        Item t;
        if (item < nitems) {
            const int tx = item % tiles_x, r = item / tiles_x;
            t.b = r / tiles_y; t.oy0 = (r - t.b * tiles_y) * HT; t.ox0 = tx * HT;
        } else { t.b = 0; t.oy0 = -(1 << 20); t.ox0 = -(1 << 20); }                 // past the end: every window pixel out of range (zeros)
        return t;
    };
    char* const dump = (char*)(bias_l + 64) + tid * 8;       // where a lane without a real destination writes
    float wreg[FWV];
    auto win_fetch = [&](const Item& t) {
        const hp3d_rsrc_t irsrc = HP3D_MAKE_RSRC(p.in + (size_t)t.b * p.H * p.W * 3, (unsigned)(p.H * p.W) * 12u);
#pragma unroll
        for (int v = 0; v < FWV; ++v) {
            const int yy = t.oy0 - 2 + (wrc[v] & 255), xx = t.ox0 - 2 + ((wrc[v] >> 8) & 255), c = wrc[v] >> 16;
            const bool ok = wrc[v] >= 0 && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            wreg[v] = HP3D_BUFFER_LOAD4(irsrc, ok ? ((yy * p.W + xx) * 3 + c) * 4 : OOR, 0);
        }
    };
    auto win_commit = [&](int wb) {
#pragma unroll
        for (int v = 0; v < FWV; ++v) {
            float* dst = (v + 1) * 256 <= FWIN || wrc[v] >= 0 ? fwin0 + wb * FWIN_PITCH + tid + v * 256 : (float*)dump;      // (no divergent branch)
            *dst = wreg[v];
        }
    };
    // the patch of item t = conv1_1 over its window, row block blk = wave + 4 j (j = 0..2; 11 blocks, so wave 3's third one is a dry run into a
    // dump slot), cut into 11 slices of four pieces each; the item loop puts one piece in front of each of a K-step's four MFMAs:
    //   slice 0: the 16 operand floats from the staged window; 1: rounding to halves; 2: the four MFMAs (K = 27 + 2 bias rows in two steps, two
    //   cout halves); 3..10: (cout half nb, register group a): leaky-ReLU, rounding, zero outside the image, one 8-byte LDS write
    int bpix[3], bwin[3], bpyx[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int pp = (wave + 4 * j) * 32 + li;
        const int ppc = pp < HPW * HPW ? pp : 0;
        const int py = ppc / HPW, px = ppc - py * HPW;
        bpix[j] = pp < HPW * HPW ? pp * (HPITCH * 4) + 4 * lh * 2 : -1;     // byte offset inside a patch buffer (-1: dump slot)
        bwin[j] = (py * FW + px) * 3;
        bpyx[j] = pp < HPW * HPW ? (py | (px << 8)) : (255 | (255 << 8));   // (dry run: "outside the image")
    }
    float br[16];
    f32x4 bah[2];
    f32x16 bc[2];
    unsigned bmask = 0, bword[2] = {0, 0};
    auto build_piece = [&](const Item& t, int pb, int wb, int j, int slice, int pc) {
        if (slice == 0) {
            const float* fw = fwin0 + wb * FWIN_PITCH;
            const float* wb0 = fw + bwin[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = 4 * pc + i;
                // slots with k >= 27 (the two bias rows' constant 1.0, the K tail's 0.0: step 1, lh = 1, slots 3..7) are absolute window indices
                br[q] = (q >= 11) ? (lh ? fw[fkoff[q]] : wb0[fkoff[q]]) : wb0[fkoff[q]];
            }
        } else if (slice == 1) {
            f16x4 hh;
#pragma unroll
            for (int i = 0; i < 4; ++i) hh[i] = (hp3d_f16)br[4 * pc + i];
            const f32x2 two = __builtin_bit_cast(f32x2, hh);
            bah[pc >> 1][(pc & 1) * 2] = two[0]; bah[pc >> 1][(pc & 1) * 2 + 1] = two[1];
        } else if (slice == 2) {
            if (pc == 0) { HP3D_MFMA_OPERAND_FENCE(); HP3D_MFMA_32x32x16_F16_V_FIRST(bc[0], bwh[0][0], bah[0]); }
            if (pc == 1) HP3D_MFMA_32x32x16_F16_V_FIRST(bc[1], bwh[1][0], bah[0]);
            if (pc == 2) HP3D_MFMA_32x32x16_F16_V(bc[0], bwh[0][1], bah[1]);
            if (pc == 3) HP3D_MFMA_32x32x16_F16_V(bc[1], bwh[1][1], bah[1]);
        } else {
            const int nb = (slice - 3) >> 2, a = (slice - 3) & 3;
            if (slice == 3 && pc == 0) {
                HP3D_MFMA_RESULT_FENCE2(bc[0], bc[1]);
                const int gy = t.oy0 - 1 + (bpyx[j] & 255), gx = t.ox0 - 1 + (bpyx[j] >> 8);
                bmask = ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) ? 0xffffffffu : 0u;
            }
            const float v = bc[nb][4 * a + pc];
            br[pc] = hp3d_vmax(v, HP3D_LEAKY_SLOPE * v);
            if (pc & 1) {                  // two halves per 32-bit word; outside the image (conv1_2's SAME padding pads conv1_1's OUTPUT): +0.0
                f16x4 h2;
                h2[0] = (hp3d_f16)br[pc - 1]; h2[1] = (hp3d_f16)br[pc]; h2[2] = h2[3] = (hp3d_f16)0.f;
                bword[pc >> 1] = __builtin_bit_cast(u32x2, h2)[0] & bmask;
            }
            if (pc == 3) {
                char* dst = bpix[j] >= 0 ? (char*)smem + pb * (HPATCH_FLOATS * 4) + bpix[j] + (32 * nb + 8 * a) * 2 : dump;
                *(u32x2*)dst = u32x2{bword[0], bword[1]};
            }
        }
    };

    // prologue: windows of this workgroup's first two items, the first item's patch
    int item = blockIdx.x;
    Item cur_t = decode(item), nxt_t = decode(item + (int)gridDim.x);
    win_fetch(cur_t); win_commit(0);
    win_fetch(nxt_t); win_commit(1);
    __syncthreads();
    static_assert((NBLK + 3) / 4 == 3, "");
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int sl = 0; sl < 11; ++sl)
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) build_piece(cur_t, 0, 0, j, sl, pc);
    __syncthreads();

    for (int n = 0; item < nitems; item += (int)gridDim.x, ++n) {
        const int cur = n & 1;
        const Item nn_t = decode(item + 2 * (int)gridDim.x);
        f32x16 acc[2][2];
        f32x4 fa[2][2];
        f32x4 bias4[2][4];
        auto a_fetch = [&](int set, int toff_b, int ks) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                fa[set][mt] = *(const f32x4*)((const char*)smem + cur * (HPATCH_FLOATS * 4) + abase[mt] + toff_b + ks * 32);
        };
        a_fetch(0, 0, 0);
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            HP3D_SCHED_BARRIER();
            if (s == 0) win_fetch(nn_t);                                  // the window after next: global -> registers ...
            if (s == 23) win_commit(cur);                                 // ... -> LDS (this item's window is spent: its patch is complete)
            if (s == 35) {                                                // the epilogue's biases (the build's registers are free by now)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int a = 0; a < 4; ++a) bias4[nt][a] = *(const f32x4*)(bias_l + nt * 32 + a * 8 + lh * 4);
            }
            if (s + 1 < 36) {
                const int t1 = (s + 1) >> 2, r1 = t1 / 3, c1 = t1 - r1 * 3;
                a_fetch((s + 1) & 1, (r1 * HPW + c1) * HPITCH * 4, (s + 1) & 3);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mt = q >> 1, nt = q & 1;
                HP3D_SCHED_BARRIER();
                if (s % 12 < 11) build_piece(nxt_t, cur ^ 1, cur ^ 1, s / 12, s % 12, q);          // the next item's patch, block wave + 4 (s / 12)
                HP3D_SCHED_BARRIER();
                if (s == 0) HP3D_MFMA_32x32x16_F16_ACC_FIRST("a", acc[mt][nt], wres[s][nt], fa[s & 1][mt]);
                else if (s < WRES_AGPR_STEPS) HP3D_MFMA_32x32x16_F16_ACC("a", acc[mt][nt], wres[s][nt], fa[s & 1][mt]);
                else HP3D_MFMA_32x32x16_F16_ACC("v", acc[mt][nt], wres[s][nt], fa[s & 1][mt]);
            }
        }
        HP3D_SCHED_BARRIER();
        HP3D_MFMA_RESULT_FENCE4(acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
        __syncthreads();            // patch[cur] is spent (the slabs below reuse it), patch[cur ^ 1] and the staged window are complete

        // ---- epilogue: accumulator (lane (li, lh), register 4 a + e) of (mt, nt) = pixel li of row block 2 wave + mt, cout 32 nt + 8 a + 4 lh + e.
        //      This wave's two slabs = the patch rows it builds itself (blocks wave and wave + 4 of patch[cur]: 32 pixels x 144 B each).
        const hp3d_rsrc_t orsrc = HP3D_MAKE_RSRC((hp3d_f16*)p.out + (size_t)cur_t.b * Hs * Ws * p.out_cs, (unsigned)(Hs * Ws) * (unsigned)p.out_cs * 2u);
        constexpr int SPITCH = HPITCH * 4;                                   // 144 B per slab pixel: 64 couts + pad (conflict-free as above)
        const int sp_w = pdy * 16 + pdx;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            char* sl = (char*)smem + cur * (HPATCH_FLOATS * 4) + (wave + 4 * mt) * (32 * SPITCH);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    f16x4 h;
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {           // (pairs: v_pk_add_f32 / v_pk_mul_f32)
                        const f32x2 x = f32x2{acc[mt][nt][a * 4 + 2 * e2], acc[mt][nt][a * 4 + 2 * e2 + 1]} + f32x2{bias4[nt][a][2 * e2], bias4[nt][a][2 * e2 + 1]};
                        const f32x2 y = x * HP3D_LEAKY_SLOPE;
                        h[2 * e2] = (hp3d_f16)hp3d_vmax(x[0], y[0]); h[2 * e2 + 1] = (hp3d_f16)hp3d_vmax(x[1], y[1]);
                    }
                    *(f16x4*)(sl + sp_w * SPITCH + lh * 8 + nt * 64 + a * 16) = h;
                }
        }
        HP3D_WAVE_LDS_SYNC();
        const int g = lane & 7, pp = lane >> 3;                              // 16-byte group of the pixel's 64 couts, pooled column 0..7
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const char* sl = (const char*)smem + cur * (HPATCH_FLOATS * 4) + (wave + 4 * mt) * (32 * SPITCH);
            f16x8 m = *(const f16x8*)(sl + (2 * pp) * SPITCH + g * 16);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const f16x8 o = *(const f16x8*)(sl + ((w >> 1) * 16 + 2 * pp + (w & 1)) * SPITCH + g * 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = m[e] > o[e] ? m[e] : o[e];
            }
            const int yp = (cur_t.oy0 >> 1) + wave * 2 + mt, xp = (cur_t.ox0 >> 1) + pp;
            const int off = (yp < Hs && xp < Ws) ? (yp * Ws + xp) * (p.out_cs * 2) + g * 16 : OOR;
            HP3D_BUFFER_STORE16(orsrc, __builtin_bit_cast(f32x4, m), off, 0);
        }
        HP3D_WAVE_LDS_SYNC();        // (the interpreter's fibers: the slab reads end before this wave's next build slices overwrite them)
        cur_t = nxt_t; nxt_t = nn_t;
    }
}

template <int NT, bool POOL, int KS = 3>
void h16_launch_t(const ConvParams& p, hipStream_t s) {
    static bool attr_done[64] = {};
    constexpr int SLABS = 4 * 4 * 32 * (64 * NT + 16);                       // epilogue: 4 waves x 4 row blocks x 32 pixels
    constexpr int PATCH1 = (HT + KS - 1) * (HT + KS - 1) * HPITCH * 4;       // one patch buffer of this filter size
    constexpr int WPS0 = NT == 4 ? 1 : NT == 2 ? 2 : 3;
    constexpr int SMEM0 = ((WPS0 == 1 ? 2 : 1) * PATCH1) > SLABS ? ((WPS0 == 1 ? 2 : 1) * PATCH1) : SLABS;
    constexpr int WPS = SMEM0 * WPS0 <= 160 * 1024 ? WPS0 : WPS0 - 1;         // (7x7, one cout block per wave: two workgroups of 70 KB, not three)
    static_assert(WPS >= 1 && (WPS > 1 || KS == 3), "");
    auto k = conv_h16_kernel<NT, POOL, WPS, false, KS>;
    constexpr int PATCHES = (WPS == 1 ? 2 : 1) * PATCH1;
    constexpr int SMEM = PATCHES > SLABS ? PATCHES : SLABS;
    static_assert(SMEM * WPS <= 160 * 1024, "LDS per CU");
    if (hp3d_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    const long items = (long)p.B * ((p.Ho + HT - 1) / HT) * ((p.Wo + HT - 1) / HT) * (p.Cout / (64 * NT));
    const int slots = hp3d_num_cus() * WPS;
    HP3D_LAUNCH(k, dim3((unsigned)(items < slots ? items : slots)), dim3(256), SMEM, s, p);
}

#ifndef HP3D_H16_MAXNT
#define HP3D_H16_MAXNT 4
#endif
// per-wave cout blocks: 4 (one workgroup per CU, double-buffered patch) only where the K loop is long enough to amortise the
// serial load / store phases of a lone workgroup (Cin >= 512: measured 1199 vs 1153 TFLOP/s at conv4_2); below that two
// workgroups per CU with 2 blocks each cover each other's phases (conv3_1 955 vs 811, conv3_2 1094 vs 985)
inline int h16_nt(int Cout, int cin_units) {
    const int nt = (Cout % 256 == 0 && cin_units >= 256) ? 4 : Cout % 128 == 0 ? 2 : 1;
    return nt < HP3D_H16_MAXNT ? nt : HP3D_H16_MAXNT;
}

void h16_fused12_launch_t(const ConvParams& p, hipStream_t s) {
    static bool attr_done[64] = {};
#ifndef HP3D_H16_FWPS
#define HP3D_H16_FWPS 2
#endif
    constexpr int WPS = HP3D_H16_FWPS;
    auto k = conv_h16_kernel<1, true, WPS, true>;
    constexpr int SLABS = 4 * 4 * 32 * (64 + 16);
    constexpr int PATCH = HPATCH_FLOATS * 4 + ((HPW + 2) * (HPW + 2) * 3 + 2 + 2) * 4;       // patch + staged image window + constants
    constexpr int SMEM = PATCH > SLABS ? PATCH : SLABS;
    if (hp3d_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    const long items = (long)p.B * ((p.Ho + HT - 1) / HT) * ((p.Wo + HT - 1) / HT);
    const int slots = hp3d_num_cus() * WPS;
    HP3D_LAUNCH(k, dim3((unsigned)(items < slots ? items : slots)), dim3(256), SMEM, s, p);
}

// the filter-resident form: one workgroup per CU walking items with a two-deep software pipeline -- worth it from a few items per workgroup on
void h16_first_resident_launch(const ConvParams& p, hipStream_t s) {
    static bool attr_done[64] = {};
    constexpr int SMEM = 2 * HPATCH_FLOATS * 4 + 2 * ((HPW + 2) * (HPW + 2) * 3 + 4) * 4 + 64 * 4 + 256 * 8;     // patches, windows, biases, dump slots
    static_assert(SMEM <= 160 * 1024, "LDS per CU");
    if (hp3d_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute((const void*)conv_h16_first_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    const long items = (long)p.B * ((p.Ho + HT - 1) / HT) * ((p.Wo + HT - 1) / HT);
    const int slots = hp3d_num_cus();
    HP3D_LAUNCH(conv_h16_first_kernel, dim3((unsigned)(items < slots ? items : slots)), dim3(256), SMEM, s, p);
}

}  // namespace

// p.first_form: 0 = by size (the filter-resident form from four items per CU on), 1 = the two-workgroups-per-CU form, 2 = the filter-resident form.
// Returns the form that ran (1 / 2) or -1.
int conv_h16_fused12_launch(const ConvParams& p, hipStream_t s) {
    if (p.Cout != 64 || p.Cin != 32 || !p.wpk1 || !p.bias1 || ((p.Ho | p.Wo) & 1) || !p.act || p.out_cs != 64 || p.cout_store != 64) return -1;
    const long items = (long)p.B * ((p.Ho + HT - 1) / HT) * ((p.Wo + HT - 1) / HT);
    const bool resident = p.first_form == 2 || (p.first_form == 0 && items >= 4L * hp3d_num_cus());
    if (resident) h16_first_resident_launch(p, s); else h16_fused12_launch_t(p, s);
    return resident ? 2 : 1;
}

// 3x3 / stride 1, half-precision operands and output (not the float32 score-map heads), Cin a multiple of 64 halves,
// Cout a multiple of 64, output pixel stride a multiple of 8 halves (16-byte stores); enough work items to fill the chip.
// Returns the per-wave cout blocks NT (1, 2 or 4) or 0.  mode 1: only when the grid fills the chip; mode 2 (tests): whenever
// the shape allows
int conv_h16_eligible(int mode, int k, int stride, int cin_units, int Cout, int Ho, int Wo, int B, int out_f32, int out_cs) {
    if (!mode || (k != 3 && k != 7 && k != 1) || stride != 1 || out_f32 || cin_units % 32 || cin_units < 32 || Cout % 64 || out_cs % 8) return 0;
    int nt = h16_nt(Cout, cin_units);
    if (k != 3 && nt == 4) nt = 2;                      // 7x7 / 1x1: the single-buffer forms only
    const long items = (long)B * ((Ho + HT - 1) / HT) * ((Wo + HT - 1) / HT) * (Cout / (64 * nt));
    return (mode == 2 || items >= 256) ? nt : 0;
}

int conv_h16_launch(const ConvParams& p, int pool, hipStream_t s) {
    int nt = p.Cout % 64 == 0 ? h16_nt(p.Cout, p.Cin) : 0;
    if (!nt || (pool && ((p.Ho | p.Wo) & 1))) return -1;
    const int k = 2 * p.pad_t + 1;                      // SAME padding, stride 1: the filter size
    if (k == 7 || k == 1) {
        if (pool) return -1;
        if (nt == 4) nt = 2;
        if (k == 7) { if (nt == 2) h16_launch_t<2, false, 7>(p, s); else h16_launch_t<1, false, 7>(p, s); }
        else { if (nt == 2) h16_launch_t<2, false, 1>(p, s); else h16_launch_t<1, false, 1>(p, s); }
        return 0;
    }
    if (k != 3) return -1;
    if (nt == 4) { if (pool) h16_launch_t<4, true>(p, s); else h16_launch_t<4, false>(p, s); }
    else if (nt == 2) { if (pool) h16_launch_t<2, true>(p, s); else h16_launch_t<2, false>(p, s); }
    else { if (pool) h16_launch_t<1, true>(p, s); else h16_launch_t<1, false>(p, s); }
    return 0;
}

// conv_wino.hip -- 3x3 / stride-1 convolution by Winograd F(2x2, 3x3) on the f32 matrix cores.
//
// Same call sites as conv_mfma.hip (NetworkOps.conv_relu + max_pool, utils/general.py:36-65) for the layers
// with Cin % 64 == 0 and Cout % 128 == 0: 2.25x fewer multiply-adds than the direct form at float32 (no
// reduced precision; the transforms only add / subtract / halve).
//
//   Y(2x2) = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A          d: 4x4 input window, g: 3x3 filter
//
// Design for gfx950 (measured facts behind it: scripts/micro/mfma_chain.hip, mfma_valu.hip):
//   * one dependent chain of v_mfma_f32_32x32x2_f32 per wave already runs at 98% of peak, but every VALU
//     instruction issued on the SIMD costs the f32 MFMA stream ~3.5-5 cycles (they do not overlap); LDS and
//     VMEM instructions cost nothing.  So the kernel is built around VALU instructions per MFMA:
//   * workgroup = 32 Winograd tiles x 128 output channels, 4 waves = ONE wave per SIMD (147 KB of LDS, 512
//     registers per lane); wave w owns couts 32w..32w+31 and all 32 tiles;
//   * the 16 Winograd planes accumulate over ALL input channels in 16 x 16 = 256 accumulation registers
//     (AGPRs, the matrix core's own C/D operands): the main loop has no output transform at all -- per
//     32-channel step 256 MFMAs against ~150 VALU instructions (the input transform).  The output transform
//     A^T M A runs once per work item in the epilogue.  (An earlier version folded every plane into the 2x2
//     outputs inside the loop to fit 256 registers and two workgroups per CU: 2.7 VALU per MFMA, 70% MFMA-busy.)
//   * the transformed input V = B^T d B is double buffered in LDS (2 x 16 planes x 32 tiles x 32 channels): the
//     loader (thread = (tile, channel quad)) has the next step's 4x4 windows in flight during a step (64
//     registers), transforms them under the last planes and writes the other buffer: ONE barrier per step;
//   * transformed weights U[plane][Cin][Cout] are pre-packed in fragment order and go global -> VGPR directly
//     (buffer_load_dwordx4, scalar offset per (plane, step)), ring of 4 planes, 3 ahead: no LDS for B;
//   * persistent grid (one workgroup per CU) walking work items blockIdx.x, +gridDim.x, ...; the next item's first
//     window / weight fragments are fetched under the current item's last step;
//   * the 2x2 outputs of a tile are one pooling window: bias + leaky-ReLU + max-pool stay a register epilogue;
//   * 7x7 filters (PoseNet2D refinement units) run on the same kernel: the filter, zero-extended to 9x9, is nine 3x3
//     blocks; block (i,j) is a 3x3 convolution of the input shifted by (3i-2, 3j-2), and all nine accumulate into
//     the SAME 16 planes (the output transform is linear) -- a 3x3 layer with 9 x Cin virtual channels: 144 instead
//     of 196 multiplies per 2x2 outputs, 45 steps per item.
#include "hp3d_common.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace {

// Two shapes of work item (template parameter NT = Winograd tiles per item):
//   NT = 32: 32 tiles x 128 couts, 32-channel steps, V pitch 36 floats (padding keeps ds_read_b128 conflict-free);
//   NT = 64: 64 tiles x  64 couts, 16-channel steps (layers with Cout % 128 != 0: conv1_2 64 -> 64): wave w owns
//            couts 32 (w & 1) .. +31 of tiles 32 (w >> 1) .. +31; V pitch 16 floats with the channel quad XOR-swizzled
//            by (tile >> 2) & 3 instead of padding (2 x 64 KB must fit next to the tables).
template <int NT> struct WinoCfg {
    static constexpr int CK = NT == 32 ? 32 : 16;           // channels per step
    static constexpr int NQ = CK / 4;                        // channel quads per step (loader threads per tile)
    static constexpr int G = CK / 8;                         // 16-B fragments per lane, plane and step (4 MFMAs each)
    static constexpr int LDA = NT == 32 ? 36 : 16;           // V row pitch in floats
    static constexpr int COUTS = NT == 32 ? 128 : 64;        // output channels per item
    static constexpr int PLANE_FLOATS = NT * LDA;            // one plane of one buffer
    static constexpr int VBUF_FLOATS = 16 * PLANE_FLOATS;    // 73728 B / 65536 B
    static constexpr int SMEM_BYTES = 2 * VBUF_FLOATS * 4 + 2 * 2 * NT * 4;     // 2 V buffers + two tile tables
    static constexpr int PL_PER_BASE = NT == 32 ? 14 : 16;   // planes reachable from one ds_read base (16-bit immediate)
};

// SPLITK (small batches: too few items to fill the chip): a work item additionally owns a contiguous range of the channel
// steps; it stores RAW partial outputs (the output transform is linear, so partial sums of M transform to partial sums of
// Y) into p.out = [ksplit][B*Ho*Wo][Cout], and conv_splitk_reduce adds them in split order (+ bias, activation).
template <bool POOL, int NT, int NSUB, bool SPLITK>
HP3D_KERNEL2(256, 1)
void conv_wino_kernel(const ConvParams p) {
    static_assert(!(POOL && SPLITK), "the fused max-pool needs complete sums");
    using Cfg = WinoCfg<NT>;
    constexpr int WCK = Cfg::CK, WLDA = Cfg::LDA, WTILES = NT, PLANE_FLOATS = Cfg::PLANE_FLOATS, VBUF_FLOATS = Cfg::VBUF_FLOATS;
    constexpr int NQ = Cfg::NQ, G = Cfg::G, COUTS = Cfg::COUTS;
    HP3D_DYN_SMEM(V);
    // tile tables, double buffered by item parity: [0..NT-1] output offset of tile t (-1: no such tile), [NT..2NT-1] edge flags;
    int* tinfo = (int*)(V + 2 * VBUF_FLOATS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = HP3D_READFIRSTLANE(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wcout = NT == 32 ? wave : (wave & 1);           // this wave's 32-cout block inside the item
    const int tbase = NT == 32 ? 0 : (wave >> 1) * 32;        // ... and its first tile
    auto swz = [](int t) { return NT == 32 ? 0 : ((t >> 2) & 3); };      // quad swizzle of tile t's V row

    // Work item = (cout block of 128) x (32 Winograd tiles).  The 32 tiles are consecutive entries of the
    // flattened (image, band of 4 tile rows, tile column, row in band) order: a 4x8 footprint where the tile
    // grid allows it, but no padding when it does not (20x20 tiles at 40x40 would waste 20% in fixed 4x8
    // blocks), and an item may continue into the next image.
    const int TXn = p.tiles_x, TYn = p.tiles_y, per_img = TXn * TYn;
    const int tile_blocks = (p.B * per_img + WTILES - 1) / WTILES;
    const int ncy = p.Cout / COUTS;
    const int per_split = tile_blocks * ncy;                  // items of one channel split
    const int nitems = per_split * (SPLITK ? p.ksplit : 1);
    auto tile_decode = [&](int id, int& tb, int& tyy, int& txx) {
        tb = id / per_img;
        const int r = id - tb * per_img;
        const int band = r / (4 * TXn), rem = r - band * 4 * TXn;
        const int rows = min(4, TYn - 4 * band);
        txx = rem / rows;
        tyy = band * 4 + rem - txx * rows;
    };
    const int Hs = POOL ? (p.Ho >> 1) : p.Ho, Ws = POOL ? (p.Wo >> 1) : p.Wo;
    auto table_write = [&](int tblock, int parity, int kz) {
        if (tid < WTILES) {
            int tb, tyy, txx;
            tile_decode(tblock * WTILES + tid, tb, tyy, txx);
            int off = -1, fl = 0;
            if (tb < p.B) {
                if (POOL) {
                    if (tyy < Hs && txx < Ws) off = ((tb * Hs + tyy) * Ws + txx) * p.out_cs;
                } else {
                    off = (((SPLITK ? kz * p.B + tb : tb) * Hs + 2 * tyy) * Ws + 2 * txx) * p.out_cs;
                    fl = (2 * txx + 1 < Ws ? 1 : 0) | (2 * tyy + 1 < Hs ? 2 : 0);
                }
            }
            tinfo[parity * 2 * WTILES + tid] = off;
            tinfo[parity * 2 * WTILES + WTILES + tid] = fl;
        }
    };

    // ---- loader role: this thread transforms the 4x4 window of tile lt for channel quad lc --------------
    // (buffer loads: a window element outside the image gets an out-of-range offset and reads as 0)
    const int lt = tid / NQ, lc = tid % NQ;
    const int cs4 = p.in_cs * 4;
    constexpr int OOR = (int)0x80000000;
    int wv[16];               // byte offsets of the 16 window elements (OOR: zero padding)
    // NSUB = 1: a 3x3 filter.  NSUB = 9: a 7x7 filter as the nine 3x3 blocks of its zero-extended 9x9 form; block
    // sub = 3i + j reads the input shifted by (3i - 2, 3j - 2), so the tile coordinates stay live through the item
    // (cb, cty, ctx_: only the NSUB = 9 instantiation keeps them; the 3x3 kernels are short of registers as it is).
    int cb = 0, cty = 0, ctx_ = 0;
    auto window_offsets = [&](bool valid, int lb, int lty, int ltx, int sub) {
        const int dy = NSUB == 1 ? 0 : 3 * (sub / 3) - 2, dx = NSUB == 1 ? 0 : 3 * (sub % 3) - 2;
        const int wy0 = 2 * lty - 1 + dy, wx0 = 2 * ltx - 1 + dx;                    // SAME padding 1 (3 for 7x7)
        const int wbase = ((lb * p.H + wy0) * p.W + wx0) * cs4 + lc * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool in = valid && lb < p.B && (unsigned)(wy0 + r) < (unsigned)p.H && (unsigned)(wx0 + c) < (unsigned)p.W;
                wv[r * 4 + c] = in ? wbase + (r * p.W + c) * cs4 : OOR;
            }
    };
    auto loader_setup = [&](int tblock, bool valid, int sub) {
        int lb, lty, ltx;
        tile_decode(tblock * WTILES + lt, lb, lty, ltx);
        if (NSUB > 1) { cb = valid ? lb : p.B; cty = lty; ctx_ = ltx; }
        window_offsets(valid, lb, lty, ltx, (NSUB > 1 && SPLITK) ? sub : 0);
    };
    auto loader_shift = [&](int sub) { window_offsets(true, cb, cty, ctx_, sub); };
    const hp3d_rsrc_t irsrc = HP3D_MAKE_RSRC(p.in, (unsigned)p.B * (unsigned)(p.H * p.W) * (unsigned)cs4);
    const hp3d_rsrc_t orsrc = HP3D_MAKE_RSRC(p.out, (unsigned)(SPLITK ? p.ksplit * p.B : p.B) * (unsigned)(Hs * Ws) * (unsigned)p.out_cs * 4u);

    f32x4 d[16];
    auto window_fetch = [&](int soff) {
#pragma unroll
        for (int e = 0; e < 16; ++e) d[e] = HP3D_BUFFER_LOAD16(irsrc, wv[e], soff);
    };
    auto transform_commit = [&](int buf) {
        // B^T d B with B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
        f32x4 t[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            t[0 * 4 + c] = d[0 * 4 + c] - d[2 * 4 + c];
            t[1 * 4 + c] = d[1 * 4 + c] + d[2 * 4 + c];
            t[2 * 4 + c] = d[2 * 4 + c] - d[1 * 4 + c];
            t[3 * 4 + c] = d[1 * 4 + c] - d[3 * 4 + c];
        }
        float* Vq = V + buf * VBUF_FLOATS + lt * WLDA + (lc ^ swz(lt)) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4 v0 = t[r * 4 + 0] - t[r * 4 + 2];
            const f32x4 v1 = t[r * 4 + 1] + t[r * 4 + 2];
            const f32x4 v2 = t[r * 4 + 2] - t[r * 4 + 1];
            const f32x4 v3 = t[r * 4 + 1] - t[r * 4 + 3];
            *(f32x4*)(Vq + (r * 4 + 0) * PLANE_FLOATS) = v0;
            *(f32x4*)(Vq + (r * 4 + 1) * PLANE_FLOATS) = v1;
            *(f32x4*)(Vq + (r * 4 + 2) * PLANE_FLOATS) = v2;
            *(f32x4*)(Vq + (r * 4 + 3) * PLANE_FLOATS) = v3;
        }
    };

    // ---- MFMA role ----------------------------------------------------------------------------------
    // packed U: [plane 16][Cin/32][Cout/32][g 4][h 2][n 32][j 4] -> the 4 fragments a wave needs for one
    // (plane, step) are 4 KB contiguous: base = one scalar offset, g = an immediate
    const int CO32 = p.Cout >> 5;
    const int nsub_rt = NSUB == 1 ? 1 : p.nsub;                     // (9; kept a run-time value on purpose: the constant-
                                                                    //  folded form of this loop nest allocates registers much worse)
    const int csteps = p.Cin / WCK;                                 // steps per sub-kernel
    const int nsteps = nsub_rt * csteps;
    const hp3d_rsrc_t wrsrc = HP3D_MAKE_RSRC(p.wpk, (unsigned)(16 * nsub_rt * p.Cin) * (unsigned)p.Cout * 4u);
    const int chunk_stride_b = CO32 * 4096;                         // bytes between 32-channel chunks
    const int plane_stride_b = nsub_rt * (p.Cin >> 5) * chunk_stride_b;    // bytes between planes
    // a 16-channel step is one half (g = 0,1 / 2,3: 2 KB) of a 32-channel chunk
    auto soff_of = [&](int plane, int step) {
        return NT == 32 ? plane * plane_stride_b + step * chunk_stride_b
                        : plane * plane_stride_b + (step >> 1) * chunk_stride_b + (step & 1) * 2048;
    };

    f32x16 M[16];          // the 16 plane accumulators (AGPRs), live across the whole item
    f32x4 bq[4][G];        // B fragments of four planes in flight
    auto b_fetch = [&](int set, int voff, int soff) {
#pragma unroll
        for (int g = 0; g < G; ++g) bq[set][g] = HP3D_BUFFER_LOAD16(wrsrc, voff + g * 1024, soff);
    };
    // A fragments: this lane's row (tile tbase + li) of the current V buffer, fragment g = channels 8g + 4 lh .. +3,
    // i.e. quad 2g + lh.  NT = 32: one base per 14 planes so that every ds_read offset fits the 16-bit immediate (an
    // address add per read would be a VALU instruction in the MFMA stream), g is an immediate; NT = 64: the quad is
    // swizzled, so g = 0 / 1 have their own base (they differ in address bit 5), all 16 planes fit the immediate.
    const int arow = tbase + li;
    const int va_lane0 = (arow * WLDA + ((lh) ^ swz(arow)) * 4) * 4;
    const int va_lane1 = (arow * WLDA + ((2 + lh) ^ swz(arow)) * 4) * 4;
    int ab0 = 0, ab1 = 0;
    f32x4 af[2][G];
    auto a_fetch = [&](int set, int plane) {
        if (NT == 32) {
            const int base = plane < 14 ? ab0 : ab1, pl = plane < 14 ? plane : plane - 14;
#pragma unroll
            for (int g = 0; g < G; ++g) af[set][g] = *(const f32x4*)((const char*)V + base + (pl * PLANE_FLOATS + g * 8) * 4);
        } else {
            af[set][0] = *(const f32x4*)((const char*)V + ab0 + plane * PLANE_FLOATS * 4);
            af[set][G - 1] = *(const f32x4*)((const char*)V + ab1 + plane * PLANE_FLOATS * 4);
        }
    };

    // ---- first item: the only exposed prologue -------------------------------------------------------------
    // item -> (channel split kz, cout block cy, tile block); split kz owns the steps [s0, s1) (>= 2 each)
    auto split_of = [&](int it, int& kz, int& cy_, int& tb_) {
        kz = SPLITK ? it / per_split : 0;
        const int r = SPLITK ? it - kz * per_split : it;
        cy_ = r / tile_blocks;
        tb_ = r - cy_ * tile_blocks;
    };
    // (every quantity here is wave-uniform; say so, or hipcc wraps the buffer loads that take them as scalar offsets into
    //  waterfall loops)
    auto first_step_of = [&](int kz) { return SPLITK ? HP3D_READFIRSTLANE((kz * nsteps) / p.ksplit) : 0; };
    int item = blockIdx.x;
    int kz, cy, tblock;
    split_of(item, kz, cy, tblock);
    int s0 = first_step_of(kz), s1 = SPLITK ? first_step_of(kz + 1) : nsteps;
    const int sub0 = SPLITK ? HP3D_READFIRSTLANE(s0 / csteps) : 0;
    int sub_cur = sub0;
    loader_setup(tblock, true, sub0);
    table_write(tblock, 0, kz);
    int wvoff = (cy * (COUTS / 32) + wcout) * 4096 + lane * 16;
    window_fetch((s0 - sub0 * csteps) * (WCK * 4));
    b_fetch(0, wvoff, soff_of(0, s0));
    b_fetch(1, wvoff, soff_of(1, s0));
    b_fetch(2, wvoff, soff_of(2, s0));
    transform_commit(0);
    __syncthreads();
    int cur = 0;

    for (int k = 0;; ++k) {
        int n_cy = cy, n_tblock = tblock, n_wvoff = wvoff, n_kz = kz, n_s0 = s0;
        const int n_item = item + (int)gridDim.x;      // static round robin (a global work counter measured 0.5 % slower:
                                                       // its returning atomic sits in the in-order vmcnt queue of wave 0)
        const int co = cy * COUTS + wcout * 32 + li;
        const float bias = SPLITK ? 0.f : p.bias[co];           // in flight during the item, used in the epilogue

        // one step (32 or 16 channels): 16 planes x 16 or 8 MFMAs on V[cur]; the first step of an item starts the accumulators
        // from the inline constant 0
        auto step_body = [&](int step, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const bool lasts = step + 1 == s1;
            // block (i, j) = (sub_cur / 3, sub_cur % 3) of the 9x9 extension this step's channels belong to
            const bool za = sub_cur >= 6, zb = sub_cur == 2 || sub_cur == 5 || sub_cur == 8;
            const int skip_a = (NSUB == 9 && !FIRST) ? HP3D_OPAQUE_SGPR(za ? 1 : 0) : 0;
            const int skip_b = (NSUB == 9 && !FIRST) ? HP3D_OPAQUE_SGPR(zb ? 1 : 0) : 0;
            const int skip_ab = (NSUB == 9 && !FIRST) ? HP3D_OPAQUE_SGPR((za || zb) ? 1 : 0) : 0;
            // Loads return in issue order (one vmcnt counter), so the first wait on a weight fragment issued AFTER
            // the window loads also waits for the windows: the ring is topped up to 4 planes (0..3) first, the
            // windows go next, and B(p+4) is issued behind plane p's MFMAs -- the windows (the next step's, or the
            // next item's first) then have 4 planes (~1.8 us) to arrive before anything depends on them.
            const int nvoff = lasts ? n_wvoff : wvoff;
            const int nstep = lasts ? (SPLITK ? n_s0 : 0) : step + 1;
            ab0 = cur * (VBUF_FLOATS * 4) + va_lane0;
            ab1 = NT == 32 ? ab0 + 14 * PLANE_FLOATS * 4 : cur * (VBUF_FLOATS * 4) + va_lane1;
            HP3D_OPAQUE_V(ab0);
            HP3D_OPAQUE_V(ab1);
            a_fetch(0, 0);                       // first: plane 0's MFMAs wait for exactly this
            b_fetch(3, wvoff, soff_of(3, step));
            // (virtual) channels of the next step: sub-kernel nsub_ = (step + 1) / csteps, channel step ncs
            const int nsub_ = NSUB == 1 ? 0 : SPLITK ? HP3D_READFIRSTLANE(nstep / csteps) : lasts ? 0 : (step + 1) / csteps;
            const int ncs = NSUB == 1 ? nstep : nstep - nsub_ * csteps;      // (NSUB == 1: nsteps == csteps)
            if (lasts) loader_setup(n_tblock, n_item < nitems, nsub_);
            else if (NSUB > 1 && ncs == 0) loader_shift(nsub_);
#pragma unroll
            for (int pl = 0; pl < 16; ++pl) {           // fully unrolled: accumulator and ring indices are static
                HP3D_SCHED_BARRIER();
                if (pl < 15) a_fetch((pl & 1) ^ 1, pl + 1);
                // 7x7 filters: in the edge blocks of the zero-extended 9x9 filter (i = 2 or j = 2: one filter row / column of
                // three) the transformed filter G g G^T has a zero row a = 3 / column b = 3, i.e. planes 12..15 / 3, 7, 11, 15
                // contribute nothing: 121 instead of 144 plane-steps per item.  Their MFMAs are skipped (not on an item's
                // first step, which also zero-initialises the accumulators).
                const bool zplane = NSUB == 9 && !FIRST && ((pl >> 2) == 3 || (pl & 3) == 3);
                if (zplane) {
                    const int skip = pl == 15 ? skip_ab : (pl >> 2) == 3 ? skip_a : skip_b;
#pragma unroll
                    for (int g = 0; g < G; ++g) HP3D_MFMA4_UNLESS(M[pl], af[pl & 1][g], bq[pl & 3][g], skip);
                } else {
                    if (FIRST) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        M[pl] = HP3D_MFMA_32x32x2(af[pl & 1][0][0], bq[pl & 3][0][0], zero);
                    } else {
                        M[pl] = HP3D_MFMA_32x32x2(af[pl & 1][0][0], bq[pl & 3][0][0], M[pl]);
                    }
#pragma unroll
                    for (int gj = 1; gj < 4 * G; ++gj)
                        M[pl] = HP3D_MFMA_32x32x2(af[pl & 1][gj >> 2][gj & 3], bq[pl & 3][gj >> 2][gj & 3], M[pl]);
                }
                if (pl == 0) {               // the window loads are issued between plane 0's MFMAs, not in front of them
                    window_fetch(NSUB > 1 ? HP3D_READFIRSTLANE(ncs * (WCK * 4)) : ncs * (WCK * 4));   // (uniform; hipcc cannot always tell)
                    HP3D_SCHED_GROUP(HP3D_SG_DS_READ, G);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        HP3D_SCHED_GROUP(HP3D_SG_MFMA, G / 2);
                        HP3D_SCHED_GROUP(HP3D_SG_VMEM_READ, 2);
                    }
                }
                // weight prefetch 4 planes ahead into the slot this plane just released; past the step: the next
                // step's planes 0..2 (its plane 3 is fetched at its start, before its windows)
                const int t = pl + 4;
                if (t < 16) b_fetch(t & 3, wvoff, soff_of(t, step));
                else if (t < 19) b_fetch(t & 3, nvoff, soff_of(t - 16, nstep));
                if (pl == 12) transform_commit(cur ^ 1);      // its LDS writes land under planes 13..15
            }
            HP3D_SCHED_BARRIER();
            __syncthreads();             // V[cur^1] complete, V[cur] free
            cur ^= 1;
            sub_cur = nsub_;             // the block of the step that runs next (this item's or the next item's first)
        };
        step_body(s0, std::true_type{});
        {   // the next item (its tile table is written here, hidden under this item's MFMAs)
            const bool has_next = n_item < nitems;
            if (has_next) split_of(n_item, n_kz, n_cy, n_tblock);
            if (SPLITK) { n_kz = HP3D_READFIRSTLANE(n_kz); n_cy = HP3D_READFIRSTLANE(n_cy); n_tblock = HP3D_READFIRSTLANE(n_tblock); }
            n_s0 = first_step_of(n_kz);
            table_write(n_tblock, (k + 1) & 1, n_kz);
            n_wvoff = (n_cy * (COUTS / 32) + wcout) * 4096 + lane * 16;
        }
        for (int step = s0 + 1; step < s1; ++step) step_body(step, std::false_type{});

        // ---- epilogue: output transform Y = A^T M A (A^T = [1 1 1 0; 0 1 -1 -1]), bias + leaky-ReLU
        //      (+ 2x2 max-pool) + NHWC store.  Nothing else runs on this SIMD meanwhile, so it is kept short and
        //      branch-free: row sums first (every plane element is read once), four accumulator registers at a
        //      time, buffer stores (an invalid lane gets an out-of-range offset and is dropped).
        const int* tab = tinfo + (k & 1) * 2 * WTILES;
        const bool cok = co < p.cout_store;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {         // four accumulator registers (= 4 tiles per lane) at a time
            int vo[4], fl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = tbase + q + 8 * rg + 4 * lh;      // MFMA row = Winograd tile
                const int off = tab[t];
                fl[q] = POOL ? 0 : tab[WTILES + t];
                vo[q] = (cok && off >= 0) ? (off + co) * 4 : OOR;
            }
            float y[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = rg * 4 + q;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float s0 = M[a * 4 + 0][r] + M[a * 4 + 1][r] + M[a * 4 + 2][r];      // column j = 0
                    const float s1 = M[a * 4 + 1][r] - M[a * 4 + 2][r] - M[a * 4 + 3][r];      // column j = 1
                    if (a == 0) { y[0][q] = s0; y[1][q] = s1; }
                    else if (a == 1) { y[0][q] += s0; y[1][q] += s1; y[2][q] = s0; y[3][q] = s1; }
                    else if (a == 2) { y[0][q] += s0; y[1][q] += s1; y[2][q] -= s0; y[3][q] -= s1; }
                    else { y[2][q] -= s0; y[3][q] -= s1; }
                }
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    float x = y[o][q] + bias;
                    if (!SPLITK && p.act) x = fmaxf(x, HP3D_LEAKY_SLOPE * x);
                    y[o][q] = x;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (POOL) {
                    HP3D_BUFFER_STORE4(orsrc, fmaxf(fmaxf(y[0][q], y[1][q]), fmaxf(y[2][q], y[3][q])), vo[q], 0);
                } else {
                    HP3D_BUFFER_STORE4(orsrc, y[0][q], vo[q], 0);
                    HP3D_BUFFER_STORE4(orsrc, y[1][q], (fl[q] & 1) ? vo[q] : OOR, p.out_cs * 4);
                    HP3D_BUFFER_STORE4(orsrc, y[2][q], (fl[q] & 2) ? vo[q] : OOR, Ws * p.out_cs * 4);
                    HP3D_BUFFER_STORE4(orsrc, y[3][q], fl[q] == 3 ? vo[q] : OOR, (Ws + 1) * p.out_cs * 4);
                }
            }
        }
        if (n_item >= nitems) break;
        item = n_item; cy = n_cy; tblock = n_tblock; wvoff = n_wvoff;
        if (SPLITK) { kz = n_kz; s0 = n_s0; s1 = first_step_of(n_kz + 1); }
    }
}

}  // namespace

// U = G g G^T per (virtual cin, cout), packed for the kernel: [plane][chunk][Cout/32][g][h][n][j] with virtual channel
// 32*chunk + 8*g + 4*h + j and cout 32*co32 + n (zero padded).  k = 3: virtual channel = engine channel e.  k = 7: nine
// 3x3 blocks (i,j) of the filter zero-extended to 9x9, virtual channel = (3i + j) * cin_pad + e.  chan_map[e] = reference
// channel of engine channel e (-1: padding); NULL = identity.
size_t wino_packed_floats(int k, int cin_pad, int cout_pad) { return (size_t)16 * (k == 7 ? 9 : 1) * cin_pad * cout_pad; }

void wino_pack_weights(const float* g_hwio /*[k][k][Cin][Cout]*/, int k, int Cin, int Cout, int cin_pad, int cout_pad,
                       const int* chan_map, float* dst) {
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const int nsub = k == 7 ? 9 : 1;
    const int nch = nsub * cin_pad / 32, CO32 = cout_pad / 32;
    memset(dst, 0, sizeof(float) * wino_packed_floats(k, cin_pad, cout_pad));
    for (int sub = 0; sub < nsub; ++sub) {
        const int u0 = k == 7 ? 3 * (sub / 3) : 0, v0 = k == 7 ? 3 * (sub % 3) : 0;
        for (int e = 0; e < cin_pad; ++e) {
            const int rc = chan_map ? chan_map[e] : (e < Cin ? e : -1);
            if (rc < 0) continue;
            const int vc = sub * cin_pad + e;
            const int chunk = vc >> 5, g = (vc >> 3) & 3, h = (vc >> 2) & 1, j = vc & 3;
            for (int co = 0; co < Cout; ++co) {
                float w3[3][3];
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c)
                        w3[r][c] = (u0 + r < k && v0 + c < k) ? g_hwio[((size_t)((u0 + r) * k + (v0 + c)) * Cin + rc) * Cout + co] : 0.f;
                for (int a = 0; a < 4; ++a)
                    for (int b = 0; b < 4; ++b) {
                        float s = 0.f;
                        for (int r = 0; r < 3; ++r)
                            for (int c = 0; c < 3; ++c) s += G[a][r] * w3[r][c] * G[b][c];
                        dst[(((((size_t)(a * 4 + b) * nch + chunk) * CO32 + (co >> 5)) * 4 + g) * 2 + h) * 128 + (co & 31) * 4 + j] = s;
                    }
            }
        }
    }
}

// mode 1 (auto): only when the grid fills the chip -- directly (>= 256 work items) or, for layers without a fused pool,
// after splitting the channel steps over up to 16 workgroups (small batches: hp3d_posenet2d at B = 1 has 8 .. 128 items
// per layer); mode 2 (forced, tests): whenever the shape allows.  Returns the tile count of the item shape (32: Cout %
// 128 == 0, 64: Cout % 64 == 0) or 0; *ksplit (may be NULL) receives the channel split to launch with.
int conv_wino_eligible(int mode, int k, int stride, int Cin, int Cout, int Ho, int Wo, int B, int in_cs, int out_cs,
                       int pool, int* ksplit) {
    if (ksplit) *ksplit = 1;
    if (mode == 0 || (k != 3 && k != 7) || stride != 1 || Cin % 32) return 0;
    const int nt = Cout % 128 == 0 ? 32 : Cout % 64 == 0 ? 64 : 0;
    if (!nt || (k == 3 && nt == 32 && Cin % 64)) return 0;    // at least two steps per item (a 7x7 filter has 9 x Cin/32)
    // the kernel addresses both tensors with 32-bit byte offsets (stride 1, SAME: input extent = output extent); the
    // executor sizes its chunks so that the largest layer passes (engine.hip:auto_micro_batch)
    if ((long)B * Ho * Wo * in_cs * 4 >= (1L << 31) || (long)B * Ho * Wo * out_cs * 4 >= (1L << 31)) return 0;
    const long tiles = (long)B * ((Ho + 1) / 2) * ((Wo + 1) / 2);
    const long items = (tiles + nt - 1) / nt * (Cout / (nt == 32 ? 128 : 64));
    if (items >= 256) return nt;
    // under-filled: split the channel steps (each split keeps >= 2 steps; partial sums are float32 [ksplit][pix][Cout])
    const int nsteps = (k == 7 ? 9 : 1) * Cin / (nt == 32 ? 32 : 16);
    int ks = (int)((256 + items - 1) / items);
    if (ks > nsteps / 2) ks = nsteps / 2;
    if (ks > 16) ks = 16;
    if (ks >= 2 && ksplit && !(pool && ((Ho | Wo) & 1)) && (long)ks * B * Ho * Wo * Cout * 4 < (1L << 31) && (mode == 2 || items * ks >= 96)) {
        *ksplit = ks;
        return nt;
    }
    return mode == 2 ? nt : 0;
}

template <bool POOL, int NT, int NSUB, bool SPLITK>
static void wino_launch_t(const ConvParams& p, long tiles, hipStream_t s) {
    using Cfg = WinoCfg<NT>;
    static bool attr_done[64] = {};
    auto k = conv_wino_kernel<POOL, NT, NSUB, SPLITK>;
    if (hp3d_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM_BYTES);
    const long items = (tiles + NT - 1) / NT * (p.Cout / Cfg::COUTS) * (SPLITK ? p.ksplit : 1);
    const int slots = hp3d_num_cus();                     // persistent grid: one workgroup per CU (of the current device)
    dim3 grid((unsigned)(items < slots ? items : slots));
    HP3D_LAUNCH(k, grid, dim3(256), Cfg::SMEM_BYTES, s, p);
}

// pin.ksplit > 1: pin.out must be the partial-sum scratch [ksplit][B*Ho*Wo][Cout] with out_cs = cout_store = Cout; the
// caller runs conv_splitk_reduce afterwards (bias + activation happen there).
int conv_wino_launch(const ConvParams& pin, int pool, hipStream_t s) {
    // 32-bit byte / element offsets inside the kernel (buffer loads, the tile table)
    const long kso = pin.ksplit > 1 ? pin.ksplit : 1;
    if ((long)pin.B * pin.H * pin.W * pin.in_cs * 4 >= (1L << 31) || kso * pin.B * pin.Ho * pin.Wo * pin.out_cs * 4 >= (1L << 31)) return -1;
    if (pin.nsub != 1 && pin.nsub != 9) return -1;
    ConvParams p = pin;
    p.tiles_x = (p.Wo + 1) / 2;          // Winograd tiles per row / column
    p.tiles_y = (p.Ho + 1) / 2;
    const long tiles = (long)p.B * p.tiles_x * p.tiles_y;
    const int nt = p.Cout % 128 == 0 ? 32 : p.Cout % 64 == 0 ? 64 : 0;
    if (!nt || (pool && p.nsub != 1)) return -1;
    if (p.ksplit > 1) {
        const int nsteps = p.nsub * p.Cin / (nt == 32 ? 32 : 16);
        if (pool || p.ksplit * 2 > nsteps || p.out_cs != p.Cout) return -1;
        if (p.nsub == 9) {
            if (nt == 32) wino_launch_t<false, 32, 9, true>(p, tiles, s); else wino_launch_t<false, 64, 9, true>(p, tiles, s);
        } else {
            if (nt == 32) wino_launch_t<false, 32, 1, true>(p, tiles, s); else wino_launch_t<false, 64, 1, true>(p, tiles, s);
        }
        return 0;
    }
    p.ksplit = 1;
    if (p.nsub == 9) {
        if (nt == 32) wino_launch_t<false, 32, 9, false>(p, tiles, s); else wino_launch_t<false, 64, 9, false>(p, tiles, s);
    } else if (nt == 32) {
        if (pool) wino_launch_t<true, 32, 1, false>(p, tiles, s); else wino_launch_t<false, 32, 1, false>(p, tiles, s);
    } else {
        if (pool) wino_launch_t<true, 64, 1, false>(p, tiles, s); else wino_launch_t<false, 64, 1, false>(p, tiles, s);
    }
    return 0;
}

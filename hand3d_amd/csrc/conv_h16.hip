// conv_h16.hip -- 3x3 / stride-1 / SAME convolution in half precision (f16 operands, float32 accumulate) for the
// half-precision trunks of BASELINE config 5: bias + leaky-ReLU (+ 2x2 max-pool) fused, halves out.
//
// Same call sites as conv_mfma.hip's F16 instantiations (NetworkOps.conv_relu + max_pool, utils/general.py:36-65; trunk layer
// lists nets/ColorHandPose3DNetwork.py:144-157,183-199) and the same arithmetic (v_mfma_f32_32x32x16_f16 on the same packed
// weights, same row -> pixel map), but built for the half-precision matrix pipe, which consumes operands 16x faster per FLOP
// than the f32 one.  What the ablations of the general kernel showed (profiles/r02_tuning_notes.md): at 16 MFMAs x 32 cycles
// per K step its weight LDS-DMA pieces, fragment reads and per-step barrier cost more than the MFMAs.  Hence, like conv_wino:
//   * ONE wave per SIMD (`__launch_bounds__(256, 1)`), 4 x NT register tiles per wave: 128 pixels x (32 NT) couts, up to 256
//     accumulators; workgroup = 2 x 2 waves = 16 x 16 output pixels x (64 NT) couts;
//   * weights go global -> VGPR straight in MFMA fragment order (the f16 blob's 1-KB pieces), ring of RING K-steps,
//     scalar offset per (tap, K-step): no LDS, no DMA issue slots, no barrier for B;
//   * the 18 x 18 x 64-half input patch (with halo) is DOUBLE buffered in LDS: the next chunk's pieces are in flight during
//     the current chunk and are committed to the other buffer near its end -- ONE barrier per 64-channel chunk (576 MFMAs per
//     wave) instead of one per 16;
//   * the nine taps x four K-steps of a chunk are one straight-line block: every LDS fragment address is base + immediate;
//   * persistent grid over (image, tile, cout block) items.
#include "hp3d_common.h"
#include <algorithm>
#ifndef HP3D_H16_ABL
#define HP3D_H16_ABL 0          // timing ablations (scripts/build_variant.sh); any non-zero value computes wrong results
#endif

namespace {

constexpr int HT = 16;                    // output tile: 16 x 16 pixels
constexpr int HPW = HT + 2;               // patch width / height (halo 1)
constexpr int HPITCH = 36;                // floats per patch pixel: 64 halves = 32 floats + 4 pad (144 B: conflict-free b128)
constexpr int HPATCH_FLOATS = HPW * HPW * HPITCH;        // 11664 floats = 46.7 KB per buffer
constexpr int HPIECES = HPW * HPW * 8;    // 16-byte pieces per patch
constexpr int HPVEC = (HPIECES + 255) / 256;             // pieces per thread (11)
constexpr int HRING = 6;                  // K-steps of weight fragments in flight per wave

template <int NT, bool POOL>
HP3D_KERNEL2(256, 1)
void conv_h16_kernel(const ConvParams p) {
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

    // A fragment rows of this lane: M-block mt of pixel half wm covers output rows 8 wm + 2 mt, +1 (32 pixels = 2 rows x 16);
    // MFMA row r -> quad r >> 2, (dy, dx) = ((r >> 1) & 1, r & 1): four accumulator registers = one 2x2 pooling window
    int abase[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int q = li >> 2, dx = li & 1, dy = (li >> 1) & 1;
        const int ly = (wm * 4 + mt) * 2 + dy, lx = 2 * q + dx;
        abase[mt] = ((ly * HPW + lx) * HPITCH + lh * 4) * 4;          // bytes
    }
    const hp3d_rsrc_t wrsrc = HP3D_MAKE_RSRC(p.wpk, (unsigned)(9 * p.Cin) * (unsigned)p.Cout * 4u);
    const int tap_stride_b = KB * CO32 * 1024;           // bytes between taps of the packed weights

    for (int item = blockIdx.x; item < nitems; item += (int)gridDim.x) {
        int it = item;
        const int cb = it % ncb; it /= ncb;
        const int tx = it % tiles_x; it /= tiles_x;
        const int ty = it % tiles_y;
        const int b = it / tiles_y;
        const int oy0 = ty * HT, ox0 = tx * HT;
        const float* inb = p.in + (size_t)b * p.H * p.W * p.in_cs;
        // this thread's patch pieces: piece idx -> (patch pixel, 16-byte slot); -1 = zero fill (SAME padding / image edge)
        int poff[HPVEC];
#pragma unroll
        for (int v = 0; v < HPVEC; ++v) {
            const int idx = tid + v * 256;
            const int pix = idx >> 3, c4 = idx & 7;
            const int py = pix / HPW, px = pix - py * HPW;
            const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
            const bool ok = idx < HPIECES && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            poff[v] = ok ? (gy * p.W + gx) * p.in_cs + c4 * 4 : -1;
        }
        f32x4 preg[HPVEC];
        auto patch_fetch = [&](int chunk) {
#pragma unroll
            for (int v = 0; v < HPVEC; ++v) {
                f32x4 val = {0.f, 0.f, 0.f, 0.f};
                if (poff[v] >= 0) val = *(const f32x4*)(inb + poff[v] + chunk * 32);
                preg[v] = val;
            }
        };
        auto patch_commit = [&](int buf) {
#pragma unroll
            for (int v = 0; v < HPVEC; ++v) {
                const int idx = tid + v * 256;
                if (idx < HPIECES) *(f32x4*)(smem + buf * HPATCH_FLOATS + (idx >> 3) * HPITCH + (idx & 7) * 4) = preg[v];
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

        __syncthreads();                       // the previous item's waves are done with both patch buffers
        patch_fetch(0);
#pragma unroll
        for (int s = 0; s < HRING - 1; ++s) b_fetch(s, s >> 2, s & 3);      // K-steps 0 .. RING-2 of chunk 0: (tap, ks) = (s / 4, s % 4)
        patch_commit(0);
        __syncthreads();
        int cur = 0;
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const bool has_next = chunk + 1 < nchunks;
            if (has_next && !(HP3D_H16_ABL & 2)) patch_fetch(chunk + 1);
            a_fetch(0, cur, 0, 0);
            // 36 K-steps of this chunk, straight line: step s = 4 tap + ks
#pragma unroll
            for (int s = 0; s < 36; ++s) {
                const int tap = s >> 2, ks = s & 3;
                HP3D_SCHED_BARRIER();
                // prefetch: A fragments of step s+1 (other register set), weight fragments of step s + RING - 1
                if (s + 1 < 36 && !(HP3D_H16_ABL & 8)) {
                    const int t1 = (s + 1) >> 2, r1 = t1 / 3, c1 = t1 - r1 * 3;
                    a_fetch((s + 1) & 1, cur, (r1 * HPW + c1) * HPITCH * 4, (s + 1) & 3);
                }
                {
                    const int s2 = s + HRING - 1;
                    if (HP3D_H16_ABL & 4) { asm volatile("" : "+v"(fb[s2 % HRING][0])); }
                    else if (s2 < 36) b_fetch(s2 % HRING, s2 >> 2, chunk * 4 + (s2 & 3));
                    else if (has_next) b_fetch(s2 % HRING, (s2 - 36) >> 2, (chunk + 1) * 4 + ((s2 - 36) & 3));
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = HP3D_MFMA_32x32x16_F16(fa[s & 1][mt], fb[s % HRING][nt], acc[mt][nt]);
                if (s == 30 && has_next) patch_commit(cur ^ 1);      // lands under the last taps
                (void)tap; (void)ks;
            }
            HP3D_SCHED_BARRIER();
            __syncthreads();                   // patch[cur ^ 1] complete, patch[cur] free
            cur ^= 1;
        }

        // ---- epilogue: bias + leaky-ReLU (+ 2x2 max-pool), halves out.  Buffer stores with 32-bit offsets inside the image
        //      (an invalid lane / pixel gets an out-of-range offset and is dropped): no branches, no 64-bit address math
        if (HP3D_H16_ABL & 1) { if (acc[0][0][0] == 12345.f) p.out[0] = acc[1][0][3] + acc[3][NT - 1][7]; continue; }
        constexpr int OOR = (int)0x80000000;
        const hp3d_rsrc_t orsrc = HP3D_MAKE_RSRC((hp3d_f16*)p.out + (size_t)b * Hs * Ws * p.out_cs,
                                                 (unsigned)(Hs * Ws) * (unsigned)p.out_cs * 2u);
        const int row_b = Ws * p.out_cs * 2, px_b = p.out_cs * 2;            // bytes per output row / pixel
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = cb * BN + (wn * NT + nt) * 32 + li;
            const float bias = p.bias[co];
            const bool cok = co < p.cout_store;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int y0 = oy0 + (wm * 4 + mt) * 2;                       // conv-output row of (dy = 0)
#pragma unroll
                for (int a4 = 0; a4 < 4; ++a4) {
                    const int x0 = ox0 + 2 * (2 * a4 + lh);                  // quad 2 a4 + lh of registers 4 a4 .. 4 a4 + 3
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[mt][nt][a4 * 4 + e] + bias;
                        if (p.act) x = fmaxf(x, HP3D_LEAKY_SLOPE * x);
                        v[e] = x;
                    }
                    if (POOL) {
                        const int yp = y0 >> 1, xp = x0 >> 1;
                        const float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                        HP3D_BUFFER_STORE2(orsrc, m, (cok && yp < Hs && xp < Ws) ? yp * row_b + xp * px_b + co * 2 : OOR, 0);
                    } else {
                        const int base = (cok && y0 < Hs && x0 < Ws) ? y0 * row_b + x0 * px_b + co * 2 : OOR;
                        const int bx = x0 + 1 < Ws ? base : OOR, by = y0 + 1 < Hs ? base : OOR;
                        HP3D_BUFFER_STORE2(orsrc, v[0], base, 0);
                        HP3D_BUFFER_STORE2(orsrc, v[1], bx, px_b);
                        HP3D_BUFFER_STORE2(orsrc, v[2], by, row_b);
                        HP3D_BUFFER_STORE2(orsrc, v[3], (x0 + 1 < Ws) ? by : OOR, row_b + px_b);
                    }
                }
            }
        }
    }
}

template <int NT, bool POOL>
void h16_launch_t(const ConvParams& p, hipStream_t s) {
    static bool attr_done[64] = {};
    auto k = conv_h16_kernel<NT, POOL>;
    constexpr int SMEM = 2 * HPATCH_FLOATS * 4;
    if (hp3d_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    const long items = (long)p.B * ((p.Ho + HT - 1) / HT) * ((p.Wo + HT - 1) / HT) * (p.Cout / (64 * NT));
    const int slots = hp3d_num_cus();
    HP3D_LAUNCH(k, dim3((unsigned)(items < slots ? items : slots)), dim3(256), SMEM, s, p);
}

}  // namespace

// 3x3 / stride 1, half-precision operands and output (not the float32 score-map heads), Cin a multiple of 64 halves,
// Cout a multiple of 64; enough work items to fill the chip.  Returns the per-wave cout blocks NT (1, 2 or 4) or 0.
// mode 1: only when the grid fills the chip; mode 2 (tests): whenever the shape allows
int conv_h16_eligible(int mode, int k, int stride, int cin_units, int Cout, int Ho, int Wo, int B, int out_f32) {
    if (!mode || k != 3 || stride != 1 || out_f32 || cin_units % 32 || cin_units < 32 || Cout % 64) return 0;
    const int nt = Cout % 256 == 0 ? 4 : Cout % 128 == 0 ? 2 : 1;
    const long items = (long)B * ((Ho + HT - 1) / HT) * ((Wo + HT - 1) / HT) * (Cout / (64 * nt));
    return (mode == 2 || items >= 256) ? nt : 0;
}

int conv_h16_launch(const ConvParams& p, int pool, hipStream_t s) {
    const int nt = p.Cout % 256 == 0 ? 4 : p.Cout % 128 == 0 ? 2 : p.Cout % 64 == 0 ? 1 : 0;
    if (!nt || (pool && ((p.Ho | p.Wo) & 1))) return -1;
    if (nt == 4) { if (pool) h16_launch_t<4, true>(p, s); else h16_launch_t<4, false>(p, s); }
    else if (nt == 2) { if (pool) h16_launch_t<2, true>(p, s); else h16_launch_t<2, false>(p, s); }
    else { if (pool) h16_launch_t<1, true>(p, s); else h16_launch_t<1, false>(p, s); }
    return 0;
}

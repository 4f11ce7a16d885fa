// conv_mfma.hip -- the general (direct) convolution: NHWC float32 / float16 implicit GEMM on the
// gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, = an fmaf chain, 157 TF peak).  It serves every layer
// shape of the path; for large batches the 3x3 / 7x7 layers take conv_wino.hip and conv1_1 takes conv_first.hip
// (engine.hip:run_conv), small batches, 1x1 / stride-2 layers, the f16 mode and conv_impl=direct stay here.
//
// Stands in for tf.nn.conv2d(SAME) + bias_add + leaky-ReLU (+ 2x2 max-pool) at every
// NetworkOps.conv / conv_relu / max_pool call site of the reference
// (utils/general.py:36-65; layer lists nets/ColorHandPose3DNetwork.py:144-161,183-214,255-258,291-294).
//
// Design (MI355X-first, not a cuDNN re-creation):
//   * one workgroup = TH x TW output pixels of ONE image x BN output channels;
//   * the input patch (with its k-1 halo) for a 32-channel slice is staged ONCE in LDS and all
//     k*k filter taps are contracted out of it (the A operand is re-addressed, never re-loaded);
//   * weights are pre-packed at load time into MFMA fragment order
//        wpk[tap][Cin/8][Cout/32][h:2][n:32][j:4]   (h = lane>>5, n = lane&31)
//     so the B fragment of a wave is ONE lane-linear ds_read_b128 and the global->LDS copy is a
//     straight 16-B-per-lane stream;
//   * the k index of each MFMA pair is permuted (lanes 0-31 take channels 8g..8g+3, lanes 32-63
//     channels 8g+4..8g+7) so the A fragment is ONE ds_read_b128 per 4 MFMAs as well;
//   * MFMA row r of a 32-pixel tile maps to pixel quad r>>2, (dy,dx) = ((r>>1)&1, r&1): the four
//     accumulator registers 4a..4a+3 of a lane are one 2x2 pooling window -> the max-pool is a
//     register-only epilogue;
//   * next tap's weights are prefetched into registers while the current tap computes; one
//     barrier per tap; two workgroups per CU hide the remaining staging latency.
#include "hp3d_common.h"
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int CK = 32;        // channels per K-chunk
constexpr int LDA = CK + 4;   // patch row pitch in floats (144 B: odd multiple of 16 B)

template <int KS, int STRIDE, int TH, int TW, int WM, int WN, int MT, int NT>
struct ConvCfg {
    static constexpr int NTHR = 64 * WM * WN;
    static constexpr int BM = TH * TW;
    static constexpr int BN = WN * NT * 32;
    static constexpr int PH = (TH - 1) * STRIDE + KS;
    static constexpr int PW = (TW - 1) * STRIDE + KS;
    static constexpr int PATCH_FLOATS = ((PH * PW * LDA + 3) / 4) * 4;
    static constexpr int WBUF_FLOATS = CK * BN;
    static constexpr int SMEM_BYTES = (PATCH_FLOATS + 3 * WBUF_FLOATS) * 4;
    static constexpr int WVEC = (CK * BN / 4) / NTHR;   // float4 of weights per thread per tap
    static_assert(BM == WM * MT * 32, "tile/wave mismatch");
    static_assert(TH == WM * MT * (32 / TW), "tile rows mismatch");
    static_assert(TW == 8 || TW == 16, "TW must be 8 or 16");
    static_assert((CK * BN / 4) % NTHR == 0, "weight copy must divide evenly");
};

// F16 = true: half-precision operands on v_mfma_f32_32x32x16_f16 (f32 accumulate).  An f16 NHWC tensor
// is addressed as a float tensor of channel PAIRS, which makes the whole data path -- patch geometry
// (64 f16 = 128 B per pixel per chunk), weight pieces (1 KB per k-block x 32 couts), LDS-DMA, fragment
// reads (16 B per lane) -- byte-identical to the f32 path: ConvParams counts channels in 4-byte units.
// Only the MFMA (one K=16 instruction per 16-B fragment pair instead of four K=2), the epilogue store
// and the conv1_1 row builder differ.
template <int KS, int STRIDE, int TH, int TW, int WM, int WN, int MT, int NT, bool POOL, bool F16>
HP3D_KERNEL(64 * WM * WN)
void conv_mfma_kernel(const ConvParams p) {
    using C = ConvCfg<KS, STRIDE, TH, TW, WM, WN, MT, NT>;
    constexpr int NTHR = C::NTHR, BN = C::BN, PH = C::PH, PW = C::PW, TAPS = KS * KS;
    constexpr int ROWS_PER_MT = 32 / TW;   // pixel rows covered by one 32-pixel MFMA tile
    constexpr int QX = TW / 2;             // 2x2 quads per tile row

    HP3D_DYN_SMEM(smem);
    float* patch = smem;
    float* wbuf = smem + C::PATCH_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // ---- which tile -------------------------------------------------------------------
    int sp = blockIdx.x;
    const int tx = sp % p.tiles_x; sp /= p.tiles_x;
    const int ty = sp % p.tiles_y;
    const int b = sp / p.tiles_y;
    const int n0 = blockIdx.y * BN;
    const int oy0 = ty * TH, ox0 = tx * TW;                 // tile origin (conv output coords)
    const int gy0 = oy0 * STRIDE - p.pad_t, gx0 = ox0 * STRIDE - p.pad_l;   // patch origin (input coords)

    const int C8 = p.Cin >> 3, CO32 = p.Cout >> 5;
    // split-K (under-filled chip, small batch): workgroup z contracts chunks [cbeg, cend) only and
    // writes raw partial sums; conv_splitk_reduce adds them in z order (+ bias, activation)
    const int nc_all = p.Cin / CK;
    const int kz = blockIdx.z;
    const int cbeg = (kz * nc_all) / p.ksplit, cend = ((kz + 1) * nc_all) / p.ksplit;
    const float* inb = p.in + (size_t)b * p.H * p.W * p.in_cs + cbeg * CK;

    // ---- per-lane A fragment base offsets ---------------------------------------------
    const int li = lane & 31, lh = lane >> 5;
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int t = wm * MT + mt;
        const int q = li >> 2, dx = li & 1, dy = (li >> 1) & 1;
        const int ly = t * ROWS_PER_MT + 2 * (q / QX) + dy;
        const int lx = 2 * (q % QX) + dx;
        abase[mt] = ((ly * STRIDE) * PW + lx * STRIDE) * LDA + lh * 4;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // patch staging, split into issue (global -> registers) and commit (registers -> LDS) so the
    // next chunk's patch is in flight while the current chunk's last tap computes
    constexpr int PVEC = (PH * PW * 8 + NTHR - 1) / NTHR;
    int poff[PVEC];        // element offset of this thread's v-th float4 inside the image, -1 = zero fill
#pragma unroll
    for (int v = 0; v < PVEC; ++v) {
        const int idx = tid + v * NTHR;
        const int pix = idx >> 3, c4 = idx & 7;
        const int py = pix / PW, px = pix - py * PW;
        const int gy = gy0 + py, gx = gx0 + px;
        const bool ok = idx < PH * PW * 8 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        poff[v] = ok ? (gy * p.W + gx) * p.in_cs + c4 * 4 : -1;
    }
    f32x4 preg[PVEC];
    auto patch_fetch = [&](int chunk) {
#pragma unroll
        for (int v = 0; v < PVEC; ++v) {
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (poff[v] >= 0) val = *(const f32x4*)(inb + poff[v] + chunk * CK);
            preg[v] = val;
        }
    };
    auto patch_commit = [&]() {
#pragma unroll
        for (int v = 0; v < PVEC; ++v) {
            const int idx = tid + v * NTHR;
            if (idx < PH * PW * 8) *(f32x4*)(patch + (idx >> 3) * LDA + (idx & 7) * 4) = preg[v];
        }
    };

    // ---- software pipeline -----------------------------------------------------------------------
    // step it = chunk*TAPS + tap.  Weight tile W(it) lives in wbuf[it % 3] and gets there by LDS-DMA
    // (buffer_load_dwordx4 ... lds: no VGPR round trip, no ds_write; the per-step part of the source
    // address is ONE scalar offset, the per-lane part a loop-invariant VGPR -> zero VALU per DMA):
    //   mid step it      : ONE barrier (the explicit vmcnt(0) before it retires the DMA of W(it+1), issued
    //                      one step earlier; it also proves every wave left step it-1: wbuf[(it+2)%3] is free)
    //   right after it   : issue the DMA of W(it+2)
    //   end of step it   : A/B fragments of step it+1, group 0 are read ahead into registers
    // The input patch is single-buffered: at a chunk boundary (once per k*k steps) two barriers bracket
    // its re-fill; its global loads are issued after the mid barrier of the chunk's last step.
    // For 3x3 the nine taps are unrolled (9 % 3 == 0 makes the ring index static too), so every LDS
    // address in the loop is base-register + immediate.
    const int nchunks = cend - cbeg;
    const int total = nchunks * TAPS;
    constexpr int NWAVES = NTHR / 64;
    constexpr int NB = BN / 32;                       // 1-KB pieces per 8-channel group
    const int wave_u = HP3D_READFIRSTLANE(wave);
    const hp3d_rsrc_t wrsrc = HP3D_MAKE_RSRC(p.wpk, (unsigned)(TAPS * p.Cin) * (unsigned)p.Cout * 4u);
    const int tap_stride_b = C8 * CO32 * 1024;        // bytes between taps of the packed weights
    const int chunk_stride_b = 4 * CO32 * 1024;       // bytes between 32-channel chunks
    const int w_base_b = (n0 >> 5) * 1024 + cbeg * chunk_stride_b;
    int wvoff[C::WVEC];                               // loop-invariant per-lane byte offsets of this wave's pieces
#pragma unroll
    for (int v = 0; v < C::WVEC; ++v) {
        const int pc = v * NWAVES + wave;
        wvoff[v] = (pc / NB) * (CO32 * 1024) + (pc % NB) * 1024 + lane * 16;
    }
    auto w_dma = [&](int tp, int ch, int bufidx) {
        const int soff = w_base_b + tp * tap_stride_b + ch * chunk_stride_b;          // scalar
#pragma unroll
        for (int v = 0; v < C::WVEC; ++v)
            HP3D_BUFFER_LDS16(wrsrc, wbuf + bufidx * C::WBUF_FLOATS + (v * NWAVES + wave_u) * 256, wvoff[v], soff, lane);
    };
    f32x4 fa[2][MT], fb[2][NT];
    const int bbase = (wn * NT) * 256 + lane * 4;
    auto load_frags = [&](int set, int toff, int g, int bufidx) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa[set][mt] = *(const f32x4*)(patch + abase[mt] + (toff + g * 8));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            fb[set][nt] = *(const f32x4*)(wbuf + bbase + (bufidx * C::WBUF_FLOATS + g * (BN * 8) + nt * 256));
    };
    auto mfma_group = [&](int set) {
        if constexpr (F16) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = HP3D_MFMA_32x32x16_F16(fa[set][mt], fb[set][nt], acc[mt][nt]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = HP3D_MFMA_32x32x2(fa[set][mt][j], fb[set][nt][j], acc[mt][nt]);
        }
    };
    auto tap_off = [&](int tap) {
        const int r = tap / KS, s = tap - r * KS;
        return (r * PW + s) * LDA;
    };
    // one pipeline step; with constant (tap, buf) after unrolling everything below folds to immediates
    auto step = [&](int tap, int chunk, int buf, int it) {
        const int buf1 = (buf + 1) % 3, buf2 = (buf + 2) % 3;
        const bool chunk_end = (tap + 1 == TAPS);
        const bool has_next = (it + 1 < total);
        const int toff = tap_off(tap);
        load_frags(1, toff, 1, buf);
        mfma_group(0);
        load_frags(0, toff, 2, buf);
        mfma_group(1);
        HP3D_SCHED_BARRIER();
        // hipcc (ROCm 7.2) does NOT carry a pending LDS-DMA across the loop back-edge into the barrier's
        // wait: retire this wave's DMA of W(it+1) explicitly before the rendezvous.
        HP3D_WAIT_VMCNT0();
        __syncthreads();
        if (it + 2 < total) {
            const int tp2 = (tap + 2 < TAPS) ? tap + 2 : tap + 2 - TAPS;     // TAPS == 1: handled below
            const int ch2 = (tap + 2 < TAPS) ? chunk : chunk + 1;
            if (TAPS == 1) w_dma(0, chunk + 2, buf2); else w_dma(tp2, ch2, buf2);
        }
        if (chunk_end && has_next) patch_fetch(chunk + 1);
        HP3D_SCHED_BARRIER();
        load_frags(1, toff, 3, buf);
        mfma_group(0);
        if (has_next && !chunk_end) load_frags(0, tap_off(tap + 1), 0, buf1);
        mfma_group(1);
        HP3D_SCHED_BARRIER();
        if (chunk_end && has_next) {
            __syncthreads();          // every wave is done reading this chunk's patch
            patch_commit();
            __syncthreads();
            load_frags(0, tap_off(0), 0, buf1);
        }
    };

    bool patch_ready = false;
    if constexpr (KS == 1 && STRIDE == 1) {
    if (p.im2col) {
        patch_ready = true;
        // conv1_1 (Cin = 3, utils/general.py:36-53 at nets/ColorHandPose3DNetwork.py:144,183): the A tile is
        // built straight from the [B,H,W,3] image as K = (r*3+s)*3+c (27 real + 5 zero) -- no im2col
        // buffer in HBM.  KS == 1 here, so the "patch" is the tile itself.
        // The raw (PH+2) x (PW+2) x 3 image window is staged once in LDS (coalesced loads, one bounds check
        // per element, zero fill = SAME padding); the 32-wide rows are then gathered from LDS.  The
        // third weight-ring slot is free here (one K step only) and serves as the staging area.
        const float* img = p.in + (size_t)b * p.H * p.W * 3;
        float* raw = wbuf + 2 * C::WBUF_FLOATS;
        constexpr int RW = PW + 2, RH = PH + 2;
        static_assert(RH * RW * 3 <= C::WBUF_FLOATS, "raw image window must fit one weight slot");
        for (int i = tid; i < RH * RW * 3; i += NTHR) {
            const int c = i % 3, q = i / 3;
            const int ry = q / RW, rx = q - ry * RW;
            const int iy = gy0 + ry - 1, ix = gx0 + rx - 1;
            float v = 0.f;
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = img[(iy * p.W + ix) * 3 + c];
            raw[i] = v;
        }
        __syncthreads();
        for (int idx = tid; idx < PH * PW * 8; idx += NTHR) {
            const int pix = idx >> 3, c4 = idx & 7;
            const int py = pix / PW, px = pix - py * PW;
            if constexpr (F16) {
                f16x8 v;                                  // 16-B slot c4 = f16 channels 8*c4 .. 8*c4+7 of 64
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = c4 * 8 + e;             // k = (r*3+s)*3 + c
                    v[e] = (k < 27) ? (hp3d_f16)raw[((py + k / 9) * RW + px + (k / 3) % 3) * 3 + k % 3] : (hp3d_f16)0.f;
                }
                *(f16x8*)(patch + pix * LDA + c4 * 4) = v;
            } else {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = c4 * 4 + e;             // k = (r*3+s)*3 + c
                    if (k < 27) v[e] = raw[((py + k / 9) * RW + px + (k / 3) % 3) * 3 + k % 3];
                }
                *(f32x4*)(patch + pix * LDA + c4 * 4) = v;
            }
        }
    }
    }
    if (!patch_ready) {
        patch_fetch(0);
        patch_commit();
    }
    w_dma(0, 0, 0);
    if (total > 1) { if (TAPS == 1) w_dma(0, 1, 1); else w_dma(1, 0, 1); }
    HP3D_WAIT_VMCNT0();
    __syncthreads();
    load_frags(0, tap_off(0), 0, 0);

    if (TAPS % 3 == 0) {
        for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) step(tap, chunk, tap % 3, chunk * TAPS + tap);
        }
    } else {
        int buf = 0, it = 0;
        for (int chunk = 0; chunk < nchunks; ++chunk)
            for (int tap = 0; tap < TAPS; ++tap, ++it) {
                step(tap, chunk, buf, it);
                buf = (buf == 2) ? 0 : buf + 1;
            }
    }

    // ---- epilogue: bias + leaky-ReLU (+ 2x2 max-pool) + NHWC store ------------------------
    const int Hs = POOL ? (p.Ho >> 1) : p.Ho, Ws = POOL ? (p.Wo >> 1) : p.Wo;
    if (p.ksplit > 1) {      // raw partial sums, dense [z][B*Ho*Wo][Cout]
        float* pb = p.partial + ((size_t)kz * p.B + b) * p.Ho * p.Wo * p.Cout;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = n0 + (wn * NT + nt) * 32 + li;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = 2 * (r >> 2) + lh;
                    const int y = oy0 + (wm * MT + mt) * ROWS_PER_MT + 2 * (q / QX) + ((r >> 1) & 1);
                    const int x = ox0 + 2 * (q % QX) + (r & 1);
                    if (y < p.Ho && x < p.Wo) pb[((size_t)y * p.Wo + x) * p.Cout + co] = acc[mt][nt][r];
                }
        }
        return;
    }
    // out_cs / cout_store count OUTPUT ELEMENTS here (f16 elements when the layer stores halves)
    const bool store_h = F16 && !p.out_f32;
    float* outb = p.out + (size_t)b * Hs * Ws * (store_h ? p.out_cs / 2 : p.out_cs);
    hp3d_f16* outh = (hp3d_f16*)outb;
    auto store = [&](size_t idx, float v) {
        if (store_h) outh[idx] = (hp3d_f16)v; else outb[idx] = v;
    };
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = n0 + (wn * NT + nt) * 32 + li;
        const float bias = p.bias[co];
        const bool cok = co < p.cout_store;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int t = wm * MT + mt;
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) {
                const int q = 2 * a4 + lh;               // quad index of registers 4*a4..4*a4+3
                const int qy = q / QX, qx = q % QX;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[mt][nt][a4 * 4 + e] + bias;
                    if (p.act) x = fmaxf(x, HP3D_LEAKY_SLOPE * x);
                    v[e] = x;
                }
                if (POOL) {
                    const int y = (oy0 + t * ROWS_PER_MT + 2 * qy) >> 1, x = (ox0 + 2 * qx) >> 1;
                    const float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                    if (cok && y < Hs && x < Ws) store(((size_t)y * Ws + x) * p.out_cs + co, m);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int y = oy0 + t * ROWS_PER_MT + 2 * qy + (e >> 1), x = ox0 + 2 * qx + (e & 1);
                        if (cok && y < Hs && x < Ws) store(((size_t)y * Ws + x) * p.out_cs + co, v[e]);
                    }
                }
            }
        }
    }
}

// ---- instantiation table -----------------------------------------------------------------
// tile configs: id 0: 8x16 x128  1: 8x16 x64  2: 8x16 x32  3: 8x8 x128  4: 8x8 x64  5: 8x8 x32
template <int KS, int STRIDE, bool POOL, bool F16>
int launch_cfg(const ConvParams& p, int cfg, hipStream_t s) {
#define HP3D_CASE(id, TH, TW, WM, WN, MT, NT)                                                        \
    case id: {                                                                                       \
        using C = ConvCfg<KS, STRIDE, TH, TW, WM, WN, MT, NT>;                                        \
        auto kern = conv_mfma_kernel<KS, STRIDE, TH, TW, WM, WN, MT, NT, POOL, F16>;                       \
        static bool attr_done[64] = {}, attr_done2[64] = {};                                         \
        if (hp3d_first_use_on_device(attr_done))                                                     \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                C::SMEM_BYTES);                                                      \
        dim3 grid(p.B * p.tiles_y * p.tiles_x, p.Cout / C::BN, p.ksplit);                            \
        const ConvParams& pp = p;                                                                    \
        static int extra_lds = getenv("HP3D_CONV_EXTRA_LDS") ? atoi(getenv("HP3D_CONV_EXTRA_LDS")) : 0; \
        if (extra_lds && hp3d_first_use_on_device(attr_done2))                                       \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                C::SMEM_BYTES + extra_lds);                                          \
        HP3D_LAUNCH(kern, grid, dim3(C::NTHR), C::SMEM_BYTES + extra_lds, s, pp);                    \
        return 0;                                                                                    \
    }
    switch (cfg) {
        HP3D_CASE(0, 8, 16, 2, 2, 2, 2)
        HP3D_CASE(1, 8, 16, 2, 2, 2, 1)
        HP3D_CASE(2, 8, 16, 4, 1, 1, 1)
        HP3D_CASE(3, 8, 8, 2, 2, 1, 2)
        HP3D_CASE(4, 8, 8, 2, 2, 1, 1)
        HP3D_CASE(5, 8, 8, 2, 1, 1, 1)
    }
#undef HP3D_CASE
    return -1;
}

const int kCfgTW[6] = {16, 16, 16, 8, 8, 8};
const int kCfgBN[6] = {128, 64, 32, 128, 64, 32};

}  // namespace

int conv_mfma_plan(int k, int stride, int Ho, int Wo, int Cin, int Cout, int pool, int B, ConvPlan* plan) {
    if (!((k == 1 && stride == 1) || (k == 3 && (stride == 1 || stride == 2)) || (k == 7 && stride == 1))) return -1;
    if (Cout % 32 || Cin % 32) return -1;
    if (pool && !(k == 3 && stride == 1)) return -1;
    int bn = (Cout % 128 == 0) ? 128 : (Cout % 64 == 0) ? 64 : 32;
    auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
    // padded pixels computed by each tiling; prefer the wider tile unless it wastes >12% more
    const long px16 = (long)cdiv(Ho, 8) * 8 * cdiv(Wo, 16) * 16;
    const long px8 = (long)cdiv(Ho, 8) * 8 * cdiv(Wo, 8) * 8;
    bool wide = (double)px16 <= 1.12 * (double)px8;
    // under-filled chip (256 CUs, 2 workgroups each): more, smaller workgroups
    const long blocks16 = (long)B * cdiv(Ho, 8) * cdiv(Wo, 16) * (Cout / bn);
    if (blocks16 < 512) wide = false;
    long blocks = (long)B * cdiv(Ho, 8) * cdiv(Wo, wide ? 16 : 8) * (Cout / bn);
    while (blocks < 256 && bn > 32) { bn >>= 1; blocks <<= 1; }       // narrower cout tiles
    int ksplit = 1;
    const int nch = Cin / 32;
    // narrow (bn = 32) workgroups are 2 waves / 26 KB of LDS: ~6 fit on a CU, so aim higher before giving up
    const long want = (bn == 32 && blocks < 768) ? 768 : 256;
    if (!pool && blocks < want && nch > 1) {                           // still under-filled: split K
        ksplit = (int)((want + blocks - 1) / blocks);
        if (ksplit > nch) ksplit = nch;
        if (ksplit > 16) ksplit = 16;
    }
    plan->th = 8;
    plan->tw = wide ? 16 : 8;
    plan->bn = bn;
    plan->ksplit = ksplit;
    plan->variant = (wide ? 0 : 3) + (bn == 128 ? 0 : bn == 64 ? 1 : 2);
    // tuning knob (benchmarks only): HP3D_CONV_CFG=<0..5> forces a tile config when it divides Cout
    static int force = getenv("HP3D_CONV_CFG") ? atoi(getenv("HP3D_CONV_CFG")) : -1;
    if (force >= 0 && force <= 5 && Cout % kCfgBN[force] == 0 && Cout >= 64) {
        plan->variant = force; plan->tw = kCfgTW[force]; plan->bn = kCfgBN[force];
    }
    return 0;
}

int conv_mfma_launch(const ConvParams& p, int k, int stride, int pool, const ConvPlan& plan, hipStream_t s) {
    const int cfg = plan.variant;
    if (cfg < 0 || cfg > 5 || kCfgTW[cfg] != plan.tw || kCfgBN[cfg] != plan.bn) return -1;
    if (p.f16) {      // half-precision trunk layers (no stride-2 layer runs in f16)
        if (k == 1 && stride == 1 && !pool) return launch_cfg<1, 1, false, true>(p, cfg, s);
        if (k == 3 && stride == 1 && !pool) return launch_cfg<3, 1, false, true>(p, cfg, s);
        if (k == 3 && stride == 1 && pool) return launch_cfg<3, 1, true, true>(p, cfg, s);
        if (k == 7 && stride == 1 && !pool) return launch_cfg<7, 1, false, true>(p, cfg, s);
        return -1;
    }
    if (k == 1 && stride == 1 && !pool) return launch_cfg<1, 1, false, false>(p, cfg, s);
    if (k == 3 && stride == 1 && !pool) return launch_cfg<3, 1, false, false>(p, cfg, s);
    if (k == 3 && stride == 1 && pool) return launch_cfg<3, 1, true, false>(p, cfg, s);
    if (k == 3 && stride == 2 && !pool) return launch_cfg<3, 2, false, false>(p, cfg, s);
    if (k == 7 && stride == 1 && !pool) return launch_cfg<7, 1, false, false>(p, cfg, s);
    return -1;
}

const char* conv_mfma_variant_name(int k, int stride, int pool, const ConvPlan& plan) {
    static char buf[6 * 5 * 3][48];
    static const int ks[5] = {1, 3, 3, 7, 3};
    int kid = (k == 1) ? 0 : (k == 3 && stride == 1) ? 1 : (k == 3 && stride == 2) ? 2 : 3;
    (void)ks;
    char* b = buf[(plan.variant * 5 + kid) * 3 + (pool ? 1 : plan.ksplit > 1 ? 2 : 0)];
    snprintf(b, 48, "conv_mfma_k%ds%d_t%dx%d_n%d%s", k, stride, plan.th, plan.tw, plan.bn,
             pool ? "_pool" : plan.ksplit > 1 ? "_splitk" : "");
    return b;
}

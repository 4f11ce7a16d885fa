// conv_wino.hip -- 3x3 / stride-1 convolution by Winograd F(2x2, 3x3) on the f32 matrix cores.
//
// Same call sites as conv_mfma.hip (NetworkOps.conv_relu + max_pool, utils/general.py:36-65) for the layers
// with Cout % 128 == 0: 2.25x fewer multiply-adds than the direct form at float32 (no reduced precision;
// the transforms only add / subtract / halve).
//
//   Y(2x2) = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A          d: 4x4 input window, g: 3x3 filter
//
// Design for gfx950:
//   * workgroup = 8x16 output pixels = 32 Winograd tiles x 128 output channels, 4 waves; wave w owns couts
//     32w..32w+31 and ALL 32 tiles, so the MFMA M dimension is the tile index and nothing about the
//     weights is shared between waves:
//   * transformed weights U[plane 0..15][Cin][Cout] are pre-packed in fragment order and go global -> VGPR
//     directly (buffer_load_dwordx4, scalar per-plane offset): no LDS ring, NO barrier inside a 32-channel
//     chunk -- 16 planes x 16 MFMAs per wave run back to back;
//   * the input transform V = B^T d B is computed by the loader (thread = (tile, channel quad): 16 loads of
//     16 B, 32 add/sub on float4) and written once per chunk to LDS as 16 planes x 32 tiles x 32 channels;
//   * the output transform is folded plane by plane: plane p = (a,b) contributes +-1 * (its 32x32 partial
//     sum) to output (i,j) with coefficient A^T[i][a] * A^T[j][b], so only 4 accumulators (the 2x2 outputs)
//     + 2 rotating plane buffers live in registers (96 instead of 256) -> two workgroups per CU;
//   * the 2x2 outputs of a tile are one pooling window: bias + leaky-ReLU + max-pool stay a register epilogue.
#include "hp3d_common.h"
#include <cstring>

namespace {

constexpr int WCK = 32;            // channels per chunk
constexpr int WLDA = WCK + 4;      // V row pitch in floats (144 B)
constexpr int WTILES = 32;         // Winograd tiles per workgroup: 4 rows x 8 cols of 2x2 outputs
constexpr int V_FLOATS = 16 * WTILES * WLDA;
constexpr int WINO_SMEM_BYTES = V_FLOATS * 4 + 2 * WTILES * 4;     // V + the tile table

template <bool POOL>
HP3D_KERNEL2(256, 2)      // <= 256 registers per lane: two workgroups (two waves per SIMD) per CU
void conv_wino_kernel(const ConvParams p) {
    HP3D_DYN_SMEM(V);
    int* tinfo = (int*)(V + V_FLOATS);     // [0..31] output offset of tile t (-1: no such tile), [32..63] edge flags
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = HP3D_READFIRSTLANE(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.y * 128;

    // The 32 tiles of a workgroup are 32 consecutive entries of the flattened (image, band of 4 tile rows,
    // tile column, row in band) order: a 4x8 footprint where the tile grid allows it, but no padding when it
    // does not (20x20 tiles at 40x40 would waste 20% in fixed 4x8 blocks), and the last workgroup of one image
    // continues into the next.
    const int TXn = p.tiles_x, TYn = p.tiles_y, per_img = TXn * TYn;
    auto tile_decode = [&](int id, int& tb, int& tyy, int& txx) {
        tb = id / per_img;
        const int r = id - tb * per_img;
        const int band = r / (4 * TXn), rem = r - band * 4 * TXn;
        const int rows = min(4, TYn - 4 * band);
        txx = rem / rows;
        tyy = band * 4 + rem - txx * rows;
    };
    const int Hs = POOL ? (p.Ho >> 1) : p.Ho, Ws = POOL ? (p.Wo >> 1) : p.Wo;
    if (tid < WTILES) {
        int tb, tyy, txx;
        tile_decode(blockIdx.x * WTILES + tid, tb, tyy, txx);
        int off = -1, fl = 0;
        if (tb < p.B) {
            if (POOL) {
                if (tyy < Hs && txx < Ws) off = ((tb * Hs + tyy) * Ws + txx) * p.out_cs;
            } else {
                off = ((tb * Hs + 2 * tyy) * Ws + 2 * txx) * p.out_cs;
                fl = (2 * txx + 1 < Ws ? 1 : 0) | (2 * tyy + 1 < Hs ? 2 : 0);
            }
        }
        tinfo[tid] = off;
        tinfo[WTILES + tid] = fl;
    }

    // ---- loader role: this thread transforms the 4x4 window of tile lt for channel quad lc ----------
    // (buffer loads: a window element outside the image gets an out-of-range offset and reads as 0)
    const int lt = tid >> 3, lc = tid & 7;
    int lb, lty, ltx;
    tile_decode(blockIdx.x * WTILES + lt, lb, lty, ltx);
    const int wy0 = 2 * lty - 1, wx0 = 2 * ltx - 1;                              // SAME padding 1
    const int cs4 = p.in_cs * 4;
    const int wbase = ((lb * p.H + wy0) * p.W + wx0) * cs4 + lc * 16;           // bytes
    unsigned wmask = 0;       // bit r*4+c: window element inside the image
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (lb < p.B && (unsigned)(wy0 + r) < (unsigned)p.H && (unsigned)(wx0 + c) < (unsigned)p.W) wmask |= 1u << (r * 4 + c);
    const hp3d_rsrc_t irsrc = HP3D_MAKE_RSRC(p.in, (unsigned)p.B * (unsigned)(p.H * p.W) * (unsigned)cs4);
    constexpr int OOR = (int)0x80000000;

    f32x4 d[16];
    // one window row (4 loads); wm = wmask, or 0 when there is no such chunk
    auto window_fetch_row = [&](int r, unsigned wm, int soff) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int voff = (wm & (1u << (r * 4 + c))) ? wbase + (r * p.W + c) * cs4 : OOR;
            d[r * 4 + c] = HP3D_BUFFER_LOAD16(irsrc, voff, soff);
        }
    };
    auto transform_commit = [&]() {
        // B^T d B with B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
        f32x4 t[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            t[0 * 4 + c] = d[0 * 4 + c] - d[2 * 4 + c];
            t[1 * 4 + c] = d[1 * 4 + c] + d[2 * 4 + c];
            t[2 * 4 + c] = d[2 * 4 + c] - d[1 * 4 + c];
            t[3 * 4 + c] = d[1 * 4 + c] - d[3 * 4 + c];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4 v0 = t[r * 4 + 0] - t[r * 4 + 2];
            const f32x4 v1 = t[r * 4 + 1] + t[r * 4 + 2];
            const f32x4 v2 = t[r * 4 + 2] - t[r * 4 + 1];
            const f32x4 v3 = t[r * 4 + 1] - t[r * 4 + 3];
            *(f32x4*)(V + ((r * 4 + 0) * WTILES + lt) * WLDA + lc * 4) = v0;
            *(f32x4*)(V + ((r * 4 + 1) * WTILES + lt) * WLDA + lc * 4) = v1;
            *(f32x4*)(V + ((r * 4 + 2) * WTILES + lt) * WLDA + lc * 4) = v2;
            *(f32x4*)(V + ((r * 4 + 3) * WTILES + lt) * WLDA + lc * 4) = v3;
        }
    };

    // ---- MFMA role ----------------------------------------------------------------------------------
    // packed U: [plane 16][chunk][Cout/32][g 4][h 2][n 32][j 4] -> the 4 fragments a wave needs for one
    // (plane, chunk) are 4 KB contiguous: base = one scalar offset, g = an immediate
    const int CO32 = p.Cout >> 5;
    const int nchunks = p.Cin / WCK;
    const hp3d_rsrc_t wrsrc = HP3D_MAKE_RSRC(p.wpk, (unsigned)(16 * p.Cin) * (unsigned)p.Cout * 4u);
    const int chunk_stride_b = CO32 * 4096;                 // bytes between chunks
    const int plane_stride_b = nchunks * chunk_stride_b;    // bytes between planes
    const int wvoff = ((n0 >> 5) + wave) * 4096 + lane * 16;
    const int abase = li * WLDA + lh * 4;

    f32x16 y[4], tmp[2];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[o][r] = 0.f;

    f32x4 bq[4][4];        // B fragments of four planes in flight (global -> VGPR): prefetch distance 3 planes
    auto b_fetch = [&](int set, int soff) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[set][g] = HP3D_BUFFER_LOAD16(wrsrc, wvoff + g * 1024, soff);
    };
    // fold a finished plane (a,b) into the 2x2 outputs: coefficient A^T[i][a] * A^T[j][b], A^T = [1 1 1 0; 0 1 -1 -1]
    // (plane is wave-uniform at run time: the zero coefficients are skipped by scalar branches)
    auto fold = [&](int plane, const f32x16& m) {
        const int a = plane >> 2, bb = plane & 3;
        const int ca0 = a < 3 ? 1 : 0, ca1 = a == 0 ? 0 : (a == 1 ? 1 : -1);
        const int cb0 = bb < 3 ? 1 : 0, cb1 = bb == 0 ? 0 : (bb == 1 ? 1 : -1);
        const int c00 = ca0 * cb0, c01 = ca0 * cb1, c10 = ca1 * cb0, c11 = ca1 * cb1;
        if (c00) y[0] += m;                                   // c00 is 0 or +1
        if (c01 > 0) y[1] += m; else if (c01 < 0) y[1] -= m;
        if (c10 > 0) y[2] += m; else if (c10 < 0) y[2] -= m;
        if (c11 > 0) y[3] += m; else if (c11 < 0) y[3] -= m;
    };
    // one plane: 16 MFMAs from fragment set u, then the previous plane's fold (VALU under these MFMAs)
    auto plane_mma = [&](int plane, int u) {
        f32x4 af[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) af[g] = *(const f32x4*)(V + plane * (WTILES * WLDA) + abase + g * 8);
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        tmp[u & 1] = HP3D_MFMA_32x32x2(af[0][0], bq[u][0][0], zero);
#pragma unroll
        for (int gj = 1; gj < 16; ++gj)
            tmp[u & 1] = HP3D_MFMA_32x32x2(af[gj >> 2][gj & 3], bq[u][gj >> 2][gj & 3], tmp[u & 1]);
        if (plane > 0) fold(plane - 1, tmp[(u & 1) ^ 1]);
    };
    // soff(it) for the flattened plane counter it = chunk*16 + plane
    auto soff_of = [&](int it) { return (it & 15) * plane_stride_b + (it >> 4) * chunk_stride_b; };

#pragma unroll
    for (int r = 0; r < 4; ++r) window_fetch_row(r, wmask, 0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __syncthreads();                 // everyone finished reading V of the previous chunk (and tinfo is written)
        transform_commit();
        __syncthreads();
        // the weight prefetch does not cross the chunk boundary: window (64) + transform temporaries
        // + 3 planes of fragments (48) + the accumulators (96) would not fit 256 registers
        HP3D_SCHED_BARRIER();
        b_fetch(0, soff_of(chunk * 16));
        b_fetch(1, soff_of(chunk * 16 + 1));
        b_fetch(2, soff_of(chunk * 16 + 2));
        for (int pq = 0; pq < 3; ++pq) {          // four planes per trip: B-fragment sets and tmp sets stay static
            const int it0 = chunk * 16 + pq * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                HP3D_SCHED_BARRIER();
                b_fetch((u + 3) & 3, soff_of(it0 + u + 3));     // 3 planes ahead
                plane_mma(pq * 4 + u, u);
            }
        }
        // planes 12..15, peeled: the fragment sets drain one per plane, and the registers they free take the
        // NEXT chunk's 4x4 window one row per plane -- its latency hides under ~3 planes of MFMAs instead of
        // being exposed at the chunk boundary
        const unsigned wmn = chunk + 1 < nchunks ? wmask : 0u;
        const int wsoff = (chunk + 1) * (WCK * 4);
        HP3D_SCHED_BARRIER();
        b_fetch(3, soff_of(chunk * 16 + 15));
        plane_mma(12, 0);
        HP3D_SCHED_BARRIER();
        window_fetch_row(0, wmn, wsoff);
        plane_mma(13, 1);
        HP3D_SCHED_BARRIER();
        window_fetch_row(1, wmn, wsoff);
        plane_mma(14, 2);
        HP3D_SCHED_BARRIER();
        window_fetch_row(2, wmn, wsoff);
        plane_mma(15, 3);
        HP3D_SCHED_BARRIER();
        window_fetch_row(3, wmn, wsoff);
        fold(15, tmp[1]);
    }

    // ---- epilogue: bias + leaky-ReLU (+ 2x2 max-pool) + NHWC store ----------------------------------------
    const int co = n0 + wave * 32 + li;
    const float bias = p.bias[co];
    const bool cok = co < p.cout_store;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = (r & 3) + 8 * (r >> 2) + 4 * lh;      // MFMA row = Winograd tile
        const int off = tinfo[t], fl = tinfo[WTILES + t];
        float v[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float x = y[o][r] + bias;
            if (p.act) x = fmaxf(x, HP3D_LEAKY_SLOPE * x);
            v[o] = x;
        }
        if (!cok || off < 0) continue;
        float* o0 = p.out + off + co;
        if (POOL) {
            *o0 = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        } else {
            o0[0] = v[0];
            if (fl & 1) o0[p.out_cs] = v[1];
            if (fl & 2) o0[(size_t)Ws * p.out_cs] = v[2];
            if (fl == 3) o0[(size_t)(Ws + 1) * p.out_cs] = v[3];
        }
    }
}

}  // namespace

// U = G g G^T per (cin, cout), packed for the kernel: [plane][chunk][Cout/32][g][h][n][j] with channel
// 32*chunk + 8*g + 4*h + j and cout 32*co32 + n (zero padded to cin_pad x cout_pad)
void wino_pack_weights(const float* g_hwio /*[3][3][Cin][Cout]*/, int Cin, int Cout, int cin_pad, int cout_pad, float* dst) {
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const int nch = cin_pad / 32, CO32 = cout_pad / 32;
    memset(dst, 0, sizeof(float) * (size_t)16 * cin_pad * cout_pad);
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b)
            for (int ci = 0; ci < Cin; ++ci) {
                const int chunk = ci >> 5, g = (ci >> 3) & 3, h = (ci >> 2) & 1, j = ci & 3;
                for (int co = 0; co < Cout; ++co) {
                    float s = 0.f;
                    for (int r = 0; r < 3; ++r)
                        for (int c = 0; c < 3; ++c) s += G[a][r] * g_hwio[((size_t)(r * 3 + c) * Cin + ci) * Cout + co] * G[b][c];
                    dst[(((((size_t)(a * 4 + b) * nch + chunk) * CO32 + (co >> 5)) * 4 + g) * 2 + h) * 128 + (co & 31) * 4 + j] = s;
                }
            }
}

// mode 1 (auto): only when the grid fills the chip (small problems stay on the direct kernel's small-batch
// plan); mode 2 (forced, tests): whenever the shape allows
int conv_wino_eligible(int mode, int k, int stride, int Cin, int Cout, int Ho, int Wo, int B) {
    if (mode == 0 || k != 3 || stride != 1 || Cin % 32 || Cout % 128) return 0;
    // the kernel addresses both tensors with 32-bit offsets (channel strides up to 2x the channel count)
    if ((long)B * Ho * Wo * (Cin > Cout ? Cin : Cout) * 8 >= (1L << 31)) return 0;
    const long tiles = (long)B * ((Ho + 1) / 2) * ((Wo + 1) / 2);
    const long blocks = (tiles + 31) / 32 * (Cout / 128);
    return mode == 2 || blocks >= 256;
}

int conv_wino_launch(const ConvParams& pin, int pool, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_wino_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WINO_SMEM_BYTES);
        (void)hipFuncSetAttribute((const void*)conv_wino_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WINO_SMEM_BYTES);
        attr_done = true;
    }
    // 32-bit byte / element offsets inside the kernel (buffer loads, the tile table)
    if ((long)pin.B * pin.H * pin.W * pin.in_cs * 4 >= (1L << 31) || (long)pin.B * pin.Ho * pin.Wo * pin.out_cs >= (1L << 31)) return -1;
    ConvParams p = pin;
    p.tiles_x = (p.Wo + 1) / 2;          // Winograd tiles per row / column
    p.tiles_y = (p.Ho + 1) / 2;
    const long tiles = (long)p.B * p.tiles_x * p.tiles_y;
    dim3 grid((unsigned)((tiles + WTILES - 1) / WTILES), p.Cout / 128);
    if (pool) {
        auto k = conv_wino_kernel<true>;
        HP3D_LAUNCH(k, grid, dim3(256), WINO_SMEM_BYTES, s, p);
    } else {
        auto k = conv_wino_kernel<false>;
        HP3D_LAUNCH(k, grid, dim3(256), WINO_SMEM_BYTES, s, p);
    }
    return 0;
}

// conv_wino2.hip -- Winograd F(2x2, 3x3) on the f32 matrix cores, TWO workgroups per CU (round 3).
//
// Same call sites, same arithmetic and the same results to rounding as conv_wino.hip (NetworkOps.conv_relu + max_pool,
// utils/general.py:36-65; 3x3 / stride-1 layers and the 7x7 layers as nine 3x3 blocks), a different machine shape.
// conv_wino.hip gives one wave a whole SIMD: 32 tiles x 32 couts x 16 planes = 256 accumulators on v_mfma_f32_32x32x2_f32.  That
// is the right shape for long reductions, but everything an item does besides MFMAs -- the output transform, 64 stores per lane
// that sit in the in-order vmcnt queue in front of the next weight fragments, the item set-up, the window latency of its first
// steps, the barrier -- runs with the matrix core of its SIMD idle, and one step of one item is 7 us.  Measured (profiles/
// r02_sq_counters.md): matrix cores 0.58-0.62 busy on the Cin = 64 layers (2-4 steps per item), and at batch 1 a launch takes
// ~25 us whatever it computes, because a 32 x 32 feature map is 8-16 items on 256 CUs.
//
// Here a wave holds HALF as many accumulators (v_mfma_f32_16x16x4_f32: 32 tiles x 16 couts x 16 planes = 128 registers), a
// workgroup of four waves is 32 tiles x 64 couts with a 2 x 32 KB transformed-input buffer, and TWO workgroups share a CU
// (__launch_bounds__(256, 2): 256 registers per lane, 66 KB of LDS each).  The two waves of a SIMD belong to different work
// items in different phases: one wave's epilogue, store drain, barrier or window wait is the other one's MFMA time.  An item-step is
// 16 channels = 128 MFMAs of 32 cycles per wave (1.8 us): four times finer than conv_wino's, which is what small batches need.
//   * loader role: thread = (tile, channel PAIR): sixteen 8-byte window loads per step, B^T d B in place, 8-byte LDS writes;
//   * A fragments: one ds_read_b128 per (plane, tile half): lane (t, q) holds channels 4q..4q+3 of tile t -> the k operands
//     of four MFMAs; the 16-byte quads of a tile row are XOR-swizzled by a function of the tile chosen for the hardware's real
//     ds_read_b128 lane groups (conflict-free; MI355X_MICROARCH.md "LDS");
//   * B fragments: transformed filters packed [plane][16-channel step][Cout/16][q][n][e], one 16-byte buffer load per plane and
//     step straight into VGPRs, ring of 8 planes (4 registers each);
//   * everything else as in conv_wino.hip: persistent grid, static round-robin items, next item's windows / fragments fetched
//     under the current item's last step, tile table in LDS, branch-free epilogue with out-of-range offsets for invalid lanes,
//     split-K over channel steps for under-filled launches (raw partial sums + conv_splitk_reduce), 7x7 filters as nine 3x3
//     blocks with the structurally zero planes of the edge blocks skipped.
#include "hp3d_common.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace {

constexpr int W2_TILES = 32;                       // Winograd tiles per item
constexpr int W2_CK = 16;                          // channels per step
constexpr int W2_COUTS = 64;                       // output channels per item (16 per wave)
constexpr int W2_PLANE_FLOATS = W2_TILES * W2_CK;  // one plane of one V buffer: 2 KB
constexpr int W2_VBUF_FLOATS = 16 * W2_PLANE_FLOATS;
constexpr int W2_SMEM_BYTES = 2 * W2_VBUF_FLOATS * 4 + 2 * 2 * W2_TILES * 4;     // 2 V buffers + two tile tables = 66048 B
constexpr int W2_RING = 8;                         // weight fragments in flight (planes)

// quad swizzle of tile t's V row: f((t >> 2) & 3) with f = {0, 2, 3, 1}.  ds_read_b128 is served in the lane groups
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32); with lane = 16 q + t reading quad q ^ f of tile t, the sixteen lanes of every
// group hit sixteen different 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int w2_swz(int t) { return (0x78 >> (((t >> 2) & 3) * 2)) & 3; }

template <bool POOL, int NSUB, bool SPLITK>
HP3D_KERNEL2(256, 2)
void conv_wino2_kernel(const ConvParams p) {
    static_assert(!(POOL && SPLITK), "the fused max-pool needs complete sums");
    HP3D_DYN_SMEM(V);
    int* tinfo = (int*)(V + 2 * W2_VBUF_FLOATS);       // [parity][0..31] output offset of tile t (-1: none), [32..63] edge flags
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = HP3D_READFIRSTLANE(tid >> 6);
    const int ln = lane & 15, lq = lane >> 4;          // MFMA column (cout / tile in block) and k slot

    const int TXn = p.tiles_x, TYn = p.tiles_y, per_img = TXn * TYn;
    const int tile_blocks = (p.B * per_img + W2_TILES - 1) / W2_TILES;
    const int ncy = p.Cout / W2_COUTS;
    const int per_split = tile_blocks * ncy;
    const int nitems = per_split * (SPLITK ? p.ksplit : 1);
    auto tile_decode = [&](int id, int& tb, int& tyy, int& txx) {      // same flattened band order as conv_wino.hip
        tb = id / per_img;
        const int r = id - tb * per_img;
        const int band = r / (4 * TXn), rem = r - band * 4 * TXn;
        const int rows = min(4, TYn - 4 * band);
        txx = rem / rows;
        tyy = band * 4 + rem - txx * rows;
    };
    const int Hs = POOL ? (p.Ho >> 1) : p.Ho, Ws = POOL ? (p.Wo >> 1) : p.Wo;
    auto table_write = [&](int tblock, int parity, int kz) {
        if (tid < W2_TILES) {
            int tb, tyy, txx;
            tile_decode(tblock * W2_TILES + tid, tb, tyy, txx);
            int off = -1, fl = 0;
            if (tb < p.B) {
                if (POOL) {
                    if (tyy < Hs && txx < Ws) off = ((tb * Hs + tyy) * Ws + txx) * p.out_cs;
                } else {
                    off = (((SPLITK ? kz * p.B + tb : tb) * Hs + 2 * tyy) * Ws + 2 * txx) * p.out_cs;
                    fl = (2 * txx + 1 < Ws ? 1 : 0) | (2 * tyy + 1 < Hs ? 2 : 0);
                }
            }
            tinfo[parity * 2 * W2_TILES + tid] = off;
            tinfo[parity * 2 * W2_TILES + W2_TILES + tid] = fl;
        }
    };

    // ---- loader role: this thread transforms the 4x4 window of tile lt for channel pair lp ------------------------------
    const int lt = tid >> 3, lp = tid & 7;
    const int cs4 = p.in_cs * 4;
    constexpr int OOR = (int)0x80000000;
    int wv[16];
    int cb = 0, cty = 0, ctx_ = 0;            // NSUB = 9 only: tile coordinates stay live for the block shifts
    auto window_offsets = [&](bool valid, int lb, int lty, int ltx, int sub) {
        const int dy = NSUB == 1 ? 0 : 3 * (sub / 3) - 2, dx = NSUB == 1 ? 0 : 3 * (sub % 3) - 2;
        const int wy0 = 2 * lty - 1 + dy, wx0 = 2 * ltx - 1 + dx;
        const int wbase = ((lb * p.H + wy0) * p.W + wx0) * cs4 + lp * 8;
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
        tile_decode(tblock * W2_TILES + lt, lb, lty, ltx);
        if (NSUB > 1) { cb = valid ? lb : p.B; cty = lty; ctx_ = ltx; }
        window_offsets(valid, lb, lty, ltx, NSUB > 1 ? sub : 0);
    };
    auto loader_shift = [&](int sub) { window_offsets(true, cb, cty, ctx_, sub); };
    const hp3d_rsrc_t irsrc = HP3D_MAKE_RSRC(p.in, (unsigned)p.B * (unsigned)(p.H * p.W) * (unsigned)cs4);
    const hp3d_rsrc_t orsrc = HP3D_MAKE_RSRC(p.out, (unsigned)(SPLITK ? p.ksplit * p.B : p.B) * (unsigned)(Hs * Ws) * (unsigned)p.out_cs * 4u);

    f32x2 d[16];
    auto window_fetch = [&](int soff) {
#pragma unroll
        for (int e = 0; e < 16; ++e) d[e] = HP3D_BUFFER_LOAD8(irsrc, wv[e], soff);
    };
    float* const Vw = V + lt * W2_CK + ((lp >> 1) ^ w2_swz(lt)) * 4 + (lp & 1) * 2;      // this thread's slot in plane 0 of buffer 0
    auto transform_commit = [&](int buf) {
        // B^T d B with B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1], in place: rows first, then columns straight into LDS
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x2 a0 = d[0 * 4 + c], a1 = d[1 * 4 + c], a2 = d[2 * 4 + c], a3 = d[3 * 4 + c];
            d[0 * 4 + c] = a0 - a2;
            d[1 * 4 + c] = a1 + a2;
            d[2 * 4 + c] = a2 - a1;
            d[3 * 4 + c] = a1 - a3;
        }
        float* Vq = Vw + buf * W2_VBUF_FLOATS;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x2 t0 = d[r * 4 + 0], t1 = d[r * 4 + 1], t2 = d[r * 4 + 2], t3 = d[r * 4 + 3];
            *(f32x2*)(Vq + (r * 4 + 0) * W2_PLANE_FLOATS) = t0 - t2;
            *(f32x2*)(Vq + (r * 4 + 1) * W2_PLANE_FLOATS) = t1 + t2;
            *(f32x2*)(Vq + (r * 4 + 2) * W2_PLANE_FLOATS) = t2 - t1;
            *(f32x2*)(Vq + (r * 4 + 3) * W2_PLANE_FLOATS) = t1 - t3;
        }
    };

    // ---- MFMA role -------------------------------------------------------------------------------------------------------
    // packed U: [plane 16][step (NSUB * Cin / 16)][Cout/16][q 4][n 16][e 4]: the fragment a wave needs for one (plane, step) is
    // 1 KB, lane-linear: base = one scalar offset per (plane, step)
    const int CO16 = p.Cout >> 4;
    const int nsub_rt = NSUB == 1 ? 1 : p.nsub;
    const int csteps = p.Cin / W2_CK;
    const int nsteps = nsub_rt * csteps;
    const hp3d_rsrc_t wrsrc = HP3D_MAKE_RSRC(p.wpk, (unsigned)(16 * nsub_rt * p.Cin) * (unsigned)p.Cout * 4u);
    const int step_stride_b = CO16 * 1024;
    const int plane_stride_b = nsteps * step_stride_b;
    auto soff_of = [&](int plane, int step) { return plane * plane_stride_b + step * step_stride_b; };

    f32x4 M[16][2];        // [plane][tile half]: rows = tiles 16 m + 4 (lane >> 4) + r, column = cout (lane & 15)
    f32x4 bq[W2_RING];
    auto b_fetch = [&](int slot, int voff, int soff) { bq[slot] = HP3D_BUFFER_LOAD16(wrsrc, voff, soff); };
    // A fragments: lane (t = ln, q = lq) reads quad q ^ swz of tile 16 m + t; the swizzle does not depend on m
    const int va_lane = (ln * W2_CK + ((lq ^ w2_swz(ln)) * 4)) * 4;
    int ab = 0;
    f32x4 af[2][2];
    auto a_fetch = [&](int set, int plane) {
#pragma unroll
        for (int m = 0; m < 2; ++m) af[set][m] = *(const f32x4*)((const char*)V + ab + (plane * W2_PLANE_FLOATS + m * 16 * W2_CK) * 4);
    };

    auto split_of = [&](int it, int& kz, int& cy_, int& tb_) {
        kz = SPLITK ? it / per_split : 0;
        const int r = SPLITK ? it - kz * per_split : it;
        cy_ = r / tile_blocks;
        tb_ = r - cy_ * tile_blocks;
    };
    auto first_step_of = [&](int kz) { return SPLITK ? HP3D_READFIRSTLANE((kz * nsteps) / p.ksplit) : 0; };
    int item = blockIdx.x;
    int kz, cy, tblock;
    split_of(item, kz, cy, tblock);
    int s0 = first_step_of(kz), s1 = SPLITK ? first_step_of(kz + 1) : nsteps;
    const int sub0 = (NSUB > 1 && SPLITK) ? HP3D_READFIRSTLANE(s0 / csteps) : 0;
    int sub_cur = sub0;
    loader_setup(tblock, true, sub0);
    table_write(tblock, 0, kz);
    int wvoff = (cy * (W2_COUTS / 16) + wave) * 1024 + lane * 16;
    window_fetch((s0 - sub0 * csteps) * (W2_CK * 4));
#pragma unroll
    for (int t = 0; t < W2_RING; ++t) b_fetch(t, wvoff, soff_of(t, s0));
    transform_commit(0);
    __syncthreads();
    int cur = 0;

    for (int k = 0;; ++k) {
        int n_cy = cy, n_tblock = tblock, n_wvoff = wvoff, n_kz = kz, n_s0 = s0;
        const int n_item = item + (int)gridDim.x;
        const int co = cy * W2_COUTS + wave * 16 + ln;
        const float bias = SPLITK ? 0.f : p.bias[co];

        auto step_body = [&](int step, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const bool lasts = step + 1 == s1;
            // 7x7 filters: planes that are structurally zero in the edge blocks of the 9x9 extension (conv_wino.hip) are skipped
            const bool za = sub_cur >= 6, zb = sub_cur == 2 || sub_cur == 5 || sub_cur == 8;
            const int skip_a = (NSUB == 9 && !FIRST) ? HP3D_OPAQUE_SGPR(za ? 1 : 0) : 0;
            const int skip_b = (NSUB == 9 && !FIRST) ? HP3D_OPAQUE_SGPR(zb ? 1 : 0) : 0;
            const int skip_ab = (NSUB == 9 && !FIRST) ? HP3D_OPAQUE_SGPR((za || zb) ? 1 : 0) : 0;
            const int nvoff = lasts ? n_wvoff : wvoff;
            const int nstep = lasts ? (SPLITK ? n_s0 : 0) : step + 1;
            ab = cur * (W2_VBUF_FLOATS * 4) + va_lane;
            HP3D_OPAQUE_V(ab);
            a_fetch(0, 0);
            const int nsub_ = NSUB == 1 ? 0 : SPLITK ? HP3D_READFIRSTLANE(nstep / csteps) : lasts ? 0 : (step + 1) / csteps;
            const int ncs = NSUB == 1 ? nstep : nstep - nsub_ * csteps;
            if (lasts) loader_setup(n_tblock, n_item < nitems, nsub_);
            else if (NSUB > 1 && ncs == 0) loader_shift(nsub_);
#pragma unroll
            for (int pl = 0; pl < 16; ++pl) {
                HP3D_SCHED_BARRIER();
                if (pl < 15) a_fetch((pl & 1) ^ 1, pl + 1);
                const bool zplane = NSUB == 9 && !FIRST && ((pl >> 2) == 3 || (pl & 3) == 3);
                if (zplane) {
                    const int skip = pl == 15 ? skip_ab : (pl >> 2) == 3 ? skip_a : skip_b;
                    HP3D_MFMA16_2x4_UNLESS(M[pl][0], M[pl][1], af[pl & 1][0], af[pl & 1][1], bq[pl & (W2_RING - 1)], skip);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int m = 0; m < 2; ++m) {       // the two tile halves alternate: 40-cycle dependent latency vs 32-cycle issue
                            if (FIRST && e == 0) {
                                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                                M[pl][m] = HP3D_MFMA_16x16x4(af[pl & 1][m][e], bq[pl & (W2_RING - 1)][e], zero);
                            } else {
                                M[pl][m] = HP3D_MFMA_16x16x4(af[pl & 1][m][e], bq[pl & (W2_RING - 1)][e], M[pl][m]);
                            }
                        }
                }
                if (pl == 0) window_fetch(NSUB > 1 ? HP3D_READFIRSTLANE(ncs * (W2_CK * 4)) : ncs * (W2_CK * 4));
                // weight prefetch W2_RING planes ahead into the slot this plane just released
                const int t = pl + W2_RING;
                if (t < 16) b_fetch(t & (W2_RING - 1), wvoff, soff_of(t, step));
                else b_fetch(t & (W2_RING - 1), nvoff, soff_of(t - 16, nstep));
                if (pl == 12) transform_commit(cur ^ 1);
            }
            HP3D_SCHED_BARRIER();
            __syncthreads();             // V[cur^1] complete, V[cur] free
            cur ^= 1;
            sub_cur = nsub_;
        };
        step_body(s0, std::true_type{});
        {
            const bool has_next = n_item < nitems;
            if (has_next) split_of(n_item, n_kz, n_cy, n_tblock);
            if (SPLITK) { n_kz = HP3D_READFIRSTLANE(n_kz); n_cy = HP3D_READFIRSTLANE(n_cy); n_tblock = HP3D_READFIRSTLANE(n_tblock); }
            n_s0 = first_step_of(n_kz);
            table_write(n_tblock, (k + 1) & 1, n_kz);
            n_wvoff = (n_cy * (W2_COUTS / 16) + wave) * 1024 + lane * 16;
        }
        for (int step = s0 + 1; step < s1; ++step) step_body(step, std::false_type{});

        // ---- epilogue: Y = A^T M A (A^T = [1 1 1 0; 0 1 -1 -1]), bias + leaky-ReLU (+ 2x2 max-pool) + NHWC store; a channel
        //      split stores its raw sums into the [ksplit][B*Ho*Wo][Cout] scratch and conv_splitk_reduce adds them up in split order.
        //      (A reduction INSIDE the launch -- write-through slabs, arrival ticket, last arriver sums -- was built and measured in
        //      round 3: slower than the extra launch at batch 1 and not reliably coherent across XCDs; profiles/r03_tuning_notes.md.)
        const int* tab = tinfo + (k & 1) * 2 * W2_TILES;
        const bool cok = co < p.cout_store;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            int vo[4], fl[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = 16 * m + 4 * lq + r;          // MFMA row = Winograd tile
                const int off = tab[t];
                fl[r] = POOL ? 0 : tab[W2_TILES + t];
                vo[r] = (cok && off >= 0) ? (off + co) * 4 : OOR;
            }
            float y[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float c0 = M[a * 4 + 0][m][r] + M[a * 4 + 1][m][r] + M[a * 4 + 2][m][r];
                    const float c1 = M[a * 4 + 1][m][r] - M[a * 4 + 2][m][r] - M[a * 4 + 3][m][r];
                    if (a == 0) { y[0][r] = c0; y[1][r] = c1; }
                    else if (a == 1) { y[0][r] += c0; y[1][r] += c1; y[2][r] = c0; y[3][r] = c1; }
                    else if (a == 2) { y[0][r] += c0; y[1][r] += c1; y[2][r] -= c0; y[3][r] -= c1; }
                    else { y[2][r] -= c0; y[3][r] -= c1; }
                }
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    float x = y[o][r] + bias;
                    if (!SPLITK && p.act) x = fmaxf(x, HP3D_LEAKY_SLOPE * x);
                    y[o][r] = x;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (POOL) {
                    HP3D_BUFFER_STORE4(orsrc, fmaxf(fmaxf(y[0][r], y[1][r]), fmaxf(y[2][r], y[3][r])), vo[r], 0);
                } else {
                    HP3D_BUFFER_STORE4(orsrc, y[0][r], vo[r], 0);
                    HP3D_BUFFER_STORE4(orsrc, y[1][r], (fl[r] & 1) ? vo[r] : OOR, p.out_cs * 4);
                    HP3D_BUFFER_STORE4(orsrc, y[2][r], (fl[r] & 2) ? vo[r] : OOR, Ws * p.out_cs * 4);
                    HP3D_BUFFER_STORE4(orsrc, y[3][r], fl[r] == 3 ? vo[r] : OOR, (Ws + 1) * p.out_cs * 4);
                }
            }
        }
        if (n_item >= nitems) break;
        item = n_item; cy = n_cy; tblock = n_tblock; wvoff = n_wvoff;
        if (SPLITK) { kz = n_kz; s0 = n_s0; s1 = first_step_of(n_kz + 1); }
    }
}

}  // namespace

// U = G g G^T per (virtual cin, cout) in conv_wino2's fragment order: [plane][step = vc / 16][Cout/16][q][n][e] with virtual channel
// vc = 16 step + 4 q + e and cout 16 co16 + n (zero padded).  Virtual channels as in wino_pack_weights (k = 7: nine 3x3 blocks of the
// filter zero-extended to 9x9, vc = (3i + j) * cin_pad + engine channel).
void wino2_pack_weights(const float* g_hwio, int k, int Cin, int Cout, int cin_pad, int cout_pad, const int* chan_map, float* dst) {
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const int nsub = k == 7 ? 9 : 1;
    const int nst = nsub * cin_pad / 16, CO16 = cout_pad / 16;
    memset(dst, 0, sizeof(float) * wino_packed_floats(k, cin_pad, cout_pad));
    for (int sub = 0; sub < nsub; ++sub) {
        const int u0 = k == 7 ? 3 * (sub / 3) : 0, v0 = k == 7 ? 3 * (sub % 3) : 0;
        for (int e_ = 0; e_ < cin_pad; ++e_) {
            const int rc = chan_map ? chan_map[e_] : (e_ < Cin ? e_ : -1);
            if (rc < 0) continue;
            const int vc = sub * cin_pad + e_;
            const int st = vc >> 4, q = (vc >> 2) & 3, e = vc & 3;
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
                        dst[((((size_t)(a * 4 + b) * nst + st) * CO16 + (co >> 4)) * 4 + q) * 64 + (co & 15) * 4 + e] = s;
                    }
            }
        }
    }
}

// Returns 1 when the layer can run here; *ksplit (may be NULL) receives the channel split that fills the chip (2 workgroups per CU).
int conv_wino2_eligible(int k, int stride, int Cin, int Cout, int Ho, int Wo, int B, int in_cs, int out_cs, int pool, int* ksplit) {
    if (ksplit) *ksplit = 1;
    if ((k != 3 && k != 7) || stride != 1 || Cin % 16 || Cout % W2_COUTS) return 0;
    if ((long)B * Ho * Wo * in_cs * 4 >= (1L << 31) || (long)B * Ho * Wo * out_cs * 4 >= (1L << 31)) return 0;
    if (pool && k != 3) return 0;
    const long tiles = (long)B * ((Ho + 1) / 2) * ((Wo + 1) / 2);
    const long items = (tiles + W2_TILES - 1) / W2_TILES * (Cout / W2_COUTS);
    const int slots = 2 * hp3d_num_cus();
    if (items >= slots || !ksplit) return 1;
    const int nsteps = (k == 7 ? 9 : 1) * Cin / W2_CK;
    int ks = (int)(slots / items);                  // one round: items * ks workgroups all resident at once
    if (ks > nsteps / 2) ks = nsteps / 2;          // at least two steps per item: the first one carries the cold prologue
    if (ks > 32) ks = 32;
    if (ks >= 2 && !(pool && ((Ho | Wo) & 1)) && (long)ks * B * Ho * Wo * Cout * 4 < (1L << 31)) *ksplit = ks;
    return 1;
}

template <bool POOL, int NSUB, bool SPLITK>
static void wino2_launch_t(const ConvParams& p, long tiles, hipStream_t s) {
    static bool attr_done[64] = {};
    auto k = conv_wino2_kernel<POOL, NSUB, SPLITK>;
    if (hp3d_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, W2_SMEM_BYTES);
    const long items = (tiles + W2_TILES - 1) / W2_TILES * (p.Cout / W2_COUTS) * (SPLITK ? p.ksplit : 1);
    const int slots = 2 * hp3d_num_cus();                 // persistent grid: two workgroups per CU
    dim3 grid((unsigned)(items < slots ? items : slots));
    HP3D_LAUNCH(k, grid, dim3(256), W2_SMEM_BYTES, s, p);
}

// pin.ksplit > 1: pin.out must be the partial-sum scratch [ksplit][B*Ho*Wo][Cout] with out_cs = cout_store = Cout; the caller runs
// conv_splitk_reduce afterwards (bias + activation + pool happen there).
int conv_wino2_launch(const ConvParams& pin, int pool, hipStream_t s) {
    const long kso = pin.ksplit > 1 ? pin.ksplit : 1;
    if ((long)pin.B * pin.H * pin.W * pin.in_cs * 4 >= (1L << 31) || kso * pin.B * pin.Ho * pin.Wo * pin.out_cs * 4 >= (1L << 31)) return -1;
    if (pin.nsub != 1 && pin.nsub != 9) return -1;
    if (pin.Cout % W2_COUTS || pin.Cin % W2_CK) return -1;
    ConvParams p = pin;
    p.tiles_x = (p.Wo + 1) / 2;
    p.tiles_y = (p.Ho + 1) / 2;
    const long tiles = (long)p.B * p.tiles_x * p.tiles_y;
    if (pool && p.nsub != 1) return -1;
    if (p.ksplit > 1) {
        const int nsteps = p.nsub * p.Cin / W2_CK;
        if (pool || p.ksplit * 2 > nsteps || p.out_cs != p.Cout) return -1;
        if (p.nsub == 9) wino2_launch_t<false, 9, true>(p, tiles, s); else wino2_launch_t<false, 1, true>(p, tiles, s);
        return 0;
    }
    p.ksplit = 1;
    if (p.nsub == 9) wino2_launch_t<false, 9, false>(p, tiles, s);
    else if (pool) wino2_launch_t<true, 1, false>(p, tiles, s);
    else wino2_launch_t<false, 1, false>(p, tiles, s);
    return 0;
}

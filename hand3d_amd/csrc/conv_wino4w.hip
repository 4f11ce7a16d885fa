// conv_wino4w.hip -- conv_wino4.hip's Winograd F(4x4,3x3) with WIDE items: 16 tiles x 128 couts in 32-channel steps (round 4, option "wino4_wide").
//
// Same arithmetic, same packed filters ([36 planes][16-channel step][Cout/16][q][n][e], wino4_pack_weights), same call sites
// (NetworkOps.conv_relu + max_pool, utils/general.py:36-65; the 3x3 / stride-1 trunk layers of nets/ColorHandPose3DNetwork.py:144-157,
// 183-199 with Cout a multiple of 128).  What changes is the shape of a work item and with it the share of non-MFMA work:
//   conv_wino4.hip : item = 32 tiles x  64 couts, step = 16 channels: per step 36 window loads + one B^T d B per loader thread for
//                    288 MFMAs per wave; the 6x6 windows of a tile block are loaded and transformed once per 64-cout block;
//   here           : item = 16 tiles x 128 couts, step = 32 channels: the same loads and the same transform per loader thread and step
//                    (thread = (tile, channel pair): 16 x 16) for 576 MFMAs per wave, and half as many cout blocks repeat a tile block's
//                    transform.  Measured in conv_wino4 (profiles/r04_sq_counters.md): transform 11.6 % + window issue ~11 % of a wave's
//                    time -- both halve per MFMA here; what doubles is the weight stream (a wave owns 32 couts: two B fragments per k quad).
// Accumulators: 36 planes x 2 cout groups x 4 = 288 per lane, pinned like conv_wino4's (planes 0..31 AGPRs, 32..35 arch VGPRs); V is
// [plane][tile 16][channel 32] = the same 2 KB per plane, double buffered (147 KB); a tile's row is eight 16-byte quads XOR-swizzled by
// s(tile) = 2 * (((tile >> 1) & 1) + 2 * (tile >> 3)): with ds_read_b128 served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}
// (+32) every group then touches 16 different bank quads (DESIGN.md section 4.4's derivation, for a 128-byte row).
// A plane is EIGHT pairs of MFMAs (channel half h, k quad e; both cout groups) with one memory instruction behind each.
// Not here (stays on conv_wino4.hip): 7x7 filters, channel splits, Cout = 64 layers.
#include "hp3d_common.h"
#include "wino4_shared.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

#ifndef HP3D_WW_TIMING
#define HP3D_WW_TIMING 0         // 1: diagnostic build (as conv_wino4.hip's HP3D_W4_TIMING): every wave sums shader-clock intervals of its steps into
#endif                           // ww_timing[] -- planes 0..29 | transform | planes 30..35 | barrier | between steps / epilogue; the launcher prints them
#if HP3D_WW_TIMING
__device__ unsigned long long ww_timing[8];
#define WW_CLOCK() __builtin_readcyclecounter()
#endif

namespace {

constexpr int WW_TILES = 16;                       // Winograd tiles (4x4 outputs each) per item
constexpr int WW_CK = 32;                          // channels per step (two 16-channel steps of the packed filters)
constexpr int WW_COUTS = 128;                      // output channels per item (32 per wave: two MFMA column groups)
constexpr int WW_NP = W4_NP;
constexpr int WW_PLANE_FLOATS = WW_TILES * WW_CK;  // 2 KB
constexpr int WW_VBUF_FLOATS = WW_NP * WW_PLANE_FLOATS;
constexpr int WW_SMEM_BYTES = 2 * WW_VBUF_FLOATS * 4 + 2 * 2 * WW_TILES * 4;
#ifndef HP3D_WW_RING
#define HP3D_WW_RING 6
#endif
constexpr int WW_RING = HP3D_WW_RING;              // weight fragments in flight, in HALF planes (one slot = both cout groups of one channel half = 8 registers)
static_assert((2 * WW_NP) % WW_RING == 0, "static ring slots");
constexpr int WW_HALF = 18;                        // planes reachable from one LDS base (16-bit immediate offsets)
constexpr int WW_AGPR_PLANES = 32;
constexpr int WW_TRANSFORM_AT = 29;
constexpr int WW_PIECE_FLOATS = WW_TILES * 16 * WW_COUTS;       // raw 4x4 outputs of one item: [tile 16][pixel 16][cout 128] = 128 KB

#define WW_ISSUE_ELEM(k) W4_ISSUE_ELEM(k)          // window issue order: class by class (wino4_shared.h)

__device__ __forceinline__ int ww_swz(int t) { return 2 * (((t >> 1) & 1) + 2 * (t >> 3)); }

// B^T / A^T of F(4x4,3x3), tile / item geometry, tail reduction: wino4_shared.h (the same association as conv_wino4.hip: bit-identical products)
using WWGeom = W4GeomT<WW_TILES, WW_COUTS>;

// TAIL pieces as in conv_wino4.hip: virtual item ids [0, nfull) are whole items; nfull + 2 w + j = piece j of workgroup w's run of the
// under-filled last round's item-steps (raw sums to the compact scratch [workgroup][piece 2][tile][pixel][cout], added by ww_tail_reduce).
template <bool POOL>
HP3D_KERNEL2(256, 1)
void conv_wino4w_kernel(const ConvParams p) {
    HP3D_DYN_SMEM(V);
    int* tinfo = (int*)(V + 2 * WW_VBUF_FLOATS);       // [parity][0..15] output offset of tile t (-1: none), [16..31] edge flags
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = HP3D_READFIRSTLANE(tid >> 6);
    const int ln = lane & 15, lq = lane >> 4;

    const WWGeom geo(p);
    const int nitems = geo.tile_blocks * geo.ncy;
    const int nfull = nitems - p.tail_items;
    const int Hs = POOL ? (p.Ho >> 1) : p.Ho, Ws = POOL ? (p.Wo >> 1) : p.Wo;
    auto table_write = [&](int tblock, int parity, int piece) {
        if (tid < WW_TILES) {
            int tb, tyy, txx;
            geo.tile_decode(tblock * WW_TILES + tid, tb, tyy, txx);
            int off = -1, fl = 0;
            if (piece >= 0) {
                off = piece * WW_PIECE_FLOATS + tid * (16 * WW_COUTS);
                fl = 4 | (4 << 4);
            } else if (tb < p.B) {
                if (POOL) {
                    if (2 * tyy < Hs && 2 * txx < Ws) {
                        off = ((tb * Hs + 2 * tyy) * Ws + 2 * txx) * p.out_cs;
                        fl = (2 * txx + 1 < Ws ? 1 : 0) | (2 * tyy + 1 < Hs ? 2 : 0);
                    }
                } else {
                    off = ((tb * Hs + 4 * tyy) * Ws + 4 * txx) * p.out_cs;
                    fl = min(4, Hs - 4 * tyy) | (min(4, Ws - 4 * txx) << 4);
                }
            }
            tinfo[parity * 2 * WW_TILES + tid] = off;
            tinfo[parity * 2 * WW_TILES + WW_TILES + tid] = fl;
        }
    };

    // ---- loader role: this thread transforms the 6x6 window of tile lt for channel pair lp (of the step's 32 channels) ---------------
    const int lt = tid >> 4, lp = tid & 15;
    const int cs4 = p.in_cs * 4;
    constexpr int OOR = (int)0x80000000;
    constexpr int COL_OOR = 0x60000000;
    int ro[6], co[6];
    auto loader_setup = [&](int tblock, bool valid) {
        int lb, lty, ltx;
        geo.tile_decode(tblock * WW_TILES + lt, lb, lty, ltx);
        const int wy0 = 4 * lty - 1, wx0 = 4 * ltx - 1;
        const int wbase = ((lb * p.H + wy0) * p.W + wx0) * cs4 + lp * 8;
        const bool tv = valid && lb < p.B;
#pragma unroll
        for (int r = 0; r < 6; ++r) ro[r] = (tv && (unsigned)(wy0 + r) < (unsigned)p.H) ? wbase + r * (p.W * cs4) : OOR;
#pragma unroll
        for (int c = 0; c < 6; ++c) co[c] = (unsigned)(wx0 + c) < (unsigned)p.W ? c * cs4 : COL_OOR;
    };
    const hp3d_rsrc_t irsrc = HP3D_MAKE_RSRC(p.in, (unsigned)p.B * (unsigned)(p.H * p.W) * (unsigned)cs4);
    const unsigned out_bytes = (unsigned)p.B * (unsigned)(Hs * Ws) * (unsigned)p.out_cs * 4u;

    f32x2 d[36];
    float* const Vw = V + lt * WW_CK + ((lp >> 1) ^ ww_swz(lt)) * 4 + (lp & 1) * 2;      // this thread's slot in plane 0 of buffer 0
    auto transform_arith = [&]() {
#pragma unroll
        for (int c = 0; c < 6; ++c) w4_bt(d[0 * 6 + c], d[1 * 6 + c], d[2 * 6 + c], d[3 * 6 + c], d[4 * 6 + c], d[5 * 6 + c]);
#pragma unroll
        for (int a = 0; a < 6; ++a) w4_bt(d[a * 6 + 0], d[a * 6 + 1], d[a * 6 + 2], d[a * 6 + 3], d[a * 6 + 4], d[a * 6 + 5]);
    };
    auto v_write = [&](int buf, int pl) {
        float* Vq0 = Vw + buf * WW_VBUF_FLOATS;
        float* dst = pl < WW_HALF ? Vq0 + pl * WW_PLANE_FLOATS : Vq0 + WW_HALF * WW_PLANE_FLOATS + (pl - WW_HALF) * WW_PLANE_FLOATS;
        *(f32x2*)dst = d[pl];
    };

    // ---- MFMA role: wave w owns couts 32 w .. 32 w + 31 of the item (two column groups g of 16), all 16 tiles ---------------------
    const int CO16 = p.Cout >> 4;
    const int nsteps = p.Cin / WW_CK;                       // 32-channel steps; the packed filters count 16-channel steps: 2 s + h
    const hp3d_rsrc_t wrsrc = HP3D_MAKE_RSRC(p.wpk, (unsigned)(WW_NP * p.Cin) * (unsigned)p.Cout * 4u);
    const int step16_stride_b = CO16 * 1024;
    const int plane_stride_b = 2 * nsteps * step16_stride_b;
    // scalar offset of a fragment: plane, 16-channel step 2 s + h, and the wave's cout block `cyoff` (bytes).  The per-lane part of the address is
    // the constant lane * 16: nothing about the NEXT item has to live in a vector register across the epilogue (a spilled VGPR reloaded there
    // costs an s_waitcnt vmcnt(0) behind the epilogue's stores, i.e. a full drain of the store queue per item)
    auto soff_of = [&](int plane, int step, int h, int cyoff) { return plane * plane_stride_b + (2 * step + h) * step16_stride_b + cyoff; };
    const int wv_lane = lane * 16;

    f32x4 M[WW_NP][2];     // [plane][cout group]: rows = tiles 4 (lane >> 4) + r, column = cout 16 g + (lane & 15)
    f32x4 bq[WW_RING][2];  // [half-plane slot][cout group]
    auto b_fetch1 = [&](int slot, int g, int soff) { bq[slot][g] = HP3D_BUFFER_LOAD16(wrsrc, wv_lane + g * 1024, soff); };
    // A fragments: channel half h, k slot lq -> quad (4 h + lq) ^ s(tile); h flips bit 2 of the quad = byte offset ^ 64
    const int va_lane0 = (ln * WW_CK + ((lq ^ ww_swz(ln)) * 4)) * 4;
    const int va_lane1 = va_lane0 ^ 64;
    int ab[2][2] = {{0, 0}, {0, 0}};                        // [channel half][plane half] LDS byte address of this lane's fragment in plane 0 / 18
    f32x4 af[2][2];                                         // [set = plane parity][channel half]
    auto a_fetch = [&](int set, int plane, int h) {
        const int base = ab[h][plane < WW_HALF ? 0 : 1], pl = plane < WW_HALF ? plane : plane - WW_HALF;
        af[set][h] = *(const f32x4*)((const char*)V + base + pl * (WW_PLANE_FLOATS * 4));
    };

    auto split_of = [&](int it, int& cy_, int& tb_, int& piece_, int& s0_, int& s1_) {
        piece_ = -1;
        if (it >= nfull) {
            piece_ = it - nfull;
            const int w = piece_ >> 1;
            const int a = w * p.tail_q, b = min(a + p.tail_q, p.tail_items * nsteps);
            const int i0 = a / nsteps;
            if (piece_ & 1) { s0_ = 0; s1_ = b - (i0 + 1) * nsteps; geo.item_decode(nfull + i0 + 1, cy_, tb_); }
            else { s0_ = a - i0 * nsteps; s1_ = min(nsteps, s0_ + b - a); geo.item_decode(nfull + i0, cy_, tb_); }
            return;
        }
        geo.item_decode(it, cy_, tb_);
        s0_ = 0;
        s1_ = nsteps;
    };
    auto next_of = [&](int it) {
        const int nx = it + (int)gridDim.x;
        if (it < nfull && nx < nfull) return nx;
        const int w = (int)blockIdx.x, a = w * p.tail_q, tot = p.tail_items * nsteps;
        if (a >= tot) return -1;
        const int b = min(a + p.tail_q, tot), i0 = a / nsteps;
        if (it < nfull) return nfull + 2 * w;
        if (it == nfull + 2 * w && b > (i0 + 1) * nsteps) return nfull + 2 * w + 1;
        return -1;
    };
    int item = blockIdx.x;
    if (item >= nfull) {
        if ((int)blockIdx.x * p.tail_q >= p.tail_items * nsteps) return;
        item = nfull + 2 * (int)blockIdx.x;
    }
    int cy, tblock, piece, s0, s1;
    split_of(item, cy, tblock, piece, s0, s1);
    cy = HP3D_READFIRSTLANE(cy); tblock = HP3D_READFIRSTLANE(tblock); piece = HP3D_READFIRSTLANE(piece);
    s0 = HP3D_READFIRSTLANE(s0); s1 = HP3D_READFIRSTLANE(s1);
    loader_setup(tblock, true);
    table_write(tblock, 0, piece);
    int cyoff = HP3D_READFIRSTLANE((cy * (WW_COUTS / 16) + wave * 2) * 1024);        // byte offset of this wave's two cout groups inside a (plane, step) block
#pragma unroll
    for (int e = 0; e < 36; ++e) d[e] = HP3D_BUFFER_LOAD8(irsrc, (int)((unsigned)ro[e / 6] + (unsigned)co[e % 6]), s0 * (WW_CK * 4));
#pragma unroll
    for (int t = 0; t < WW_RING; ++t) {
        b_fetch1(t, 0, soff_of(t >> 1, s0, t & 1, cyoff));
        b_fetch1(t, 1, soff_of(t >> 1, s0, t & 1, cyoff));
    }
    transform_arith();
#pragma unroll
    for (int pl = 0; pl < WW_NP; ++pl) v_write(0, pl);
    __syncthreads();
    int cur = 0;
#if HP3D_WW_TIMING
    unsigned long long tsum[5] = {0, 0, 0, 0, 0}, t_mark = WW_CLOCK();
#endif

    for (int k = 0;; ++k) {
        int n_cy = cy, n_tblock = tblock, n_cyoff = cyoff, n_s0 = s0, n_s1 = s1, n_piece = -1;
        const int n_item = next_of(item);
        const bool raw = piece >= 0;

        auto step_body = [&](int step, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const bool lasts = step + 1 == s1;
            const int ncyoff = lasts ? n_cyoff : cyoff;
            const int nstep = lasts ? n_s0 : step + 1;
#if HP3D_WW_TIMING
            { const unsigned long long t = WW_CLOCK(); tsum[4] += t - t_mark; t_mark = t; }      // (item switch / epilogue / step prologue)
#endif
            ab[0][0] = cur * (WW_VBUF_FLOATS * 4) + va_lane0;
            ab[1][0] = cur * (WW_VBUF_FLOATS * 4) + va_lane1;
            ab[0][1] = ab[0][0] + WW_HALF * WW_PLANE_FLOATS * 4;
            ab[1][1] = ab[1][0] + WW_HALF * WW_PLANE_FLOATS * 4;
            HP3D_OPAQUE_V(ab[0][0]);
            HP3D_OPAQUE_V(ab[1][0]);
            HP3D_OPAQUE_V(ab[0][1]);
            HP3D_OPAQUE_V(ab[1][1]);
            a_fetch(0, 0, 0);
            a_fetch(0, 0, 1);
            if (lasts) loader_setup(n_tblock, n_item >= 0);
            const int wsoff = nstep * (WW_CK * 4);
#pragma unroll
            for (int pl = 0; pl < WW_NP; ++pl) {
                const int as = pl & 1;
#pragma unroll
                for (int gi = 0; gi < 8; ++gi) {
                    const int h = gi >> 2, e = gi & 3;
                    const int bs = (2 * pl + h) % WW_RING;
                    HP3D_SCHED_BARRIER();
                    if (FIRST && gi == 0) {
                        if (pl < WW_AGPR_PLANES) HP3D_MFMA16_PAIRB_FIRST("a", M[pl][0], M[pl][1], af[as][h][e], bq[bs][0][e], bq[bs][1][e]);
                        else HP3D_MFMA16_PAIRB_FIRST("v", M[pl][0], M[pl][1], af[as][h][e], bq[bs][0][e], bq[bs][1][e]);
                    } else {
                        if (pl < WW_AGPR_PLANES) HP3D_MFMA16_PAIRB("a", M[pl][0], M[pl][1], af[as][h][e], bq[bs][0][e], bq[bs][1][e]);
                        else HP3D_MFMA16_PAIRB("v", M[pl][0], M[pl][1], af[as][h][e], bq[bs][0][e], bq[bs][1][e]);
                    }
                    HP3D_SCHED_BARRIER();
                    // one memory instruction behind each pair: next plane's A fragments (gaps 0, 4) | window / V write (gaps 2, 6) | the weight
                    // fragments of the half-plane slot just released (slot (pl, 0): gaps 3 and 5; slot (pl, 1): both behind gap 7)
                    if (gi == 0 || gi == 4) {
                        if (pl + 1 < WW_NP) a_fetch((pl + 1) & 1, pl + 1, gi >> 2);
                    } else if (gi == 3 || gi == 5 || gi == 7) {
                        const int hh = gi == 7 ? 1 : 0;
                        const int t = 2 * pl + hh + WW_RING;            // half-plane index this slot serves next
                        const int tp = t >> 1, th = t & 1;
                        const int slot = (2 * pl + hh) % WW_RING;
                        if (gi != 5) {
                            if (tp < WW_NP) b_fetch1(slot, 0, soff_of(tp, step, th, cyoff));
                            else b_fetch1(slot, 0, soff_of(tp - WW_NP, nstep, th, ncyoff));
                        }
                        if (gi != 3) {
                            if (tp < WW_NP) b_fetch1(slot, 1, soff_of(tp, step, th, cyoff));
                            else b_fetch1(slot, 1, soff_of(tp - WW_NP, nstep, th, ncyoff));
                        }
                    } else if (gi == 2 || gi == 6) {
                        const int j = gi == 2 ? 0 : 1;
                        if (pl < 18) {
                            const int we = WW_ISSUE_ELEM(2 * pl + j);
                            d[we] = HP3D_BUFFER_LOAD8(irsrc, (int)((unsigned)ro[we / 6] + (unsigned)co[we % 6]), wsoff);
                        } else if (pl > WW_TRANSFORM_AT) {
                            constexpr int PER = 36 / (2 * (WW_NP - 1 - WW_TRANSFORM_AT));
                            static_assert(PER * 2 * (WW_NP - 1 - WW_TRANSFORM_AT) == 36, "");
#pragma unroll
                            for (int q = 0; q < PER; ++q) v_write(cur ^ 1, ((pl - WW_TRANSFORM_AT - 1) * 2 + j) * PER + q);
                        }
                    }
                }
                if (pl == WW_TRANSFORM_AT) {
#if HP3D_WW_TIMING
                    { const unsigned long long t = WW_CLOCK(); tsum[0] += t - t_mark; t_mark = t; }
#endif
                    transform_arith();
#if HP3D_WW_TIMING
                    { const unsigned long long t = WW_CLOCK(); tsum[1] += t - t_mark; t_mark = t; }
#endif
                }
            }
            HP3D_SCHED_BARRIER();
#if HP3D_WW_TIMING
            { const unsigned long long t = WW_CLOCK(); tsum[2] += t - t_mark; t_mark = t; }
#endif
            __syncthreads();             // V[cur^1] complete, V[cur] free
#if HP3D_WW_TIMING
            { const unsigned long long t = WW_CLOCK(); tsum[3] += t - t_mark; t_mark = t; }
#endif
            cur ^= 1;
        };
        {
            const bool has_next = n_item >= 0;
            if (has_next) split_of(n_item, n_cy, n_tblock, n_piece, n_s0, n_s1);
            n_cy = HP3D_READFIRSTLANE(n_cy); n_tblock = HP3D_READFIRSTLANE(n_tblock); n_piece = HP3D_READFIRSTLANE(n_piece);
            n_s0 = HP3D_READFIRSTLANE(n_s0); n_s1 = HP3D_READFIRSTLANE(n_s1);
            n_cyoff = HP3D_READFIRSTLANE((n_cy * (WW_COUTS / 16) + wave * 2) * 1024);
        }
        step_body(s0, std::true_type{});
        table_write(n_tblock, (k + 1) & 1, n_piece);
        for (int step = s0 + 1; step < s1; ++step) step_body(step, std::false_type{});

        // ---- epilogue: Y = A^T M A per (tile, cout), bias + leaky-ReLU (+ 2x2 max-pool) + NHWC store (tail pieces: raw sums) ----------
#ifndef HP3D_EMU
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3");       // MFMA results written inside inline asm: the hazard recogniser cannot see them
#endif
        // (the two bias values are loaded HERE, not before the steps: live across the plane loops they were spilled, and a spill reloaded
        //  behind the first stores costs an s_waitcnt vmcnt(0) = a drain of the store queue)
        const int cout0 = cy * WW_COUTS + wave * 32 + (int)(threadIdx.x & 15);            // group g: + 16 g
        const float bias0 = raw ? 0.f : p.bias[cout0], bias1 = raw ? 0.f : p.bias[cout0 + 16];
        const int* tab = tinfo + (k & 1) * 2 * WW_TILES;
        const bool full = HP3D_OPAQUE_SGPR((((p.Ho | p.Wo) & 3) == 0 || raw) ? 1 : 0) != 0;
        const hp3d_rsrc_t orsrc = HP3D_MAKE_RSRC(raw ? (float*)p.partial : p.out, raw ? 2u * gridDim.x * (unsigned)(WW_PIECE_FLOATS * 4) : out_bytes);
        const int srow = raw ? 4 * WW_COUTS * 4 : Ws * p.out_cs * 4, scol = raw ? WW_COUTS * 4 : p.out_cs * 4;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            int tx = (int)threadIdx.x;
            HP3D_OPAQUE_V(tx);                               // (recomputed per group on purpose: kept live from group 0 it is spilled and reloaded behind 128 stores)
            const int cout = cy * WW_COUTS + wave * 32 + 16 * g + (tx & 15);
            const float bias = g ? bias1 : bias0;
            const bool cok = raw || cout < p.cout_store;
            const int cout_off = raw ? wave * 32 + 16 * g + ln : cout;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = 4 * lq + r;                    // MFMA row = Winograd tile
                const int off = tab[t];
                const int fl = tab[WW_TILES + t];
                const int vo = (cok && off >= 0) ? (off + cout_off) * 4 : OOR;
                float z[6][4];
#pragma unroll
                for (int b = 0; b < 6; ++b)
                    w4_at(M[0 * 6 + b][g][r], M[1 * 6 + b][g][r], M[2 * 6 + b][g][r], M[3 * 6 + b][g][r], M[4 * 6 + b][g][r], M[5 * 6 + b][g][r],
                          z[b][0], z[b][1], z[b][2], z[b][3]);
                float y[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    w4_at(z[0][i], z[1][i], z[2][i], z[3][i], z[4][i], z[5][i], y[i][0], y[i][1], y[i][2], y[i][3]);
                    if (!POOL && !raw) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float x = y[i][j] + bias;
                            if (p.act) x = fmaxf(x, HP3D_LEAKY_SLOPE * x);
                            y[i][j] = x;
                        }
                    }
                }
                auto store_tile = [&](auto full_tag) {
                    constexpr bool FULL = decltype(full_tag)::value;
                    if (POOL && !raw) {
#pragma unroll
                        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                            for (int pj = 0; pj < 2; ++pj) {
                                // bias + leaky-ReLU after the max (monotonic: same bits as activating the four values first)
                                float v = fmaxf(fmaxf(y[2 * pi][2 * pj], y[2 * pi][2 * pj + 1]), fmaxf(y[2 * pi + 1][2 * pj], y[2 * pi + 1][2 * pj + 1])) + bias;
                                if (p.act) v = fmaxf(v, HP3D_LEAKY_SLOPE * v);
                                const bool ok = FULL || ((pj == 0 || (fl & 1)) && (pi == 0 || (fl & 2)));
                                HP3D_BUFFER_STORE4(orsrc, v, ok ? vo : OOR, (pi * Ws + pj) * p.out_cs * 4);
                            }
                    } else {
                        const int vr = fl & 15, vc = fl >> 4;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int vrow = (FULL || i < vr) ? vo : OOR;
#pragma unroll
                            for (int j = 0; j < 4; ++j) HP3D_BUFFER_STORE4(orsrc, y[i][j], (FULL || j < vc) ? vrow : OOR, i * srow + j * scol);
                        }
                    }
                };
                if (!POOL && full) store_tile(std::true_type{});
                else store_tile(std::false_type{});
            }
        }
#if HP3D_WW_TIMING
        if (n_item < 0) {
            const unsigned long long t = WW_CLOCK();
            tsum[4] += t - t_mark;
            if (lane == 0) {
                for (int i = 0; i < 5; ++i) atomicAdd(&ww_timing[i], tsum[i]);
                atomicAdd(&ww_timing[5], 1ull);
            }
        }
#endif
        if (n_item < 0) break;
        item = n_item; cy = n_cy; tblock = n_tblock; cyoff = n_cyoff; piece = n_piece; s0 = n_s0; s1 = n_s1;
    }
}

// Tail pieces -> outputs (wino4_shared.h: w4_tail_reduce_body with this file's item shape)
template <bool POOL>
HP3D_KERNEL(256)
void ww_tail_reduce_kernel(const ConvParams p) { w4_tail_reduce_body<POOL, WW_TILES, WW_COUTS, WW_CK>(p, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x); }

}  // namespace

// 1: this layer can run on the wide-item kernel (3x3 / stride 1, Cout a multiple of 128, Cin of 32, enough items to fill the chip without a channel split)
int conv_wino4w_eligible(int mode, int k, int stride, int Cin, int Cout, int Ho, int Wo, int B, int in_cs, int out_cs, int pool) {
    if (!mode || k != 3 || stride != 1 || Cin % WW_CK || Cout % WW_COUTS) return 0;
    if ((long)B * Ho * Wo * in_cs * 4 >= (1L << 30) || (long)B * Ho * Wo * out_cs * 4 >= (1L << 31)) return 0;
    if (pool && ((Ho | Wo) & 1)) return 0;
    const long tiles = (long)B * ((Ho + 3) / 4) * ((Wo + 3) / 4);
    const long items = (tiles + WW_TILES - 1) / WW_TILES * (Cout / WW_COUTS);
    // mode 1: where it measured faster than conv_wino4.hip (profiles/r04_wide_items.md, B = 32 at 320x320, per layer): Cin >= 256 -2..-6 % (512 -> 512 at
    // 40x40 0.565 -> 0.535 ms, 256 -> 256 at 64x64 0.366 -> 0.349), Cin = 128 +-1..4 %, Cin = 64 +11..17 % slower (two steps per item; a weight ring of 9
    // half planes instead of 6 does not change it)
    return mode == 2 || (items >= 2 * hp3d_num_cus() && Cin >= 256);          // (one-round launches, PoseNet2D conv4_4..6: +2 % slower)
}

// tail plan of conv_wino4.hip with this file's item shape (scratch: conv_wino4_tail_floats(), the piece size is the same 128 KB)
int conv_wino4w_tail_plan(int Cin, int Cout, int Ho, int Wo, int B, int* tail_items) {
    if (tail_items) *tail_items = 0;
    const long tiles = (long)B * ((Ho + 3) / 4) * ((Wo + 3) / 4);
    const long items = (tiles + WW_TILES - 1) / WW_TILES * (Cout / WW_COUTS);
    const int slots = hp3d_num_cus(), nsteps = Cin / WW_CK;
    const int rem = (int)(items % slots);
    if (rem == 0 || rem * 8 > slots * 7 || nsteps < 2) return 0;
    int q = (int)(((long)rem * nsteps + slots - 1) / slots);
    if (q < 2) q = 2;
    if (q >= nsteps) return 0;
    if (tail_items) *tail_items = rem;
    return q;
}

template <bool POOL>
static void ww_launch_t(const ConvParams& p, long tiles, hipStream_t s) {
    static bool attr_done[64] = {};
    auto k = conv_wino4w_kernel<POOL>;
    if (hp3d_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, WW_SMEM_BYTES);
    const long items = (tiles + WW_TILES - 1) / WW_TILES * (p.Cout / WW_COUTS);
    const int slots = hp3d_num_cus();
    dim3 grid((unsigned)((items < slots && p.tail_items == 0) ? items : slots));
    HP3D_LAUNCH(k, grid, dim3(256), WW_SMEM_BYTES, s, p);
}

#if HP3D_WW_TIMING
static void ww_timing_report(const ConvParams& p, hipStream_t s, const char* what) {
    unsigned long long h[8] = {};
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(ww_timing), sizeof(h));
    if (h[5]) {
        const double w = (double)h[5];
        fprintf(stderr, "ww_timing %s Cin %d Cout %d %dx%d B %d: per wave (cycles) planes 0..29 %.0f | transform %.0f | planes 30..35 %.0f | barrier %.0f | "
                        "between steps / epilogue %.0f | waves %.0f\n", what, p.Cin, p.Cout, p.Ho, p.Wo, p.B, h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, w);
    }
    unsigned long long z[8] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(ww_timing), z, sizeof(z));
}
#endif
int conv_wino4w_launch(const ConvParams& pin, int pool, hipStream_t s) {
    if ((long)pin.B * pin.H * pin.W * pin.in_cs * 4 >= (1L << 30) || (long)pin.B * pin.Ho * pin.Wo * pin.out_cs * 4 >= (1L << 31)) return -1;
    if (pin.nsub != 1 || pin.ksplit > 1 || pin.Cout % WW_COUTS || pin.Cin % WW_CK) return -1;
    if (pool && ((pin.Ho | pin.Wo) & 1)) return -1;
    ConvParams p = pin;
    p.tiles_x = (p.Wo + 3) / 4;
    p.tiles_y = (p.Ho + 3) / 4;
    p.ksplit = 1;
    p.tail_items = p.tail_q = 0;
    const long tiles = (long)p.B * p.tiles_x * p.tiles_y;
    if (p.partial && p.partial_cap >= conv_wino4_tail_floats() && !(pool && (p.cout_store & 3)) && (p.out_cs & 3) == 0 && ((uintptr_t)p.out & 15) == 0)
        p.tail_q = conv_wino4w_tail_plan(p.Cin, p.Cout, p.Ho, p.Wo, p.B, &p.tail_items);
    if (pool) ww_launch_t<true>(p, tiles, s);
    else ww_launch_t<false>(p, tiles, s);
#if HP3D_WW_TIMING
    ww_timing_report(p, s, pool ? "3x3 pool" : "3x3");
#endif
    if (p.tail_items > 0) {
        const long total = (long)p.tail_items * WW_TILES * (pool ? 4 : 16) * (WW_COUTS / 4);
        const unsigned blocks = (unsigned)((total + 255) / 256);
        if (pool) HP3D_LAUNCH(ww_tail_reduce_kernel<true>, dim3(blocks), dim3(256), 0, s, p);
        else HP3D_LAUNCH(ww_tail_reduce_kernel<false>, dim3(blocks), dim3(256), 0, s, p);
        return 1;                        // (as conv_wino4_launch: 1 = the last round ran as tail pieces)
    }
    return 0;
}

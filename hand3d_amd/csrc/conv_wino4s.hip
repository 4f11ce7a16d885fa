// conv_wino4s.hip -- Winograd F(4x4, 3x3) with the 36 plane products on the bfloat16 matrix pipe, SPLIT OPERANDS (round 6).
//
// Same call sites and the same arithmetic outside the products as conv_wino4.hip (NetworkOps.conv_relu + max_pool, utils/general.py:36-65;
// the 3x3 / stride-1 trunk layers with Cin >= 128, nets/ColorHandPose3DNetwork.py:146-157,185-199): U = G g G^T evaluated in double and
// rounded once to float32, V = B^T d B in float32, Y = A^T M A in float32.  The plane products M = sum_cin U .* V leave the float32 matrix
// instruction (v_mfma_f32_16x16x4_f32: 1/16 of the bf16 rate) for v_mfma_f32_16x16x32_bf16 on THREE bfloat16 pieces per operand,
//     x = x0 + x1 + x2 exactly (each piece the round-to-nearest bfloat16 of what the previous ones left),
// and the SIX piece products of weight >= 2^-16, accumulated in float32:
//     u v ~= u0 v0 + u0 v1 + u1 v0 + u0 v2 + u1 v1 + u2 v0        (dropped: u1 v2 + u2 v1 + u2 v2 <= 2^-23 |u v|, signs mixed)
// -- the pieces and their products are exact, only the accumulation rounds.  Round 5's CPU emulation priced that at HALF conv_wino4's error
// (profiles/r05_splithalf_numerics.md); measured on the GPU the two kernels are EQUALLY exact (0.7 ... 1.6x per shape, profiles/r06_split_numerics.md):
// the matrix pipe's accumulation of an instruction's 32 products is worth a float32 fmaf chain.  With that, and 1.03-1.09x on the layers it was
// built for (profiles/r06_tuning_notes.md section 1), this kernel is an OPTION (hp3d_set_option "wino4_split"), off by default.
//
// The K = 32 of the instruction holds 16 channels x 2 pieces: lane (n, q) carries channels 4 q .. 4 q + 3 of piece X in k slots 0..3 and of
// piece Y in slots 4..7, so THREE instructions cover the six products of a 16-channel step,
//     [V1|V0] x [U1|U0]  +  [V1|V0] x [U0|U2]  +  [V0|V2] x [U1|U0]   =  11 + 00 + 10 + 02 + 01 + 20,
// with two A fragments per tile block and two B fragments per cout block ([U0|U2] is the upper half of [U1|U0] + the third piece: the filter
// stream is 6 bytes per value, nothing stored twice).
//
// Machine shape -- what changes against conv_wino4.hip and why (profiles/r06_tuning_notes.md section 1, scripts/micro/split_mfma.hip):
//   * the matrix pipe is 2.7x shorter (96 cycles per plane, tile block pair and cout block), so the split of V -- 5.5 VALU instructions per value,
//     ~4.5 cycles each beyond the two that hide under an MFMA -- must be done ONCE per value.  conv_wino4's waves share V (32 tiles) and own 16
//     couts each: four waves would split the same values.  Here a wave owns NINE PLANES of the item's 32 tiles x 64 couts (9 planes x 2 tile
//     blocks x 4 cout blocks x 4 = 288 accumulators, the same budget): every V value is read from LDS, and split, by exactly one wave;
//   * V stays float32 in LDS (double buffered, conv_wino4's layout and loader unchanged) and is split at fragment-read time in registers;
//   * U: pre-split at pack time, [plane][step][Cout/64][cout block 4]{64 lanes x 16 B [U1|U0], 64 lanes x 8 B U2}, global -> VGPR through a
//     ring of W4S_RING fragments; a wave reads all four cout blocks of its planes: no fragment is loaded twice by a CU;
//   * the output transform needs all 36 planes of a (tile, cout): the accumulators cross the waves through LDS (the V buffer the last
//     step freed), one 16-cout block per pass: [plane][cout][tile] with a quad swizzle; a thread then transforms two tiles of one cout
//     (packed, as conv_wino4 does) and stores.
// Item order, tail pieces (stream-K cut of an under-filled last round + wino4_tail_reduce), tile tables and edge handling: conv_wino4's.
#include "hp3d_common.h"
#include "wino4_shared.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#define W4S_WLOAD HP3D_BUFFER_LOAD8

namespace {

constexpr int W4S_TILES = 32;                        // Winograd tiles (4x4 outputs each) per item
constexpr int W4S_CK = 16;                           // channels per step
constexpr int W4S_COUTS = 64;                        // output channels per item
constexpr int W4S_PLANE_FLOATS = W4S_TILES * W4S_CK; // one plane of one V buffer: 2 KB
constexpr int W4S_VBUF_FLOATS = W4_NP * W4S_PLANE_FLOATS;
constexpr int W4S_SMEM_BYTES = 2 * W4S_VBUF_FLOATS * 4 + 2 * 2 * W4S_TILES * 4;
constexpr int W4S_WP = 9;                            // planes per wave
constexpr int W4S_CB = W4S_COUTS / 16;               // 16-cout blocks per item
constexpr int W4S_SLOTS = W4S_WP * W4S_CB;           // filter fragments per wave and step
#ifndef HP3D_W4S_RING
#define HP3D_W4S_RING 6
#endif
constexpr int W4S_RING = HP3D_W4S_RING;              // filter fragments in flight per wave (6 registers each); must divide 36
static_assert(W4S_SLOTS % W4S_RING == 0, "static ring slots need a ring that divides the fragment count of a step");
constexpr int W4S_FRAG_BYTES = 64 * 16 + 64 * 8;     // one (plane, step, cout block): [U1|U0] per lane, then U2 per lane
constexpr int W4S_WINDOW_PLANES = 4;                 // the 36 window loads of the next step go behind the gaps of the first 4 planes (9 each)
constexpr int W4S_TRANSFORM_FIRST = 5;               // its input transform runs under planes 5..7 (four of the twelve B^T passes each)
constexpr int W4S_PIECE_FLOATS = W4S_TILES * 16 * W4S_COUTS;

__device__ __forceinline__ int w4s_swz(int t) { return (0x78 >> (((t >> 2) & 3) * 2)) & 3; }
__device__ __forceinline__ float w4s_hi16(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }
__device__ __forceinline__ float w4s_lo16(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }

using W4sGeom = W4GeomT<W4S_TILES, W4S_COUTS>;

template <bool POOL>
HP3D_KERNEL2(256, 1)
void conv_wino4s_kernel(const ConvParams p) {
    HP3D_DYN_SMEM(V);
    int* tinfo = (int*)(V + 2 * W4S_VBUF_FLOATS);      // [parity][0..31] output offset of tile t (-1: none), [32..63] edge flags
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = HP3D_READFIRSTLANE(tid >> 6);
    const int ln = lane & 15, lq = lane >> 4;          // MFMA row / column within a block, k group

    const W4sGeom geo(p);
    const int nitems = geo.tile_blocks * geo.ncy;
    const int nfull = nitems - p.tail_items;           // virtual item ids [0, nfull): whole items; nfull + 2 w + j: piece j of workgroup w's run
    auto tile_decode = [&](int id, int& tb, int& tyy, int& txx) { geo.tile_decode(id, tb, tyy, txx); };
    const int Hs = POOL ? (p.Ho >> 1) : p.Ho, Ws = POOL ? (p.Wo >> 1) : p.Wo;
    auto table_write = [&](int tblock, int parity, int piece) {
        if (tid < W4S_TILES) {
            int tb, tyy, txx;
            tile_decode(tblock * W4S_TILES + tid, tb, tyy, txx);
            int off = -1, fl = 0;
            if (piece >= 0) {                      // raw piece: every tile's 4x4 outputs, compact
                off = piece * W4S_PIECE_FLOATS + tid * (16 * W4S_COUTS);
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
            tinfo[parity * 2 * W4S_TILES + tid] = off;
            tinfo[parity * 2 * W4S_TILES + W4S_TILES + tid] = fl;
        }
    };

    // ---- loader role (conv_wino4.hip's): this thread transforms the 6x6 window of tile lt for channel pair lp -------------
    const int lt = tid >> 3, lp = tid & 7;
    const int cs4 = p.in_cs * 4;
    constexpr int OOR = (int)0x80000000;          // row outside the image / no such tile
    constexpr int COL_OOR = 0x60000000;           // column outside the image: any row term + this is >= 2^30 > the buffer's extent
    int ro[6], co[6];
    auto loader_setup = [&](int tblock, bool valid) {
        int lb, lty, ltx;
        tile_decode(tblock * W4S_TILES + lt, lb, lty, ltx);
        const int wy0 = 4 * lty - 1, wx0 = 4 * ltx - 1;
        const int wbase = ((lb * p.H + wy0) * p.W + wx0) * cs4 + lp * 8;
        const bool tv = valid && lb < p.B;
#pragma unroll
        for (int r = 0; r < 6; ++r) ro[r] = (tv && (unsigned)(wy0 + r) < (unsigned)p.H) ? wbase + r * (p.W * cs4) : OOR;
#pragma unroll
        for (int c = 0; c < 6; ++c) co[c] = (unsigned)(wx0 + c) < (unsigned)p.W ? c * cs4 : COL_OOR;
    };
    const hp3d_rsrc_t irsrc = HP3D_MAKE_RSRC(p.in, (unsigned)p.B * (unsigned)(p.H * p.W) * (unsigned)cs4);
    [[maybe_unused]] const unsigned out_bytes = (unsigned)p.B * (unsigned)(Hs * Ws) * (unsigned)p.out_cs * 4u;

    f32x2 d[36];
    auto window_fetch = [&](int soff) {
#pragma unroll
        for (int e = 0; e < 36; ++e) d[e] = W4S_WLOAD(irsrc, (int)((unsigned)ro[e / 6] + (unsigned)co[e % 6]), soff);
    };
    float* const Vw = V + lt * W4S_CK + ((lp >> 1) ^ w4s_swz(lt)) * 4 + (lp & 1) * 2;      // this thread's slot in plane 0 of buffer 0
    // B^T d B in place, as twelve passes: 0..5 along the window rows (column c), 6..11 along the columns (row a)
    auto transform_pass = [&](int i) {
        if (i < 6) w4_bt(d[0 * 6 + i], d[1 * 6 + i], d[2 * 6 + i], d[3 * 6 + i], d[4 * 6 + i], d[5 * 6 + i]);
        else { const int a = i - 6; w4_bt(d[a * 6 + 0], d[a * 6 + 1], d[a * 6 + 2], d[a * 6 + 3], d[a * 6 + 4], d[a * 6 + 5]); }
    };
    auto v_write = [&](int buf, int pl) { *(f32x2*)(Vw + buf * W4S_VBUF_FLOATS + pl * W4S_PLANE_FLOATS) = d[pl]; };

    // ---- MFMA role: planes 9 wave .. 9 wave + 8 of every tile and cout of the item -------------------------------------------------------
    const int csteps = p.Cin / W4S_CK;
    const int nsteps = csteps;
    const hp3d_rsrc_t wrsrc = HP3D_MAKE_RSRC(p.wpk, (unsigned)(W4_NP * p.Cin) * (unsigned)p.Cout * 6u);
    const int cy_stride_b = W4S_CB * W4S_FRAG_BYTES;
    const int step_stride_b = geo.ncy * cy_stride_b;
    const int plane_stride_b = nsteps * step_stride_b;
    const int wplane0 = wave * W4S_WP;
    // fragment q = 4 j + c of (step, cout block group cy): plane 9 wave + j, cout block c
    auto soff_of = [&](int q, int step, int cy_) { return (wplane0 + q / W4S_CB) * plane_stride_b + step * step_stride_b + cy_ * cy_stride_b + (q % W4S_CB) * W4S_FRAG_BYTES; };
    const int bv16 = lane * 16, bv8 = 64 * 16 + lane * 8;

    f32x4 M[W4S_WP][2][W4S_CB];   // [plane][tile block][cout block]: rows = tiles 16 m + 4 (lane >> 4) + r, column = cout 16 c + (lane & 15)
    f32x4 bq[W4S_RING];           // [U1|U0]: registers 0..1 = piece 1 of channels 4 q .. 4 q + 3, registers 2..3 = piece 0
    f32x2 bq2[W4S_RING];          // U2
    f32x4 b02;                    // [U0|U2] of the cout block whose second product comes next (built one gap ahead: see the hazard note below)
    auto b_fetch = [&](int slot, int soff) {
        bq[slot] = HP3D_BUFFER_LOAD16(wrsrc, bv16, soff);
        bq2[slot] = HP3D_BUFFER_LOAD8(wrsrc, bv8, soff);
    };
    const int va_lane = (ln * W4S_CK + ((lq ^ w4s_swz(ln)) * 4)) * 4 + wplane0 * (W4S_PLANE_FLOATS * 4);
    int ab = 0;
    f32x4 raw[2][2];              // [set][tile block]: V of channels 4 q .. 4 q + 3, float32
    u32x4 fa10[2][2], fa02[2][2]; // [set][tile block]: [V1|V0], [V0|V2]
    auto a_fetch = [&](int set, int j) {
#pragma unroll
        for (int m = 0; m < 2; ++m) raw[set][m] = *(const f32x4*)((const char*)V + ab + (j * W4S_PLANE_FLOATS + m * 16 * W4S_CK) * 4);
    };
    // the split of one channel quad in six stages of four instructions (they go behind six MFMA pairs)
    unsigned sp1[2], sp2[2];
    float sh[4], sr[4];
    auto split_stage = [&](int st, int set, int m) {
        const f32x4 x = raw[set][m];
        if (st == 0) { sp1[0] = hp3d_cvt_pk_bf16(x[0], x[1]); sp1[1] = hp3d_cvt_pk_bf16(x[2], x[3]); sh[0] = w4s_lo16(sp1[0]); sh[1] = w4s_hi16(sp1[0]); }
        if (st == 1) { sh[2] = w4s_lo16(sp1[1]); sh[3] = w4s_hi16(sp1[1]); sr[0] = x[0] - sh[0]; sr[1] = x[1] - sh[1]; }
        if (st == 2) { sr[2] = x[2] - sh[2]; sr[3] = x[3] - sh[3]; sp2[0] = hp3d_cvt_pk_bf16(sr[0], sr[1]); sp2[1] = hp3d_cvt_pk_bf16(sr[2], sr[3]); }
        if (st == 3) { sh[0] = w4s_lo16(sp2[0]); sh[1] = w4s_hi16(sp2[0]); sh[2] = w4s_lo16(sp2[1]); sh[3] = w4s_hi16(sp2[1]); }
        if (st == 4) { sr[0] -= sh[0]; sr[1] -= sh[1]; sr[2] -= sh[2]; sr[3] -= sh[3]; }
        if (st == 5) {
            const unsigned p3a = hp3d_cvt_pk_bf16(sr[0], sr[1]), p3b = hp3d_cvt_pk_bf16(sr[2], sr[3]);
            fa10[set][m] = u32x4{sp2[0], sp2[1], sp1[0], sp1[1]};
            fa02[set][m] = u32x4{sp1[0], sp1[1], p3a, p3b};
            HP3D_OPAQUE_V(fa10[set][m]);     // (assembled HERE: an MFMA must not read a register a VALU instruction wrote in the two slots before it)
            HP3D_OPAQUE_V(fa02[set][m]);
        }
    };

    // virtual item id -> cout block group, tile block, channel steps [s0_, s1_), piece slot (-1 = a whole item)
    auto split_of = [&](int it, int& cy_, int& tb_, int& piece_, int& s0_, int& s1_) {
        piece_ = -1;
        if (it >= nfull) {
            piece_ = it - nfull;
            const int w = piece_ >> 1;
            const int a = w * p.tail_q, b = min(a + p.tail_q, p.tail_items * nsteps);       // this workgroup's run of item-steps
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
        if (it == nfull + 2 * w && b > (i0 + 1) * nsteps) return nfull + 2 * w + 1;      // the run crosses into the next item
        return -1;
    };
    int item = blockIdx.x;
    if (item >= nfull) {                            // no whole item for this workgroup: straight to its run of the tail
        if ((int)blockIdx.x * p.tail_q >= p.tail_items * nsteps) return;
        item = nfull + 2 * (int)blockIdx.x;
    }
    int cy, tblock, piece, s0, s1;
    split_of(item, cy, tblock, piece, s0, s1);
    cy = HP3D_READFIRSTLANE(cy); tblock = HP3D_READFIRSTLANE(tblock); piece = HP3D_READFIRSTLANE(piece);
    s0 = HP3D_READFIRSTLANE(s0); s1 = HP3D_READFIRSTLANE(s1);
    loader_setup(tblock, true);
    table_write(tblock, 0, piece);
    window_fetch(s0 * (W4S_CK * 4));
#pragma unroll
    for (int t = 0; t < W4S_RING; ++t) b_fetch(t, soff_of(t, s0, cy));
    b02 = f32x4{bq[0][2], bq[0][3], bq2[0][0], bq2[0][1]};
    HP3D_OPAQUE_V(b02);
#pragma unroll
    for (int i = 0; i < 12; ++i) transform_pass(i);
#pragma unroll
    for (int pl = 0; pl < W4_NP; ++pl) v_write(0, pl);
    __syncthreads();
    int cur = 0;

    for (int k = 0;; ++k) {
        int n_cy = cy, n_tblock = tblock, n_s0 = s0, n_s1 = s1, n_piece = -1;
        const int n_item = next_of(item);
        const bool rawp = piece >= 0;                 // this item is a tail piece: raw sums into the compact scratch

        auto step_body = [&](int step, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const bool lasts = step + 1 == s1;
            const int ncy_ = lasts ? n_cy : cy;
            const int nstep = lasts ? n_s0 : step + 1;
            ab = cur * (W4S_VBUF_FLOATS * 4) + va_lane;
            HP3D_OPAQUE_V(ab);
            a_fetch(0, 0);
            a_fetch(1, 1);
            if (lasts) loader_setup(n_tblock, n_item >= 0);
            const int wsoff = nstep * (W4S_CK * 4);
            // plane 0's fragments: nothing to run them under (V of this step was complete only at the barrier)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int st = 0; st < 6; ++st) split_stage(st, 0, m);
            // HAZARD (measured, scripts/micro/split_unit.hip): an MFMA that reads a register a VALU instruction wrote less than two issue slots
            // earlier sees the OLD value, and the compiler's hazard recogniser does not look into the asm statements that hold the MFMAs.
            // Every VALU-made operand below (the split fragments, [U0|U2]) is therefore produced at least one whole gap -- one MFMA pair --
            // ahead of the pair that reads it; here, at the head of a step, two wait states are spent instead.
            HP3D_SCHED_BARRIER();
            HP3D_MFMA_OPERAND_FENCE();
#pragma unroll
            for (int j = 0; j < W4S_WP; ++j) {
                const int as = j & 1;
#pragma unroll
                for (int g = 0; g < 12; ++g) {        // twelve MFMA pairs (both tile blocks): cout block g / 3, product g % 3
                    const int c = g / 3, pr = g % 3, q = j * W4S_CB + c, bs = q % W4S_RING;
                    HP3D_SCHED_BARRIER();
                    if (pr == 0) {
                        if (FIRST) {
                            if (j < 8) HP3D_MFMA16B_PAIR_FIRST("a", M[j][0][c], M[j][1][c], fa10[as][0], fa10[as][1], bq[bs]);
                            else HP3D_MFMA16B_PAIR_FIRST("v", M[j][0][c], M[j][1][c], fa10[as][0], fa10[as][1], bq[bs]);
                        } else {
                            if (j < 8) HP3D_MFMA16B_PAIR("a", M[j][0][c], M[j][1][c], fa10[as][0], fa10[as][1], bq[bs]);
                            else HP3D_MFMA16B_PAIR("v", M[j][0][c], M[j][1][c], fa10[as][0], fa10[as][1], bq[bs]);
                        }
                    } else if (pr == 1) {
                        if (j < 8) HP3D_MFMA16B_PAIR("a", M[j][0][c], M[j][1][c], fa10[as][0], fa10[as][1], b02);
                        else HP3D_MFMA16B_PAIR("v", M[j][0][c], M[j][1][c], fa10[as][0], fa10[as][1], b02);
                    } else {
                        if (j < 8) HP3D_MFMA16B_PAIR("a", M[j][0][c], M[j][1][c], fa02[as][0], fa02[as][1], bq[bs]);
                        else HP3D_MFMA16B_PAIR("v", M[j][0][c], M[j][1][c], fa02[as][0], fa02[as][1], bq[bs]);
                    }
                    HP3D_SCHED_BARRIER();
                    // ---- what issues behind this pair ----
                    // the next plane's fragments: tile block 0 behind pairs 0..4 (two stages behind the first), tile block 1 behind pairs 5..10;
                    // nothing behind pair 11 -- its successor, the next plane's first pair, reads them
                    if (j + 1 < W4S_WP) {
                        if (g == 0) { split_stage(0, as ^ 1, 0); split_stage(1, as ^ 1, 0); }
                        else if (g < 5) split_stage(g + 1, as ^ 1, 0);
                        else if (g < 11) split_stage(g - 5, as ^ 1, 1);
                    }
                    if (pr == 2) {
                        // [U0|U2] of the NEXT cout block (the next plane's / step's first at the end of a plane): read two pairs from here
                        const int qn = (q + 1) % W4S_SLOTS, bn = qn % W4S_RING;
                        b02 = f32x4{bq[bn][2], bq[bn][3], bq2[bn][0], bq2[bn][1]};
                        HP3D_OPAQUE_V(b02);          // (the copies happen HERE, not in front of the pair that reads them)
                        // ... and the filter fragment this cout block releases
                        const int t = q + W4S_RING;
                        if (t < W4S_SLOTS) b_fetch(bs, soff_of(t, step, cy));
                        else b_fetch(bs, soff_of(t - W4S_SLOTS, nstep, ncy_));
                    }
                    if (g == 11 && j + 2 < W4S_WP) a_fetch(as, j + 2);               // (its set was consumed by this plane's split one plane ago)
                    if (j < W4S_WINDOW_PLANES && g < 9) {                             // next step's window: 9 loads behind each of the first four planes
                        const int we = W4_ISSUE_ELEM(j * 9 + g);
                        d[we] = W4S_WLOAD(irsrc, (int)((unsigned)ro[we / 6] + (unsigned)co[we % 6]), wsoff);
                    }
                    if (j >= W4S_TRANSFORM_FIRST && j < W4S_TRANSFORM_FIRST + 3 && g % 3 == 0) transform_pass((j - W4S_TRANSFORM_FIRST) * 4 + g / 3);
                    if (j == W4S_WP - 1) {                                            // V of the next step: three values behind each pair of the last plane
#pragma unroll
                        for (int i = 0; i < 3; ++i) v_write(cur ^ 1, g * 3 + i);
                    }
                }
            }
            HP3D_SCHED_BARRIER();
            __syncthreads();             // V[cur^1] complete, V[cur] free
            cur ^= 1;
        };
        {   // the next item of this workgroup, known BEFORE the first step (an item may be a single step)
            if (n_item >= 0) split_of(n_item, n_cy, n_tblock, n_piece, n_s0, n_s1);
            n_cy = HP3D_READFIRSTLANE(n_cy); n_tblock = HP3D_READFIRSTLANE(n_tblock); n_piece = HP3D_READFIRSTLANE(n_piece);
            n_s0 = HP3D_READFIRSTLANE(n_s0); n_s1 = HP3D_READFIRSTLANE(n_s1);
        }
        step_body(s0, std::true_type{});
        // (the tile table of the next item goes into the other parity only now: the barrier that ended the step above tells that every
        //  wave has finished reading that parity in the PREVIOUS item's epilogue)
        table_write(n_tblock, (k + 1) & 1, n_piece);
        for (int step = s0 + 1; step < s1; ++step) step_body(step, std::false_type{});

        // ---- epilogue: the accumulators cross the waves through the V buffer the last step freed, one cout block per pass
        //      X[plane 36][cout 16][tile 32] (the tile quads of a cout row XOR-swizzled by cout >> 1); then thread (tile pair tp, cout n)
        //      runs Y = A^T M A on two tiles at once, bias + leaky-ReLU (+ 2x2 max-pool), NHWC stores (raw sums for a tail piece).
#ifndef HP3D_EMU
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3");      // (MFMA results written inside inline asm: see conv_wino4.hip)
#endif
        float* const X = V + (cur ^ 1) * W4S_VBUF_FLOATS;
        const int en = tid & 15, tp = tid >> 4;
        const int* tab = tinfo + (k & 1) * 2 * W4S_TILES;
        const float slope = p.act ? HP3D_LEAKY_SLOPE : 1.f;
        const bool full = HP3D_OPAQUE_SGPR((((p.Ho | p.Wo) & 3) == 0 || rawp) ? 1 : 0) != 0;
        const hp3d_rsrc_t orsrc = HP3D_MAKE_RSRC(rawp ? (float*)p.partial : p.out, rawp ? 2u * gridDim.x * (unsigned)(W4S_PIECE_FLOATS * 4) : out_bytes);
        const int srow = rawp ? 4 * W4S_COUTS * 4 : Ws * p.out_cs * 4, scol = rawp ? W4S_COUTS * 4 : p.out_cs * 4;      // byte strides of the 4x4 block
        const int xw = (wplane0 * 16 + ln) * W4S_TILES, xsw = (ln >> 1) & 7;
        const int xr = en * W4S_TILES + 4 * ((tp >> 1) ^ ((en >> 1) & 7)) + 2 * (tp & 1);
        int toff[2], tfl[2];
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) { toff[mm] = tab[2 * tp + mm]; tfl[mm] = tab[W4S_TILES + 2 * tp + mm]; }
#pragma unroll
        for (int c = 0; c < W4S_CB; ++c) {
#pragma unroll
            for (int j = 0; j < W4S_WP; ++j)
#pragma unroll
                for (int m = 0; m < 2; ++m) *(f32x4*)(X + xw + j * 16 * W4S_TILES + 4 * ((4 * m + lq) ^ xsw)) = M[j][m][c];
            __syncthreads();
            f32x2 mv[W4_NP];
#pragma unroll
            for (int pl = 0; pl < W4_NP; ++pl) mv[pl] = *(const f32x2*)(X + pl * 16 * W4S_TILES + xr);
            __syncthreads();             // (the next pass, or the next item's last plane, overwrites X)
            const int cout = cy * W4S_COUTS + c * 16 + en;
            const float bias = rawp ? 0.f : p.bias[cout];
            const bool cok = rawp || cout < p.cout_store;
            const int cout_off = rawp ? c * 16 + en : cout;                 // a piece holds the item's 64 couts only
            f32x2 z[6][4];                               // A^T M: along the plane rows a
#pragma unroll
            for (int b = 0; b < 6; ++b)
                w4_at_t<f32x2>(mv[0 * 6 + b], mv[1 * 6 + b], mv[2 * 6 + b], mv[3 * 6 + b], mv[4 * 6 + b], mv[5 * 6 + b], z[b][0], z[b][1], z[b][2], z[b][3]);
            f32x2 yy[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w4_at_t<f32x2>(z[0][i], z[1][i], z[2][i], z[3][i], z[4][i], z[5][i], yy[i][0], yy[i][1], yy[i][2], yy[i][3]);
                if (!POOL && !rawp) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        f32x2 x = yy[i][jj] + bias;
                        const f32x2 sx = slope * x;          // (slope 1 = a linear layer: max(x, x))
                        yy[i][jj] = f32x2{fmaxf(x[0], sx[0]), fmaxf(x[1], sx[1])};
                    }
                }
            }
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                const int off = toff[mm], fl = tfl[mm];
                const int vo = (cok && off >= 0) ? (off + cout_off) * 4 : OOR;
                auto store_tile = [&](auto full_tag) {
                    constexpr bool FULL = decltype(full_tag)::value;       // no edge selects: every store of the tile goes to `vo`
                    if (POOL && !rawp) {
#pragma unroll
                        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                            for (int pj = 0; pj < 2; ++pj) {
                                // bias + leaky-ReLU AFTER the max (both monotonic: the same bits as activating first; conv_wino4.hip)
                                float v = fmaxf(fmaxf(yy[2 * pi][2 * pj][mm], yy[2 * pi][2 * pj + 1][mm]), fmaxf(yy[2 * pi + 1][2 * pj][mm], yy[2 * pi + 1][2 * pj + 1][mm])) + bias;
                                v = fmaxf(v, slope * v);
                                const bool ok = FULL || ((pj == 0 || (fl & 1)) && (pi == 0 || (fl & 2)));
                                HP3D_BUFFER_STORE4(orsrc, v, ok ? vo : OOR, (pi * Ws + pj) * p.out_cs * 4);
                            }
                    } else {
                        const int vr = fl & 15, vc = fl >> 4;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int vrow = (FULL || i < vr) ? vo : OOR;
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) HP3D_BUFFER_STORE4(orsrc, yy[i][jj][mm], (FULL || jj < vc) ? vrow : OOR, i * srow + jj * scol);
                        }
                    }
                };
                if (!POOL && full) store_tile(std::true_type{});
                else store_tile(std::false_type{});
            }
        }
        if (n_item < 0) break;
        item = n_item; cy = n_cy; tblock = n_tblock; piece = n_piece; s0 = n_s0; s1 = n_s1;
    }
}

// Tail pieces -> outputs (wino4_shared.h: w4_tail_reduce_body; the same piece layout as conv_wino4.hip's)
template <bool POOL>
HP3D_KERNEL(256)
void wino4s_tail_reduce_kernel(const ConvParams p) { w4_tail_reduce_body<POOL, W4S_TILES, W4S_COUTS, W4S_CK>(p, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x); }

// float32 -> three bfloat16 pieces, each the round-to-nearest-even of what the previous ones left (x = p0 + p1 + p2 exactly)
inline unsigned short w4s_bf16_rne(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline float w4s_bf16_f32(unsigned short h) { const unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
inline void w4s_split3(float x, unsigned short pc[3]) {
    pc[0] = w4s_bf16_rne(x);
    const float r1 = x - w4s_bf16_f32(pc[0]);
    pc[1] = w4s_bf16_rne(r1);
    const float r2 = r1 - w4s_bf16_f32(pc[1]);
    pc[2] = w4s_bf16_rne(r2);
}

}  // namespace

// U = G g G^T per (cin, cout) exactly as wino4_pack_weights evaluates it (double, rounded once to float32), then split into three
// bfloat16 pieces and laid out in fragment order: [plane 36][step Cin/16][Cout/64][cout block 4]{[q 4][n 16]{U1 e0..3, U0 e0..3} (16 B per
// lane), [q 4][n 16]{U2 e0..3} (8 B per lane)}, channel 16 step + 4 q + e, cout 64 cy + 16 cb + n.  Sizes in BYTES (6 per value).
size_t wino4s_packed_bytes(int cin_pad, int cout_pad) { return (size_t)W4_NP * cin_pad * cout_pad * 6; }

void wino4s_pack_weights(const float* g_hwio, int Cin, int Cout, int cin_pad, int cout_pad, const int* chan_map, void* dst_) {
    const double G[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    unsigned char* dst = (unsigned char*)dst_;
    const int nst = cin_pad / 16, ncy = cout_pad / 64;
    memset(dst, 0, wino4s_packed_bytes(cin_pad, cout_pad));
    // one (channel quad, 16-cout block) at a time: its values are ONE 256-byte run ([U1|U0] of the 16 lanes (q, n = 0..15)) + one 128-byte run (U2) per plane
    // (cout-innermost with a fragment per (plane, cout) scattered every store over 36 cache lines: 110 s for the weight set on 8 host threads)
    for (int eq = 0; eq < cin_pad; eq += 4) {
        int rc[4];
        bool any = false;
        for (int e = 0; e < 4; ++e) {
            rc[e] = chan_map ? chan_map[eq + e] : (eq + e < Cin ? eq + e : -1);
            any |= rc[e] >= 0;
        }
        if (!any) continue;
        const int st = eq >> 4, q = (eq >> 2) & 3;
        for (int c16 = 0; c16 < (Cout + 15) / 16; ++c16) {
            const int cyb = c16 >> 2, cb = c16 & 3;
            unsigned short a16[W4_NP][16][8], a8[W4_NP][16][4];       // [plane][n]{U1 e0..3, U0 e0..3}, {U2 e0..3}
            memset(a16, 0, sizeof(a16));
            memset(a8, 0, sizeof(a8));
            for (int n = 0; n < 16; ++n) {
                const int co = 16 * c16 + n;
                if (co >= Cout) break;
                for (int e = 0; e < 4; ++e) {
                    if (rc[e] < 0) continue;
                    double w3[3][3];
                    for (int r = 0; r < 3; ++r)
                        for (int c = 0; c < 3; ++c) w3[r][c] = (double)g_hwio[((size_t)(r * 3 + c) * Cin + rc[e]) * Cout + co];
                    for (int a = 0; a < 6; ++a) {
                        double ga[3];
                        for (int c = 0; c < 3; ++c) ga[c] = G[a][0] * w3[0][c] + G[a][1] * w3[1][c] + G[a][2] * w3[2][c];
                        for (int bb = 0; bb < 6; ++bb) {
                            unsigned short pc[3];
                            w4s_split3((float)(ga[0] * G[bb][0] + ga[1] * G[bb][1] + ga[2] * G[bb][2]), pc);
                            a16[a * 6 + bb][n][e] = pc[1]; a16[a * 6 + bb][n][4 + e] = pc[0]; a8[a * 6 + bb][n][e] = pc[2];
                        }
                    }
                }
            }
            for (int pl = 0; pl < W4_NP; ++pl) {
                unsigned char* frag = dst + ((((size_t)pl * nst + st) * ncy + cyb) * W4S_CB + cb) * W4S_FRAG_BYTES;
                memcpy(frag + q * 256, a16[pl], 256);
                memcpy(frag + 64 * 16 + q * 128, a8[pl], 128);
            }
        }
    }
}

// Returns 1 when the layer can run here: a 3x3 / stride-1 layer with Cin a multiple of 16 and Cout of 64; *filled (may be NULL) says whether
// the launch has an item per CU (there is no channel split on this kernel: the executor leaves under-filled launches on conv_wino4 / conv_wino2).
int conv_wino4s_eligible(int k, int stride, int Cin, int Cout, int Ho, int Wo, int B, int in_cs, int out_cs, int pool, int* filled) {
    if (filled) *filled = 0;
    if (k != 3 || stride != 1 || Cin % W4S_CK || Cout % W4S_COUTS) return 0;
    if ((long)B * Ho * Wo * in_cs * 4 >= (1L << 30) || (long)B * Ho * Wo * out_cs * 4 >= (1L << 31)) return 0;
    if ((size_t)W4_NP * Cin * Cout * 6 >= ((size_t)1 << 32)) return 0;
    if (pool && ((Ho | Wo) & 1)) return 0;
    const long tiles = (long)B * ((Ho + 3) / 4) * ((Wo + 3) / 4);
    const long items = (tiles + W4S_TILES - 1) / W4S_TILES * (Cout / W4S_COUTS);
    if (filled) *filled = items >= hp3d_num_cus() ? 1 : 0;
    return 1;
}

size_t conv_wino4s_tail_floats() { return (size_t)2 * hp3d_num_cus() * W4S_PIECE_FLOATS; }

template <bool POOL>
static void wino4s_launch_t(const ConvParams& p, long tiles, hipStream_t s) {
    static bool attr_done[64] = {};
    auto k = conv_wino4s_kernel<POOL>;
    if (hp3d_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, W4S_SMEM_BYTES);
    const long items = (tiles + W4S_TILES - 1) / W4S_TILES * (p.Cout / W4S_COUTS);
    const int slots = hp3d_num_cus();                     // persistent grid: one workgroup per CU
    dim3 grid((unsigned)((items < slots && !(p.tail_items > 0)) ? items : slots));
    HP3D_LAUNCH(k, grid, dim3(256), W4S_SMEM_BYTES, s, p);
}

// pin.wpk = the split filters (wino4s_pack_weights).  Returns < 0: refused; 0: launched; 1: launched AND the last round ran as tail pieces.
int conv_wino4s_launch(const ConvParams& pin, int pool, hipStream_t s) {
    if ((long)pin.B * pin.H * pin.W * pin.in_cs * 4 >= (1L << 30) || (long)pin.B * pin.Ho * pin.Wo * pin.out_cs * 4 >= (1L << 31)) return -1;
    if (pin.Cout % W4S_COUTS || pin.Cin % W4S_CK || (size_t)W4_NP * pin.Cin * pin.Cout * 6 >= ((size_t)1 << 32)) return -1;
    ConvParams p = pin;
    p.tiles_x = (p.Wo + 3) / 4;
    p.tiles_y = (p.Ho + 3) / 4;
    const long tiles = (long)p.B * p.tiles_x * p.tiles_y;
    if (pool && ((p.Ho | p.Wo) & 1)) return -1;
    p.ksplit = 1;
    p.nsub = 1;
    p.tail_items = p.tail_q = 0;
    if (p.partial && p.partial_cap >= conv_wino4s_tail_floats() && !(pool && (p.cout_store & 3)) && (p.out_cs & 3) == 0 && ((uintptr_t)p.out & 15) == 0)
        p.tail_q = conv_wino4_tail_plan(p.Cin, p.Cout, p.Ho, p.Wo, p.B, &p.tail_items);
    if (pool) wino4s_launch_t<true>(p, tiles, s);
    else wino4s_launch_t<false>(p, tiles, s);
    if (p.tail_items > 0) {
        const long total = (long)p.tail_items * W4S_TILES * (pool ? 4 : 16) * (W4S_COUTS / 4);
        const unsigned blocks = (unsigned)((total + 255) / 256);
        if (pool) HP3D_LAUNCH(wino4s_tail_reduce_kernel<true>, dim3(blocks), dim3(256), 0, s, p);
        else HP3D_LAUNCH(wino4s_tail_reduce_kernel<false>, dim3(blocks), dim3(256), 0, s, p);
        return 1;
    }
    return 0;
}

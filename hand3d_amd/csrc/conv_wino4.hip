// conv_wino4.hip -- Winograd F(4x4, 3x3) on the f32 matrix cores (round 3; plane loop, window issue order, tail pieces: round 4).
//
// Same call sites as conv_wino.hip / conv_wino2.hip (NetworkOps.conv_relu + max_pool, utils/general.py:36-65; the 3x3 /
// stride-1 layers and the 7x7 layers of PoseNet2D, nets/ColorHandPose3DNetwork.py:170-219, as nine 3x3 blocks), with the
// larger Winograd tile: a 6x6 input window gives 4x4 outputs through 36 element-wise products per (cin, cout) -- 2.25
// multiply-adds per output instead of 4 (F(2x2,3x3)) or 9 (direct).  Float32 throughout; the transforms now multiply by
// 2, 4, 5, 8 and the transformed filters by 1/4 .. 1/24, which costs 3.5x the rounding error of F(2x2,3x3) END TO END
// (heat-maps 6e-6 against a gate of 1e-3, 3-D keypoints 3e-6 against 1e-4: profiles/r03_tuning_notes.md sections 5 and 7, where the
// effect on the thresholded hand mask is measured too).  The executor gives it every filled 3x3 / 7x7 trunk launch (option "wino4").
//
//   Y(4x4) = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A          d: 6x6 window, g: 3x3 filter, points {0, +-1, +-2, inf}
//
// Machine shape: conv_wino2.hip's (v_mfma_f32_16x16x4_f32, a wave = 32 tiles x 16 couts, a workgroup = 32 tiles x 64
// couts in 16-channel steps, K order permuted so that A and B are one 16-byte access per four MFMAs, weights global ->
// VGPR in fragment order) with 36 planes instead of 16:
//   * 36 planes x 2 tile halves x 4 = 288 accumulators per lane -> one wave per SIMD (__launch_bounds__(256, 1): 512
//     registers per lane); planes 0..31 are PINNED to the 256 AGPRs and planes 32..35 to arch VGPRs through the constraint of the
//     inline-asm statement that holds a plane's eight MFMAs (left alone, hipcc moved ~10 tuples per step between the files);
//   * V = B^T d B double buffered in LDS: 2 x 36 planes x 32 tiles x 16 channels = 147 KB;
//   * loader thread = (tile, channel pair): 36 window loads of 8 bytes; the offsets are 6 row + 6 column terms (a row /
//     column outside the image carries a constant that pushes the sum out of the buffer's range: reads as 0) added at
//     issue time, not 36 registers; the loads are spread two per plane over the first 18 planes of a step, in CLASS order: the
//     elements (r, c), (r + 4, c), (r, c + 4), (r + 4, c + 4) -- the same pixels seen from neighbouring tiles -- follow each other, so
//     the repeats hit the L1 line or its pending fill (round 4: +2.7 %);
//   * a 3x3 plane is FOUR statements of two MFMAs with one of the plane's memory instructions behind each (next plane's A fragments |
//     window | window | weight fragment): each issues under 64 matrix-core cycles instead of all at the plane boundary (round 4: +1.6 %;
//     where the time goes was measured with -DHP3D_W4_TIMING=1, profiles/r04_sq_counters.md);
//   * weights [36 planes][step][Cout/16][q][n][e] through a ring of 9 planes; work items in XCD-affine order (the cout blocks of a
//     tile block run on one XCD, so its windows cross the fabric once per XCD);
//   * a tile is 16 output pixels: the fused 2x2 max-pool takes four maxima per tile; ragged image edges (Ho, Wo not a
//     multiple of 4) drop rows / columns through out-of-range store offsets (no edge selects at all when Ho, Wo are multiples of 4);
//   * a last round of items that is not full is cut into equal runs of item-steps, one per CU (TAIL, below).
#include "hp3d_common.h"
#include "wino4_shared.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

#define W4_WLOAD HP3D_BUFFER_LOAD8
// -DHP3D_W4_TIMING=1: diagnostic build -- every wave sums shader-clock intervals of its steps into w4_timing[] and conv_wino4_launch prints
// them (wino4_diag.h; profiles/r04_tuning_notes.md section 4).  The shipped build sees empty macros.  (Round 4's timing ABLATIONS -- builds
// that compute wrong results on purpose, HP3D_W4_ABL -- were removed from this file in round 5: all settled, recorded in r03 / r04_tuning_notes.md.)
#include "wino4_diag.h"

namespace {

constexpr int W4_TILES = 32;                       // Winograd tiles (4x4 outputs each) per item
constexpr int W4_CK = 16;                          // channels per step
constexpr int W4_COUTS = 64;                       // output channels per item (16 per wave)
constexpr int W4_PLANE_FLOATS = W4_TILES * W4_CK;  // one plane of one V buffer: 2 KB
constexpr int W4_VBUF_FLOATS = W4_NP * W4_PLANE_FLOATS;
constexpr int W4_SMEM_BYTES = 2 * W4_VBUF_FLOATS * 4 + 2 * 2 * W4_TILES * 4;     // 2 V buffers + two tile tables = 147968 B
#ifndef HP3D_W4_RING
#define HP3D_W4_RING 9
#endif
constexpr int W4_RING = HP3D_W4_RING;              // weight fragments in flight (planes); must divide 36
static_assert(W4_NP % W4_RING == 0, "static ring slots need a ring that divides the plane count");
constexpr int W4_ADEPTH = 2;                       // A fragments (planes) in flight from LDS: a plane is only 8 MFMAs = 256 cycles, and with one wave
static_assert(W4_NP % W4_ADEPTH == 0, "");         // per SIMD an LDS round trip that is not covered idles the matrix core
constexpr int W4_HALF = 18;                        // planes reachable from one LDS base (16-bit immediate offsets)
constexpr int W4_AGPR_PLANES = 32;                 // planes whose accumulators live in AGPRs (32 x 8 = 256); the rest in arch VGPRs
constexpr int W4_WPP = 2;                          // window loads per plane: the 36 loads of the next step's window are SPREAD over the first
                                                   // 36 / W4_WPP planes.  Loads return in issue order, so a weight fragment issued behind a burst of
                                                   // 36 x 4 waves window loads waits for the whole burst to pass the CU's one address unit
                                                   // (measured: the wave stalled at plane 9 of every step, 25 % of the kernel's time)
constexpr int W4_TRANSFORM_AT = 29;               // the plane under which the next step's windows are transformed

// (issue order of a step's 36 window loads: W4_ISSUE_ELEM, wino4_shared.h)

// same quad swizzle as conv_wino2.hip (the V row of a tile is 16 channels = four 16-byte quads)
__device__ __forceinline__ int w4_swz(int t) { return (0x78 >> (((t >> 2) & 3) * 2)) & 3; }

using W4Geom = W4GeomT<W4_TILES, W4_COUTS>;          // tile / item geometry (wino4_shared.h)
constexpr int W4_PIECE_FLOATS = W4_TILES * 16 * W4_COUTS;       // raw 4x4 outputs of one item: [tile 32][pixel 16][cout 64] = 128 KB

// TAIL (3x3 filters, no whole-launch channel split): the items of a launch are dealt round-robin to one workgroup per CU; a last round
// that is not full (HandSegNet's 40x40 layers at B = 32: 800 items on 256 CUs = 3.125 rounds, paid as 4) is shared out stream-K
// fashion: its `tail_items` items x nsteps channel steps are ONE run of item-steps, cut into equal runs of `tail_q` steps, one per
// workgroup.  A run lies in one item or crosses into the next: at most two PIECES per workgroup, whose raw sums (the output
// transform is linear) go to a compact scratch [workgroup][piece 2][tile][pixel][cout]; wino4_tail_reduce adds an item's pieces in
// step order, + bias, activation (+ pool), and stores.  Deterministic; only the summation order of the tail items changes.
template <bool POOL, int NSUB, bool SPLITK, bool TAIL>
HP3D_KERNEL2(256, 1)
void conv_wino4_kernel(const ConvParams p) {
    static_assert(!(POOL && SPLITK), "the fused max-pool needs complete sums");
    static_assert(!(TAIL && (SPLITK || NSUB != 1)), "tail pieces: plain 3x3 launches only");
    constexpr bool VARSTEPS = SPLITK || TAIL;         // items own a RANGE of the channel steps
    HP3D_DYN_SMEM(V);
    int* tinfo = (int*)(V + 2 * W4_VBUF_FLOATS);       // [parity][0..31] output offset of tile t (-1: none), [32..63] edge flags
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = HP3D_READFIRSTLANE(tid >> 6);
    const int ln = lane & 15, lq = lane >> 4;          // MFMA column (cout / tile in block) and k slot

    const W4Geom geo(p);
    const int per_split = geo.tile_blocks * geo.ncy;
    const int nitems = per_split * (SPLITK ? p.ksplit : 1);
    // TAIL: virtual item ids [0, nfull) are whole items; nfull + 2 w + j = piece j of workgroup w's run of the tail's item-steps
    const int nfull = TAIL ? nitems - p.tail_items : nitems;
    auto tile_decode = [&](int id, int& tb, int& tyy, int& txx) { geo.tile_decode(id, tb, tyy, txx); };
    const int Hs = POOL ? (p.Ho >> 1) : p.Ho, Ws = POOL ? (p.Wo >> 1) : p.Wo;
    // tile table.  Plain: offset of output (4 ty, 4 tx), flags = valid rows | valid columns << 4 (1..4 each).  Pooled: offset of
    // pooled output (2 ty, 2 tx), flags = bit 0: column 2 tx + 1 exists, bit 1: row 2 ty + 1 exists.
    auto table_write = [&](int tblock, int parity, int kz, int piece) {
        if (tid < W4_TILES) {
            int tb, tyy, txx;
            tile_decode(tblock * W4_TILES + tid, tb, tyy, txx);
            int off = -1, fl = 0;
            if (TAIL && piece >= 0) {          // raw piece: every tile's 4x4 outputs, compact
                off = piece * W4_PIECE_FLOATS + tid * (16 * W4_COUTS);
                fl = 4 | (4 << 4);
            } else if (tb < p.B) {
                if (POOL) {
                    if (2 * tyy < Hs && 2 * txx < Ws) {
                        off = ((tb * Hs + 2 * tyy) * Ws + 2 * txx) * p.out_cs;
                        fl = (2 * txx + 1 < Ws ? 1 : 0) | (2 * tyy + 1 < Hs ? 2 : 0);
                    }
                } else {
                    off = (((SPLITK ? kz * p.B + tb : tb) * Hs + 4 * tyy) * Ws + 4 * txx) * p.out_cs;
                    fl = min(4, Hs - 4 * tyy) | (min(4, Ws - 4 * txx) << 4);
                }
            }
            tinfo[parity * 2 * W4_TILES + tid] = off;
            tinfo[parity * 2 * W4_TILES + W4_TILES + tid] = fl;
        }
    };

    // ---- loader role: this thread transforms the 6x6 window of tile lt for channel pair lp ------------------------------
    const int lt = tid >> 3, lp = tid & 7;
    const int cs4 = p.in_cs * 4;
    constexpr int OOR = (int)0x80000000;          // row outside the image / no such tile
    constexpr int COL_OOR = 0x60000000;           // column outside the image: any row term + this is >= 2^30 > the buffer's extent
    int ro[6], co[6];
    int cb = 0, cty = 0, ctx_ = 0;                // NSUB = 9 only: tile coordinates stay live for the block shifts
    auto window_offsets = [&](bool valid, int lb, int lty, int ltx, int sub) {
        const int dy = NSUB == 1 ? 0 : 3 * (sub / 3) - 2, dx = NSUB == 1 ? 0 : 3 * (sub % 3) - 2;
        const int wy0 = 4 * lty - 1 + dy, wx0 = 4 * ltx - 1 + dx;
        const int wbase = ((lb * p.H + wy0) * p.W + wx0) * cs4 + lp * 8;
        const bool tv = valid && lb < p.B;
#pragma unroll
        for (int r = 0; r < 6; ++r) ro[r] = (tv && (unsigned)(wy0 + r) < (unsigned)p.H) ? wbase + r * (p.W * cs4) : OOR;
#pragma unroll
        for (int c = 0; c < 6; ++c) co[c] = (unsigned)(wx0 + c) < (unsigned)p.W ? c * cs4 : COL_OOR;
    };
    auto loader_setup = [&](int tblock, bool valid, int sub) {
        int lb, lty, ltx;
        tile_decode(tblock * W4_TILES + lt, lb, lty, ltx);
        if (NSUB > 1) { cb = valid ? lb : p.B; cty = lty; ctx_ = ltx; }
        window_offsets(valid, lb, lty, ltx, NSUB > 1 ? sub : 0);
    };
    auto loader_shift = [&](int sub) { window_offsets(true, cb, cty, ctx_, sub); };
    const hp3d_rsrc_t irsrc = HP3D_MAKE_RSRC(p.in, (unsigned)p.B * (unsigned)(p.H * p.W) * (unsigned)cs4);
    [[maybe_unused]] const unsigned out_bytes = (unsigned)(SPLITK ? p.ksplit * p.B : p.B) * (unsigned)(Hs * Ws) * (unsigned)p.out_cs * 4u;

    f32x2 d[36];
    auto window_fetch = [&](int soff) {
#pragma unroll
        for (int e = 0; e < 36; ++e) d[e] = W4_WLOAD(irsrc, (int)((unsigned)ro[e / 6] + (unsigned)co[e % 6]), soff);
    };
    float* const Vw = V + lt * W4_CK + ((lp >> 1) ^ w4_swz(lt)) * 4 + (lp & 1) * 2;      // this thread's slot in plane 0 of buffer 0
    // B^T d B in place: along the window rows first (plane row a), then along the columns (plane column b); afterwards d[a * 6 + b] is this
    // thread's value of plane a * 6 + b
    auto transform_arith = [&]() {
#pragma unroll
        for (int c = 0; c < 6; ++c)
            w4_bt(d[0 * 6 + c], d[1 * 6 + c], d[2 * 6 + c], d[3 * 6 + c], d[4 * 6 + c], d[5 * 6 + c]);
#pragma unroll
        for (int a = 0; a < 6; ++a)
            w4_bt(d[a * 6 + 0], d[a * 6 + 1], d[a * 6 + 2], d[a * 6 + 3], d[a * 6 + 4], d[a * 6 + 5]);
    };
    auto v_write = [&](int buf, int pl) {
        float* Vq0 = Vw + buf * W4_VBUF_FLOATS;
        float* dst = pl < W4_HALF ? Vq0 + pl * W4_PLANE_FLOATS : Vq0 + W4_HALF * W4_PLANE_FLOATS + (pl - W4_HALF) * W4_PLANE_FLOATS;
        *(f32x2*)dst = d[pl];
    };
    auto transform_commit = [&](int buf) {
        transform_arith();
#pragma unroll
        for (int pl = 0; pl < W4_NP; ++pl) v_write(buf, pl);
    };

    // ---- MFMA role -------------------------------------------------------------------------------------------------------
    // packed U: [plane 36][step (NSUB * Cin / 16)][Cout/16][q 4][n 16][e 4]: the fragment a wave needs for one (plane, step) is
    // 1 KB, lane-linear: base = one scalar offset per (plane, step)
    const int CO16 = p.Cout >> 4;
    const int nsub_rt = NSUB == 1 ? 1 : p.nsub;
    const int csteps = p.Cin / W4_CK;
    const int nsteps = nsub_rt * csteps;
    const hp3d_rsrc_t wrsrc = HP3D_MAKE_RSRC(p.wpk, (unsigned)(W4_NP * nsub_rt * p.Cin) * (unsigned)p.Cout * 4u);
    const int step_stride_b = CO16 * 1024;
    const int plane_stride_b = nsteps * step_stride_b;
    auto soff_of = [&](int plane, int step) { return plane * plane_stride_b + step * step_stride_b; };

    f32x4 M[W4_NP][2];     // [plane][tile half]: rows = tiles 16 m + 4 (lane >> 4) + r, column = cout (lane & 15)
    f32x4 bq[W4_RING];
    auto b_fetch = [&](int slot, int voff, int soff) { bq[slot] = HP3D_BUFFER_LOAD16(wrsrc, voff, soff); };
    const int va_lane = (ln * W4_CK + ((lq ^ w4_swz(ln)) * 4)) * 4;
    int ab0 = 0, ab1 = 0;
    f32x4 af[W4_ADEPTH][2];
    auto a_fetch = [&](int set, int plane) {
        const int base = plane < W4_HALF ? ab0 : ab1, pl = plane < W4_HALF ? plane : plane - W4_HALF;
#pragma unroll
        for (int m = 0; m < 2; ++m) af[set][m] = *(const f32x4*)((const char*)V + base + (pl * W4_PLANE_FLOATS + m * 16 * W4_CK) * 4);
    };

    // virtual item id -> cout block, tile block, channel steps [s0_, s1_), piece slot (-1 = a whole item / a slice of the [ksplit] scratch)
    auto split_of = [&](int it, int& kz, int& cy_, int& tb_, int& piece_, int& s0_, int& s1_) {
        piece_ = -1;
        kz = 0;
        if (TAIL && it >= nfull) {
            piece_ = it - nfull;
            const int w = piece_ >> 1;
            const int a = w * p.tail_q, b = min(a + p.tail_q, p.tail_items * nsteps);       // this workgroup's run of item-steps
            const int i0 = a / nsteps;
            if (piece_ & 1) { s0_ = 0; s1_ = b - (i0 + 1) * nsteps; geo.item_decode(nfull + i0 + 1, cy_, tb_); }
            else { s0_ = a - i0 * nsteps; s1_ = min(nsteps, s0_ + b - a); geo.item_decode(nfull + i0, cy_, tb_); }
            return;
        }
        kz = SPLITK ? it / per_split : 0;
        geo.item_decode(SPLITK ? it - kz * per_split : it, cy_, tb_);
        s0_ = SPLITK ? (kz * nsteps) / p.ksplit : 0;
        s1_ = SPLITK ? ((kz + 1) * nsteps) / p.ksplit : nsteps;
    };
    // the workgroup's sequence of virtual items: blockIdx.x, + gridDim.x, ... below nfull, then (TAIL) the one or two pieces of its run
    auto next_of = [&](int it) {
        const int nx = it + (int)gridDim.x;
        if (!TAIL) return nx < nitems ? nx : -1;
        if (it < nfull && nx < nfull) return nx;
        const int w = (int)blockIdx.x, a = w * p.tail_q, tot = p.tail_items * nsteps;
        if (a >= tot) return -1;
        const int b = min(a + p.tail_q, tot), i0 = a / nsteps;
        if (it < nfull) return nfull + 2 * w;
        if (it == nfull + 2 * w && b > (i0 + 1) * nsteps) return nfull + 2 * w + 1;      // the run crosses into the next item
        return -1;
    };
    int item = blockIdx.x;
    if (TAIL && item >= nfull) {                    // no whole item for this workgroup (a launch of less than one round): straight to its run
        if ((int)blockIdx.x * p.tail_q >= p.tail_items * nsteps) return;
        item = nfull + 2 * (int)blockIdx.x;
    }
    int kz, cy, tblock, piece, s0, s1;
    split_of(item, kz, cy, tblock, piece, s0, s1);
    if (VARSTEPS) { kz = HP3D_READFIRSTLANE(kz); cy = HP3D_READFIRSTLANE(cy); tblock = HP3D_READFIRSTLANE(tblock); piece = HP3D_READFIRSTLANE(piece);
                    s0 = HP3D_READFIRSTLANE(s0); s1 = HP3D_READFIRSTLANE(s1); }
    const int sub0 = (NSUB > 1 && SPLITK) ? HP3D_READFIRSTLANE(s0 / csteps) : 0;
    int sub_cur = sub0;                           // block (i, j) = (sub_cur / 3, sub_cur % 3) of the 9x9 extension the current step belongs to
    loader_setup(tblock, true, sub0);
    table_write(tblock, 0, kz, piece);
    int wvoff = (cy * (W4_COUTS / 16) + wave) * 1024 + lane * 16;
    window_fetch((s0 - sub0 * csteps) * (W4_CK * 4));
#pragma unroll
    for (int t = 0; t < W4_RING; ++t) b_fetch(t, wvoff, soff_of(t, s0));
    transform_commit(0);
    __syncthreads();
    int cur = 0;

    W4_T_DECL();
    for (int k = 0;; ++k) {
        int n_cy = cy, n_tblock = tblock, n_wvoff = wvoff, n_kz = kz, n_s0 = s0, n_s1 = s1, n_piece = -1;
        const int n_item = next_of(item);
        const bool raw = TAIL && piece >= 0;          // this item is a tail piece: raw sums into the compact scratch
        const int cout = cy * W4_COUTS + wave * 16 + ln;
        const float bias = (SPLITK || raw) ? 0.f : p.bias[cout];

        auto step_body = [&](int step, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const bool lasts = step + 1 == s1;
            const bool za = sub_cur >= 6, zb = sub_cur == 2 || sub_cur == 5 || sub_cur == 8;
            const int skip_a = (NSUB == 9 && !FIRST) ? HP3D_OPAQUE_SGPR(za ? 1 : 0) : 0;
            const int skip_b = (NSUB == 9 && !FIRST) ? HP3D_OPAQUE_SGPR(zb ? 1 : 0) : 0;
            const int skip_ab = (NSUB == 9 && !FIRST) ? HP3D_OPAQUE_SGPR((za || zb) ? 1 : 0) : 0;
            const int nvoff = lasts ? n_wvoff : wvoff;
            const int nstep = lasts ? (VARSTEPS ? n_s0 : 0) : step + 1;
            W4_T_MARK(4);      // (item switch / epilogue / step prologue)
            ab0 = cur * (W4_VBUF_FLOATS * 4) + va_lane;
            ab1 = ab0 + W4_HALF * W4_PLANE_FLOATS * 4;
            HP3D_OPAQUE_V(ab0);
            HP3D_OPAQUE_V(ab1);
#pragma unroll
            for (int t = 0; t < W4_ADEPTH - 1; ++t) a_fetch(t, t);
            const int nsub_ = NSUB == 1 ? 0 : SPLITK ? HP3D_READFIRSTLANE(nstep / csteps) : lasts ? 0 : (step + 1) / csteps;
            const int ncs = NSUB == 1 ? nstep : nstep - nsub_ * csteps;
            if (lasts) loader_setup(n_tblock, n_item >= 0, nsub_);
            else if (NSUB > 1 && ncs == 0) loader_shift(nsub_);
            const int wstep_b = W4_CK * 4;
            const int wsoff = NSUB > 1 ? HP3D_READFIRSTLANE(ncs * wstep_b) : ncs * wstep_b;
            // The 36 window addresses (row term + column term) in ONE block here, at the top of the step, where the wave waits for its first A
            // fragments anyway -- not one v_add_u32 in front of each load between the MFMA pairs: a float32 MFMA and a VALU instruction of the same
            // wave do not overlap, and a lone VALU result feeding a load address between MFMAs costs 15 ns in isolation (profiles/r06_tuning_notes.md
            // section 7; in situ the adds were worth 1.0-1.5 %).  An address lives in the register pair its load fills.
            int wa[36];
            if (NSUB == 1) {
#pragma unroll
                for (int e = 0; e < 36; ++e) { wa[e] = (int)((unsigned)ro[e / 6] + (unsigned)co[e % 6]); HP3D_OPAQUE_V(wa[e]); }
            }
            // The plane as four PAIRS of MFMAs (one k quad, both tile halves) with ONE of the plane's other instructions after each: the next
            // plane's A fragments | window load | window load | the weight fragment for the slot this plane releases.  Each issues under the
            // 64 matrix-core cycles of the pair in front of it; as one block of eight with everything at the plane boundary the wave spent
            // 40 (windows in L2) to 80 (windows cold) idle matrix-core cycles per plane issuing them (section 4 of the round-4 log).
            // 7x7 filters keep the block form: their skip test would run four times per plane (measured 2-3 % slower).
#pragma unroll
            for (int pl = 0; pl < W4_NP; ++pl) {
                if (NSUB == 1) {
                    const int skip = (NSUB == 9 && !FIRST) ? ((pl / 6 == 5 && pl % 6 == 5) ? skip_ab : pl / 6 == 5 ? skip_a : pl % 6 == 5 ? skip_b : 0) : 0;
                    const int as = pl % W4_ADEPTH, bs = pl % W4_RING;
    #pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        HP3D_SCHED_BARRIER();
                        if (FIRST && e == 0) {
                            if (pl < W4_AGPR_PLANES) HP3D_MFMA16_PAIR_FIRST("a", M[pl][0], M[pl][1], af[as][0][e], af[as][1][e], bq[bs][e]);
                            else HP3D_MFMA16_PAIR_FIRST("v", M[pl][0], M[pl][1], af[as][0][e], af[as][1][e], bq[bs][e]);
                        } else if (NSUB == 9 && !FIRST) {
                            if (pl < W4_AGPR_PLANES) HP3D_MFMA16_PAIR_UNLESS("a", M[pl][0], M[pl][1], af[as][0][e], af[as][1][e], bq[bs][e], skip);
                            else HP3D_MFMA16_PAIR_UNLESS("v", M[pl][0], M[pl][1], af[as][0][e], af[as][1][e], bq[bs][e], skip);
                        } else {
                            if (pl < W4_AGPR_PLANES) HP3D_MFMA16_PAIR("a", M[pl][0], M[pl][1], af[as][0][e], af[as][1][e], bq[bs][e]);
                            else HP3D_MFMA16_PAIR("v", M[pl][0], M[pl][1], af[as][0][e], af[as][1][e], bq[bs][e]);
                        }
                        HP3D_SCHED_BARRIER();
                        if (e == 0) {
                            if (pl + W4_ADEPTH - 1 < W4_NP) a_fetch((pl + W4_ADEPTH - 1) % W4_ADEPTH, pl + W4_ADEPTH - 1);
                        } else if (e == 3) {
                            // weight prefetch W4_RING planes ahead into the slot this plane just released
                            const int t = pl + W4_RING;
                            if (t < W4_NP) b_fetch(t % W4_RING, wvoff, soff_of(t, step));
                            else b_fetch(t % W4_RING, nvoff, soff_of(t - W4_NP, nstep));
                        } else if (pl > W4_TRANSFORM_AT) {
                            // V of the next step: the 36 values this thread transformed under plane 29 go to LDS three behind each of the two
                            // middle pairs of planes 30..35 (as one burst under plane 29 the four waves queued 74 KB on the LDS port at once and
                            // each sat ~900 cycles in front of its next MFMA: timing ablations, round-4 log section 5)
                            constexpr int PER = 36 / (2 * (W4_NP - 1 - W4_TRANSFORM_AT));
                            static_assert(PER * 2 * (W4_NP - 1 - W4_TRANSFORM_AT) == 36, "");
#pragma unroll
                            for (int j = 0; j < PER; ++j) v_write(cur ^ 1, ((pl - W4_TRANSFORM_AT - 1) * 2 + (e - 1)) * PER + j);
                        } else {
                            static_assert(W4_WPP % 2 == 0, "the plane's window loads go behind its two middle pairs, half each");
#pragma unroll
                            for (int j = 0; j < W4_WPP / 2; ++j) {
                                const int wk = pl * W4_WPP + (e - 1) * (W4_WPP / 2) + j;
                                if (wk < 36) {
                                    const int we = W4_ISSUE_ELEM(wk < 36 ? wk : 0);
                                    d[we] = W4_WLOAD(irsrc, wa[we], wsoff);
                                }
                            }
                        }
                    }
                } else {
                    HP3D_SCHED_BARRIER();
                    if (pl + W4_ADEPTH - 1 < W4_NP) a_fetch((pl + W4_ADEPTH - 1) % W4_ADEPTH, pl + W4_ADEPTH - 1);
                    // the eight MFMAs of the plane (two tile halves alternating: 40-cycle dependent latency vs 32-cycle issue) as ONE
                    // statement that pins the accumulators' register file: planes 0..31 in the 256 AGPRs, planes 32..35 in arch VGPRs.
                    // 7x7 filters: in the edge blocks of the zero-extended 9x9 filter (i = 2 / j = 2: one filter row / column of three)
                    // G g G^T has a zero row a = 5 / column b = 5: those planes' MFMAs are skipped (289 instead of 324 plane-steps).
                    const int skip = (NSUB == 9 && !FIRST) ? ((pl / 6 == 5 && pl % 6 == 5) ? skip_ab : pl / 6 == 5 ? skip_a : pl % 6 == 5 ? skip_b : 0) : 0;
                    if (FIRST) {
                        if (pl < W4_AGPR_PLANES) HP3D_MFMA16_PLANE_FIRST("a", M[pl][0], M[pl][1], af[pl % W4_ADEPTH][0], af[pl % W4_ADEPTH][1], bq[pl % W4_RING]);
                        else HP3D_MFMA16_PLANE_FIRST("v", M[pl][0], M[pl][1], af[pl % W4_ADEPTH][0], af[pl % W4_ADEPTH][1], bq[pl % W4_RING]);
                    } else {
                        if (pl < W4_AGPR_PLANES) HP3D_MFMA16_PLANE_UNLESS("a", M[pl][0], M[pl][1], af[pl % W4_ADEPTH][0], af[pl % W4_ADEPTH][1], bq[pl % W4_RING], skip);
                        else HP3D_MFMA16_PLANE_UNLESS("v", M[pl][0], M[pl][1], af[pl % W4_ADEPTH][0], af[pl % W4_ADEPTH][1], bq[pl % W4_RING], skip);
                    }
                    // weight prefetch W4_RING planes ahead into the slot this plane just released
                    const int t = pl + W4_RING;
                    if (t < W4_NP) b_fetch(t % W4_RING, wvoff, soff_of(t, step));
                    else b_fetch(t % W4_RING, nvoff, soff_of(t - W4_NP, nstep));
                    if (pl * W4_WPP < 36) {
    #pragma unroll
                        for (int j = 0; j < W4_WPP; ++j) {        // (indices are constants once the plane loop is unrolled)
                            const int e = W4_ISSUE_ELEM(pl * W4_WPP + j);
                            d[e] = W4_WLOAD(irsrc, (int)((unsigned)ro[e / 6] + (unsigned)co[e % 6]), wsoff);
                        }
                    }
                    if (pl > W4_TRANSFORM_AT) {          // V of the next step, six values per plane (see the pair form)
    #pragma unroll
                        for (int j = 0; j < 36 / (W4_NP - 1 - W4_TRANSFORM_AT); ++j) v_write(cur ^ 1, (pl - W4_TRANSFORM_AT - 1) * (36 / (W4_NP - 1 - W4_TRANSFORM_AT)) + j);
                    }
                }
                if (pl == W4_TRANSFORM_AT) {
                    W4_T_WINDOW_WAIT(W4_TRANSFORM_AT + 1 - 36 / W4_WPP);      // (timing build: marks 0 and 5 around the wait for the window data)
                    transform_arith();
                    W4_T_MARK(1);
                }
            }
            HP3D_SCHED_BARRIER();
            W4_T_MARK(2);
            __syncthreads();             // V[cur^1] complete, V[cur] free
            W4_T_MARK(3);
            cur ^= 1;
            sub_cur = nsub_;             // the block of the step that runs next (this item's or the next item's first)
        };
        {   // the next item of this workgroup, known BEFORE the first step: an item may be a single step (a one-step tail piece), whose
            // only step is also the one that prefetches the next item's first windows and weight fragments
            const bool has_next = n_item >= 0;
            if (has_next) split_of(n_item, n_kz, n_cy, n_tblock, n_piece, n_s0, n_s1);
            if (VARSTEPS) { n_kz = HP3D_READFIRSTLANE(n_kz); n_cy = HP3D_READFIRSTLANE(n_cy); n_tblock = HP3D_READFIRSTLANE(n_tblock); n_piece = HP3D_READFIRSTLANE(n_piece);
                            n_s0 = HP3D_READFIRSTLANE(n_s0); n_s1 = HP3D_READFIRSTLANE(n_s1); }
            n_wvoff = (n_cy * (W4_COUTS / 16) + wave) * 1024 + lane * 16;
        }
        step_body(s0, std::true_type{});
        // (the tile table of the next item goes into the other parity only now: the barrier that ended the step above is what tells
        //  that every wave has finished reading that parity in the PREVIOUS item's epilogue)
        table_write(n_tblock, (k + 1) & 1, n_kz, n_piece);
        for (int step = s0 + 1; step < s1; ++step) step_body(step, std::false_type{});

        // ---- epilogue: Y = A^T M A per (tile, cout), bias + leaky-ReLU (+ 2x2 max-pool) + NHWC store; a channel split stores its
        //      raw sums into the [ksplit][B*Ho*Wo][Cout] scratch and conv_splitk_reduce adds them up in split order.
#ifndef HP3D_EMU
        // the accumulators were last written by MFMAs inside inline-asm statements, which the compiler's hazard recogniser cannot see
        // into: an 8-pass MFMA result needs up to 18 idle cycles before a VALU / v_accvgpr read (CDNA3 ISA, "dependency resolution")
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3");
#endif
        const int* tab = tinfo + (k & 1) * 2 * W4_TILES;
        const float slope = p.act ? HP3D_LEAKY_SLOPE : 1.f;
        // every tile of the launch is a whole 4x4 block (pooled: 2x2) when the output extent is a multiple of 4 -- all trunk layers at the
        // network's own sizes: the per-store edge selects then fall away (tiles beyond the batch carry no offset and stay out of range)
        const bool full = HP3D_OPAQUE_SGPR((((p.Ho | p.Wo) & 3) == 0 || raw) ? 1 : 0) != 0;
        const bool cok = raw || cout < p.cout_store;
        const int cout_off = raw ? wave * 16 + ln : cout;                // a piece holds the item's 64 couts only
        const hp3d_rsrc_t orsrc = HP3D_MAKE_RSRC(raw ? (float*)p.partial : p.out, raw ? 2u * gridDim.x * (unsigned)(W4_PIECE_FLOATS * 4) : out_bytes);
        const int srow = raw ? 4 * W4_COUTS * 4 : Ws * p.out_cs * 4, scol = raw ? W4_COUTS * 4 : p.out_cs * 4;      // byte strides of the 4x4 block
        // The two tile halves m = 0, 1 of an accumulator pair (same register index r, same cout) go through A^T . A as ONE packed value: the
        // output transform is 100 additions / multiply-adds per (tile, cout), and an instruction costs the f32 matrix pipe the same 4-5 cycles
        // whether it carries one float or two (round 5: the pooled epilogue was all scalar -- 832 transform instructions per item and wave)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x2 z[6][4];                               // A^T M: along the plane rows a
#pragma unroll
            for (int b = 0; b < 6; ++b)
                w4_at_t<f32x2>(f32x2{M[0 * 6 + b][0][r], M[0 * 6 + b][1][r]}, f32x2{M[1 * 6 + b][0][r], M[1 * 6 + b][1][r]}, f32x2{M[2 * 6 + b][0][r], M[2 * 6 + b][1][r]},
                               f32x2{M[3 * 6 + b][0][r], M[3 * 6 + b][1][r]}, f32x2{M[4 * 6 + b][0][r], M[4 * 6 + b][1][r]}, f32x2{M[5 * 6 + b][0][r], M[5 * 6 + b][1][r]},
                               z[b][0], z[b][1], z[b][2], z[b][3]);
            f32x2 yy[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w4_at_t<f32x2>(z[0][i], z[1][i], z[2][i], z[3][i], z[4][i], z[5][i], yy[i][0], yy[i][1], yy[i][2], yy[i][3]);
                if (!POOL && !raw) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x2 x = yy[i][j] + bias;
                        if (!SPLITK) {                      // (slope 1 = a linear layer: max(x, x); no per-value select on p.act)
                            const f32x2 sx = slope * x;
                            x = f32x2{fmaxf(x[0], sx[0]), fmaxf(x[1], sx[1])};
                        }
                        yy[i][j] = x;
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int t = 16 * m + 4 * lq + r;          // MFMA row = Winograd tile
                const int off = tab[t];
                const int fl = tab[W4_TILES + t];
                const int vo = (cok && off >= 0) ? (off + cout_off) * 4 : OOR;
                auto store_tile = [&](auto full_tag) {
                    constexpr bool FULL = decltype(full_tag)::value;       // no edge selects: every store of the tile goes to `vo`
                    if (POOL && !raw) {
#pragma unroll
                        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                            for (int pj = 0; pj < 2; ++pj) {
                                // bias + leaky-ReLU AFTER the max: x -> fl(x + bias) and the leaky-ReLU are monotonic, so
                                // max_i act(fl(y_i + b)) == act(fl(max_i y_i + b)) bit for bit -- 4 instead of 16 per tile and cout
                                float v = fmaxf(fmaxf(yy[2 * pi][2 * pj][m], yy[2 * pi][2 * pj + 1][m]), fmaxf(yy[2 * pi + 1][2 * pj][m], yy[2 * pi + 1][2 * pj + 1][m])) + bias;
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
                            for (int j = 0; j < 4; ++j) HP3D_BUFFER_STORE4(orsrc, yy[i][j][m], (FULL || j < vc) ? vrow : OOR, i * srow + j * scol);
                        }
                    }
                };
                if (!POOL && full) store_tile(std::true_type{});        // (the pooled epilogue measured slower with the second path: 4 stores per tile)
                else store_tile(std::false_type{});
            }
        }
        if (n_item < 0) W4_T_FLUSH(lane);
        if (n_item < 0) break;
        item = n_item; cy = n_cy; tblock = n_tblock; wvoff = n_wvoff;
        if (VARSTEPS) { kz = n_kz; piece = n_piece; s0 = n_s0; s1 = n_s1; }
    }
}

// Tail pieces -> outputs (wino4_shared.h: w4_tail_reduce_body)
template <bool POOL>
HP3D_KERNEL(256)
void wino4_tail_reduce_kernel(const ConvParams p) { w4_tail_reduce_body<POOL, W4_TILES, W4_COUTS, W4_CK>(p, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x); }

}  // namespace

// U = G g G^T per (virtual cin, cout), G of F(4x4,3x3), evaluated in double and rounded once; packed in conv_wino2's fragment
// order with 36 planes: [plane a*6+b][step = vc / 16][Cout/16][q][n][e], virtual channel vc = 16 step + 4 q + e, cout 16 co16 + n
// (zero padded).  Virtual channels as in wino_pack_weights (k = 7: nine 3x3 blocks of the filter zero-extended to 9x9).
size_t wino4_packed_floats(int k, int cin_pad, int cout_pad) { return (size_t)W4_NP * (k == 7 ? 9 : 1) * cin_pad * cout_pad; }

void wino4_pack_weights(const float* g_hwio, int k, int Cin, int Cout, int cin_pad, int cout_pad, const int* chan_map, float* dst) {
    const double G[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int nsub = k == 7 ? 9 : 1;
    const int nst = nsub * cin_pad / 16, CO16 = cout_pad / 16;
    memset(dst, 0, sizeof(float) * wino4_packed_floats(k, cin_pad, cout_pad));
    // one (16-cout block, channel quad) at a time: its 36 x 64 transformed values are 36 contiguous 256-byte runs of the packed array
    // (the plane-innermost order of the first version scattered every store over 36 cache lines: 60 s for the full weight set)
    float blk[W4_NP][64];
    for (int sub = 0; sub < nsub; ++sub) {
        const int u0 = k == 7 ? 3 * (sub / 3) : 0, v0 = k == 7 ? 3 * (sub % 3) : 0;
        for (int eq = 0; eq < cin_pad; eq += 4) {
            int rc[4];
            bool any = false;
            for (int e = 0; e < 4; ++e) {
                rc[e] = chan_map ? chan_map[eq + e] : (eq + e < Cin ? eq + e : -1);
                any |= rc[e] >= 0;
            }
            if (!any) continue;
            const int vc = sub * cin_pad + eq;
            const int st = vc >> 4, q = (vc >> 2) & 3;
            for (int c16 = 0; c16 < (Cout + 15) / 16; ++c16) {
                memset(blk, 0, sizeof(blk));
                for (int n = 0; n < 16; ++n) {
                    const int co = 16 * c16 + n;
                    if (co >= Cout) break;
                    for (int e = 0; e < 4; ++e) {
                        if (rc[e] < 0) continue;
                        double w3[3][3];
                        for (int r = 0; r < 3; ++r)
                            for (int c = 0; c < 3; ++c)
                                w3[r][c] = (u0 + r < k && v0 + c < k) ? (double)g_hwio[((size_t)((u0 + r) * k + (v0 + c)) * Cin + rc[e]) * Cout + co] : 0.0;
                        for (int a = 0; a < 6; ++a) {
                            double ga[3];
                            for (int c = 0; c < 3; ++c) ga[c] = G[a][0] * w3[0][c] + G[a][1] * w3[1][c] + G[a][2] * w3[2][c];
                            for (int b = 0; b < 6; ++b) blk[a * 6 + b][n * 4 + e] = (float)(ga[0] * G[b][0] + ga[1] * G[b][1] + ga[2] * G[b][2]);
                        }
                    }
                }
                for (int pl = 0; pl < W4_NP; ++pl)
                    memcpy(dst + ((((size_t)pl * nst + st) * CO16 + c16) * 4 + q) * 64, blk[pl], sizeof(blk[pl]));
            }
        }
    }
}

// Returns 1 when the layer can run here; *ksplit (may be NULL) receives the channel split that fills the chip (one workgroup per CU).
int conv_wino4_eligible(int k, int stride, int Cin, int Cout, int Ho, int Wo, int B, int in_cs, int out_cs, int pool, int* ksplit) {
    if (ksplit) *ksplit = 1;
    if ((k != 3 && k != 7) || stride != 1 || Cin % W4_CK || Cout % W4_COUTS) return 0;
    // 32-bit offsets, and the column-outside-the-image constant needs the input below 2^30 bytes
    if ((long)B * Ho * Wo * in_cs * 4 >= (1L << 30) || (long)B * Ho * Wo * out_cs * 4 >= (1L << 31)) return 0;
    if (pool && (k != 3 || ((Ho | Wo) & 1))) return 0;
    const long tiles = (long)B * ((Ho + 3) / 4) * ((Wo + 3) / 4);
    const long items = (tiles + W4_TILES - 1) / W4_TILES * (Cout / W4_COUTS);
    const int slots = hp3d_num_cus();
    if (items >= slots || !ksplit) return 1;
    const int nsteps = (k == 7 ? 9 : 1) * Cin / W4_CK;
    int ks = (int)(slots / items);
    if (ks > nsteps / 2) ks = nsteps / 2;
    if (ks > 32) ks = 32;
    if (ks >= 2 && !pool && (long)ks * B * Ho * Wo * Cout * 4 < (1L << 31)) *ksplit = ks;
    return 1;
}

size_t conv_wino4_tail_floats() { return (size_t)2 * hp3d_num_cus() * W4_PIECE_FLOATS; }        // two pieces per workgroup

// Tail plan: items = full rounds of one workgroup per CU + a remainder.  The remainder's rem x nsteps item-steps are shared out in equal
// runs of q = ceil(rem x nsteps / CUs) steps (at least 2: a piece pays a whole epilogue), one run per workgroup.  Returns q (0 = no
// tail pieces: the last round is full, or so nearly full that the reduction would cost more than the idle CUs).
int conv_wino4_tail_plan(int Cin, int Cout, int Ho, int Wo, int B, int* tail_items) {
    if (tail_items) *tail_items = 0;
    const long tiles = (long)B * ((Ho + 3) / 4) * ((Wo + 3) / 4);
    const long items = (tiles + W4_TILES - 1) / W4_TILES * (Cout / W4_COUTS);
    const int slots = hp3d_num_cus(), nsteps = Cin / W4_CK;
    const int rem = (int)(items % slots);
    if (rem == 0 || rem * 8 > slots * 7 || nsteps < 2) return 0;
    int q = (int)(((long)rem * nsteps + slots - 1) / slots);
    if (q < 2) q = 2;
    if (q >= nsteps) return 0;                   // whole items per workgroup: nothing to cut
    if (tail_items) *tail_items = rem;
    return q;
}

template <bool POOL, int NSUB, bool SPLITK, bool TAIL>
static void wino4_launch_t(const ConvParams& p, long tiles, hipStream_t s) {
    static bool attr_done[64] = {};
    auto k = conv_wino4_kernel<POOL, NSUB, SPLITK, TAIL>;
    if (hp3d_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, W4_SMEM_BYTES);
    const long items = (tiles + W4_TILES - 1) / W4_TILES * (p.Cout / W4_COUTS) * (SPLITK ? p.ksplit : 1);
    const int slots = hp3d_num_cus();                     // persistent grid: one workgroup per CU
    dim3 grid((unsigned)((items < slots && !(TAIL && p.tail_items > 0)) ? items : slots));
    HP3D_LAUNCH(k, grid, dim3(256), W4_SMEM_BYTES, s, p);
}

// pin.ksplit > 1: pin.out must be the partial-sum scratch [ksplit][B*Ho*Wo][Cout] with out_cs = cout_store = Cout; the caller runs
// conv_splitk_reduce afterwards (bias + activation happen there).
// Returns < 0: refused; 0: launched; 1: launched AND the last round ran as tail pieces (the scratch alone does not say so: an unaligned
// output or a pooled layer with cout_store % 4 != 0 drops the tail here).
int conv_wino4_launch(const ConvParams& pin, int pool, hipStream_t s) {
    const long kso = pin.ksplit > 1 ? pin.ksplit : 1;
    if ((long)pin.B * pin.H * pin.W * pin.in_cs * 4 >= (1L << 30) || kso * pin.B * pin.Ho * pin.Wo * pin.out_cs * 4 >= (1L << 31)) return -1;
    if (pin.nsub != 1 && pin.nsub != 9) return -1;
    if (pin.Cout % W4_COUTS || pin.Cin % W4_CK) return -1;
    ConvParams p = pin;
    p.tiles_x = (p.Wo + 3) / 4;
    p.tiles_y = (p.Ho + 3) / 4;
    const long tiles = (long)p.B * p.tiles_x * p.tiles_y;
    if (pool && (p.nsub != 1 || ((p.Ho | p.Wo) & 1))) return -1;
    if (p.ksplit > 1) {
        const int nsteps = p.nsub * p.Cin / W4_CK;
        if (pool || p.ksplit * 2 > nsteps || p.out_cs != p.Cout) return -1;
        if (p.nsub == 9) wino4_launch_t<false, 9, true, false>(p, tiles, s); else wino4_launch_t<false, 1, true, false>(p, tiles, s);
        W4_T_REPORT(p, s, p.nsub == 9 ? "7x7 split" : "3x3 split");
        return 0;
    }
    p.ksplit = 1;
    p.tail_items = p.tail_q = 0;
    if (p.nsub == 9) {
        wino4_launch_t<false, 9, false, false>(p, tiles, s);
        W4_T_REPORT(p, s, "7x7");
        return 0;
    }
    // 3x3: the same instantiation serves launches with and without tail pieces (tail_items = 0: every item is a whole item)
    if (p.partial && p.partial_cap >= conv_wino4_tail_floats() && !(pool && (p.cout_store & 3)) && (p.out_cs & 3) == 0 && ((uintptr_t)p.out & 15) == 0)
        p.tail_q = conv_wino4_tail_plan(p.Cin, p.Cout, p.Ho, p.Wo, p.B, &p.tail_items);
    if (pool) wino4_launch_t<true, 1, false, true>(p, tiles, s);
    else wino4_launch_t<false, 1, false, true>(p, tiles, s);
    W4_T_REPORT(p, s, pool ? "3x3 pool" : "3x3");
    if (p.tail_items > 0) {
        const long total = (long)p.tail_items * W4_TILES * (pool ? 4 : 16) * (W4_COUTS / 4);
        const unsigned blocks = (unsigned)((total + 255) / 256);
        if (pool) HP3D_LAUNCH(wino4_tail_reduce_kernel<true>, dim3(blocks), dim3(256), 0, s, p);
        else HP3D_LAUNCH(wino4_tail_reduce_kernel<false>, dim3(blocks), dim3(256), 0, s, p);
        return 1;
    }
    return 0;
}

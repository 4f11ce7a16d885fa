// conv_wino7.hip -- PoseNet2D's 7x7 refinement layers as Winograd F(4x4, 4x4) on the f32 matrix cores (round 5).
//
// Call sites: the ten 7x7 / stride-1 layers of PoseNet2D's two refinement units (NetworkOps.conv_relu, utils/general.py:36-59;
// layer list nets/ColorHandPose3DNetwork.py:206-215), float32 mode, launches that fill the chip.  Until round 4 they ran on
// conv_wino4.hip as NINE 3x3 blocks of the filter zero-extended to 9x9, F(4x4,3x3) each: 289 plane products per 16 outputs after
// the structurally zero planes.  Here the filter is zero-extended to 8x8 = FOUR blocks of 4x4 taps, each a Winograd F(4x4,4x4)
// over the points {0, +-1, +-2, 1/2, inf}: a 7x7 input window, 49 products per 16 outputs, 169 for the four blocks after the
// structurally zero planes (the tap row / column 7 of the extension is zero: the "inf" row a = 6 of block row 1, column b = 6 of
// block column 1) -- 0.585 of the matrix-core work, at the SAME rounding error as the nine-block form (one layer on unit-variance
// data against float64: rms 2.2e-6 both, max 1.1e-5 vs 2.8e-5; end to end: profiles/r05_wino7_numerics.md).
//
//   Y(4x4) = A^T [ sum_{i,j in {0,1}} sum_cin (G g_ij G^T) .* (B^T d_ij B) ] A        g_ij: taps (4i..4i+3, 4j..4j+3) of the 8x8 extension
//
// What the 4-tap blocks buy besides fewer products: block (i, j) of tile (ty, tx) reads the input window that starts at pixel
// (4 (ty + i) - 3, 4 (tx + j) - 3) -- block shift = tile stride = 4 -- which IS the window of block (0, 0) of tile (ty + i, tx + j).
// So V = B^T d B is computed ONCE per tile for a 4x4 tile block plus one halo row and column of tiles (25 windows) and the four
// blocks only read their neighbours' V: 25 transforms per 64 (tile, block) uses.  (With 3-tap blocks on 4x4 tiles the shifts are
// multiples of 3: nothing lines up, every block transforms its own windows.)
//
// Machine shape (one workgroup of four waves per CU, `__launch_bounds__(256, 1)`):
//   * work item = one 4x4 TILE BLOCK (16 x 16 output pixels of one image) x 64 couts; at B = 32 on the 32 x 32 score maps that is
//     32 x 4 x 2 = exactly 256 items: one round on 256 CUs, NO channel split, no partial sums, no reduce launch (round 4: 128
//     items of 32 tiles x 64 couts, split in two + conv_splitk_reduce: 0.212 + 0.014 ms per layer);
//   * a wave owns all 16 tiles x 16 couts: one `v_mfma_f32_16x16x4_f32` tile per plane, 49 planes x 4 = 196 accumulators, all
//     PINNED to AGPRs (inline-asm "+a", as in conv_wino4.hip); consecutive MFMAs alternate between two planes (a dependent
//     pair would wait 40 cycles for a 32-cycle instruction);
//   * K loop in 16-channel CHUNKS: V of a chunk = 49 planes x 25 windows x 16 channels = 78.4 KB, double buffered (156.8 KB of
//     the 160 KB); per chunk a wave runs the 169 (block, plane) products = 676 MFMAs out of one V buffer, ONE barrier per chunk;
//     the skip of the structurally zero planes is a COMPILE-TIME list (the four blocks are unrolled), not a branch;
//   * loader thread = (window slot 0..31 of which 25 exist, channel pair): 49 window loads of 8 bytes spread one per MFMA group
//     over the first 49 groups, the packed 7-point transform (23 operations per 1-D pass) under group 60, 49 LDS writes behind
//     the groups after it; V rows are 16 channels = four 16-byte quads, quad q of window row hy stored at q ^ 2 (hy & 1): every
//     `ds_read_b128` lane group of every block then touches 16 different bank quads (searched exhaustively,
//     profiles/r05_tuning_notes.md);
//   * transformed filters U[chunk][169 (block, plane)][Cout/16][q][n][e] -- zero planes not stored -- stream global -> VGPR through a
//     ring of 13 fragments (169 = 13 x 13: static slots), one scalar offset per fragment; the weight stream per MFMA is four times
//     conv_wino4.hip's (a fragment serves 16 tiles for four MFMAs): 32 B/clk/CU from L2.
#include "hp3d_common.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace {

constexpr int W7_NP = 49;                          // planes of F(4x4,4x4)
constexpr int W7_TILES = 16;                       // a 4x4 block of Winograd tiles (4x4 outputs each) per item
constexpr int W7_HALO = 25;                        // 5x5 transformed windows serve the four tap blocks of the 16 tiles
constexpr int W7_CK = 16;                          // channels per chunk
constexpr int W7_COUTS = 64;                       // output channels per item (16 per wave)
constexpr int W7_SEQ = 169;                        // (block, plane) products per chunk: 49 + 42 + 42 + 36
constexpr int W7_PLANE_FLOATS = W7_HALO * W7_CK;   // 400 floats = 1600 B
constexpr int W7_VBUF_FLOATS = W7_NP * W7_PLANE_FLOATS;
constexpr int W7_SMEM_BYTES = 2 * W7_VBUF_FLOATS * 4 + 2 * 2 * W7_TILES * 4;      // 156800 + 256 B
constexpr int W7_RING = 13;                        // weight fragments in flight; must divide W7_SEQ (static slots)
static_assert(W7_SEQ % W7_RING == 0, "static ring slots");
constexpr int W7_HALF = 25;                        // planes reachable from one LDS base (16-bit immediate offsets: 24 x 1600 B)
constexpr int W7_GROUPS = (W7_SEQ + 1) / 2;        // MFMA groups per chunk: 84 pairs of products + one single
constexpr int W7_TRANSFORM_AT = 60;                // the group under which the next chunk's windows are transformed
constexpr int W7_WRITES_PER_GROUP = 3;             // V writes behind each group after it (49 over 17 groups)

// block b = 2 i + j covers taps (4 i .. 4 i + 3, 4 j .. 4 j + 3) of the 8x8 extension; row / column 7 is zero, so G g G^T has a zero
// row a = 6 for i = 1 and a zero column b = 6 for j = 1 (G's "inf" row picks the block's last tap)
constexpr bool w7_zero_plane(int blk, int pl) { return ((blk >> 1) && pl / 7 == 6) || ((blk & 1) && pl % 7 == 6); }
struct W7Seq { int blk[W7_SEQ], pl[W7_SEQ]; };
constexpr W7Seq w7_make_seq() {
    W7Seq s{};
    int n = 0;
    for (int b = 0; b < 4; ++b)
        for (int pl = 0; pl < W7_NP; ++pl)
            if (!w7_zero_plane(b, pl)) { s.blk[n] = b; s.pl[n] = pl; ++n; }
    return s;
}
constexpr W7Seq W7_SEQ_TAB = w7_make_seq();
// ISSUE ORDER of a chunk's 49 window loads.  Window element (r, c) of slot (hy, hx) is the pixel (4 hy + r - 3, 4 hx + c - 3): elements (r, c),
// (r + 4, c), (r, c + 4), (r + 4, c + 4) of neighbouring slots are the SAME pixels, i.e. the same cache lines asked for by other lanes.  Issued
// class by class ((r mod 4, c mod 4): 16 classes of 4 / 2 / 1 elements) the repeats follow each other within a group or two and hit the L1
// line or its pending fill; in row-major order they lie four groups (1000 cycles, 40 KB of weight stream through the 32 KB L1) apart
// (conv_wino4.hip's finding of round 4, same reason).
struct W7Issue { int e[W7_NP]; };
constexpr W7Issue w7_make_issue() {
    W7Issue s{};
    int n = 0;
    for (int rc = 0; rc < 4; ++rc)
        for (int cc = 0; cc < 4; ++cc)
            for (int r = rc; r < 7; r += 4)
                for (int c = cc; c < 7; c += 4) s.e[n++] = r * 7 + c;
    return s;
}
constexpr W7Issue W7_ISSUE = w7_make_issue();
static_assert(W7_SEQ_TAB.pl[W7_SEQ - 1] == 40 && W7_SEQ_TAB.blk[W7_SEQ - 1] == 3, "169 products: the last one is block (1,1), plane (5,5)");

// B^T of F(4,4) over {0, 1, -1, 2, -2, 1/2, inf}, applied to seven values in place (rows printed by scripts/micro/wino_f44.py):
//   [-2 4 5/2 -5 -1/2 1 0; 0 2 -2 -9/2 1/2 1 0; 0 -2 6 -7/2 -3/2 1 0; 0 1 -3/2 -2 3/2 1 0; 0 -1 5/2 0 -5/2 1 0; 0 4 0 -5 0 1 0;
//    0 -2 4 5/2 -5 -1/2 1]            (23 operations: rows 0 and 6 reuse row 5's 4 x1 - 5 x3 + x5 pattern)
template <typename T>
__device__ __forceinline__ void w7_bt(T& x0, T& x1, T& x2, T& x3, T& x4, T& x5, T& x6) {
    const T t5 = (4.f * x1 + x5) - 5.f * x3;
    const T v6 = (4.f * x2 + x6) - 5.f * x4;
    const T t0 = t5 + ((2.5f * x2 - 0.5f * x4) - 2.f * x0);
    const T t6 = v6 - 0.5f * t5;
    const T t1 = ((2.f * x1 + x5) - 2.f * x2) + (0.5f * x4 - 4.5f * x3);
    const T t2 = ((6.f * x2 + x5) - 2.f * x1) - (3.5f * x3 + 1.5f * x4);
    const T t3 = ((x1 + x5) - 1.5f * x2) + (1.5f * x4 - 2.f * x3);
    const T t4 = (x5 - x1) + 2.5f * (x2 - x4);
    x0 = t0; x1 = t1; x2 = t2; x3 = t3; x4 = t4; x5 = t5; x6 = t6;
}
// A^T of F(4,4): [1 1 1 1 1 1 0; 0 1 -1 2 -2 1/2 0; 0 1 1 4 4 1/4 0; 0 1 -1 8 -8 1/8 1]
template <typename T>
__device__ __forceinline__ void w7_at(T m0, T m1, T m2, T m3, T m4, T m5, T m6, T& y0, T& y1, T& y2, T& y3) {
    const T s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    y0 = ((m0 + s12) + s34) + m5;
    y1 = (d12 + 2.f * d34) + 0.5f * m5;
    y2 = (s12 + 4.f * s34) + 0.25f * m5;
    y3 = ((d12 + 8.f * d34) + 0.125f * m5) + m6;
}

// quad swizzle of a V row: window slot (hy, hx) stores channel quad q at position q ^ 2 (hy & 1)
__device__ __forceinline__ int w7_swz_row(int hy) { return (hy & 1) << 1; }

// SPLITK (under-filled launches, i.e. small batches): an item additionally owns a contiguous range of the 16-channel chunks and stores its RAW 4x4 sums
// (the output transform is linear) into `[ksplit][B*Ho*Wo][Cout]`; conv_splitk_reduce adds the slices in order, + bias, activation (deterministic).
template <bool SPLITK>
HP3D_KERNEL2(256, 1)
void conv_wino7_kernel(const ConvParams p) {
    HP3D_DYN_SMEM(V);
    int* tinfo = (int*)(V + 2 * W7_VBUF_FLOATS);       // [parity][0..15] output offset of tile t (-1: none), [16..31] valid rows | valid columns << 4
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = HP3D_READFIRSTLANE(tid >> 6);
    const int ln = lane & 15, lq = lane >> 4;          // MFMA column (cout) / row group; k slot

    // ---- geometry: 4x4 tile blocks on each image's own grid (blocks at the right / bottom edge may hold tiles outside the image) ----
    const int TXn = p.tiles_x, TYn = p.tiles_y;
    const int bxn = (TXn + 3) >> 2, byn = (TYn + 3) >> 2, per_img = bxn * byn;
    const int nblocks = p.B * per_img, ncy = p.Cout / W7_COUTS;
    const int per_split = nblocks * ncy;
    const int nitems = per_split * (SPLITK ? p.ksplit : 1);
    // item -> (cout block, tile block): XCD-affine (workgroup ids go round-robin over the 8 XCDs; the cout blocks of one tile block run on ONE XCD,
    // so its windows cross the fabric once), as conv_wino4.hip's item_decode
    auto item_decode = [&](int r, int& cy_, int& tb_) {
        const int aff = (nblocks >> 3) * 8 * ncy;
        if (r < aff) {
            const int xcd = r & 7, j = r >> 3, tbq = j / ncy;
            cy_ = j - tbq * ncy;
            tb_ = tbq * 8 + xcd;
        } else {
            const int q = r - aff, tbi = q / ncy;
            cy_ = q - tbi * ncy;
            tb_ = (nblocks & ~7) + tbi;
        }
    };
    auto block_decode = [&](int tb, int& img, int& by, int& bx) {
        img = tb / per_img;
        const int r = tb - img * per_img;
        by = r / bxn;
        bx = r - by * bxn;
    };
    auto table_write = [&](int tb, int parity, int kz) {
        if (tid < W7_TILES) {
            int img, by, bx;
            block_decode(tb, img, by, bx);
            const int ty = 4 * by + (tid >> 2), tx = 4 * bx + (tid & 3);
            int off = -1, fl = 0;
            if (ty < TYn && tx < TXn) {
                off = (((SPLITK ? kz * p.B + img : img) * p.Ho + 4 * ty) * p.Wo + 4 * tx) * p.out_cs;
                fl = min(4, p.Ho - 4 * ty) | (min(4, p.Wo - 4 * tx) << 4);
            }
            tinfo[parity * 2 * W7_TILES + tid] = off;
            tinfo[parity * 2 * W7_TILES + W7_TILES + tid] = fl;
        }
    };

    // ---- loader role: this thread transforms the 7x7 window of slot (hy, hx) for channel pair cp ---------------------------------------
    // (threads 200..255 have no window of their own: they MIRROR slots 17..23 -- the same loads, the same arithmetic, the same LDS writes of
    //  the same values as the owning thread -- so that no per-lane predicate, i.e. no exec-mask branch, sits in the chunk body)
    const int ht = (tid >> 3) < W7_HALO ? (tid >> 3) : (tid >> 3) - 8, cp = tid & 7;
    const int hy = ht / 5, hx = ht - hy * 5;
    const int cs4 = p.in_cs * 4;
    constexpr int OOR = (int)0x80000000;          // row outside the image / no such window
    constexpr int COL_OOR = 0x60000000;           // column outside the image: any row term + this is >= 2^30 > the buffer's extent
    int ro[7], co[7];
    auto loader_setup = [&](int tb) {
        int img, by, bx;
        block_decode(tb, img, by, bx);
        const int wy0 = 4 * (4 * by + hy) - p.pad_t, wx0 = 4 * (4 * bx + hx) - p.pad_l;
        const int wbase = ((img * p.H + wy0) * p.W + wx0) * cs4 + cp * 8;
#pragma unroll
        for (int r = 0; r < 7; ++r) ro[r] = (unsigned)(wy0 + r) < (unsigned)p.H ? wbase + r * (p.W * cs4) : OOR;
#pragma unroll
        for (int c = 0; c < 7; ++c) co[c] = (unsigned)(wx0 + c) < (unsigned)p.W ? c * cs4 : COL_OOR;
    };
    const hp3d_rsrc_t irsrc = HP3D_MAKE_RSRC(p.in, (unsigned)p.B * (unsigned)(p.H * p.W) * (unsigned)cs4);
    [[maybe_unused]] const unsigned out_bytes = (unsigned)(SPLITK ? p.ksplit * p.B : p.B) * (unsigned)(p.Ho * p.Wo) * (unsigned)p.out_cs * 4u;

    f32x2 d[W7_NP];
    auto window_load = [&](int e, int soff) { d[e] = HP3D_BUFFER_LOAD8(irsrc, (int)((unsigned)ro[e / 7] + (unsigned)co[e % 7]), soff); };
    // B^T d B in place: along the window rows first (plane row a), then along the columns (plane column b): d[a * 7 + b] = plane a * 7 + b
    auto transform_arith = [&]() {
#pragma unroll
        for (int c = 0; c < 7; ++c) w7_bt(d[0 * 7 + c], d[1 * 7 + c], d[2 * 7 + c], d[3 * 7 + c], d[4 * 7 + c], d[5 * 7 + c], d[6 * 7 + c]);
#pragma unroll
        for (int a = 0; a < 7; ++a) w7_bt(d[a * 7 + 0], d[a * 7 + 1], d[a * 7 + 2], d[a * 7 + 3], d[a * 7 + 4], d[a * 7 + 5], d[a * 7 + 6]);
    };
    float* const Vw = V + ht * W7_CK + (((cp >> 1) ^ w7_swz_row(hy)) * 4) + (cp & 1) * 2;      // this thread's slot in plane 0 of buffer 0
    auto v_write = [&](int buf, int pl) {
        float* q0 = Vw + buf * W7_VBUF_FLOATS;
        float* dst = pl < W7_HALF ? q0 + pl * W7_PLANE_FLOATS : q0 + W7_HALF * W7_PLANE_FLOATS + (pl - W7_HALF) * W7_PLANE_FLOATS;
        *(f32x2*)dst = d[pl];
    };

    // ---- MFMA role -----------------------------------------------------------------------------------------------------------------------
    // packed U: [chunk][169 products][Cout/16][q 4][n 16][e 4]; the fragment of one (chunk, product) for a wave is 1 KB, lane-linear
    const int CO16 = p.Cout >> 4;
    const int nchunks = p.Cin / W7_CK;
    const hp3d_rsrc_t wrsrc = HP3D_MAKE_RSRC(p.wpk, ((unsigned)(W7_SEQ * p.Cin) * (unsigned)p.Cout + (unsigned)(W7_RING * 256) * (unsigned)(p.Cout >> 4)) * 4u);      // (incl. the ring's slack: wino7_packed_floats)
    const int entry_stride_b = CO16 * 1024;
    const int wv_lane = lane * 16;

    f32x4 M[W7_NP];          // [plane]: rows = tiles 4 (lane >> 4) + r, column = cout (lane & 15)
    f32x4 bq[W7_RING];
    f32x4 af[4];
    // A fragment of product (block b = 2 i + j, plane): tile ln = (y, x) of the block reads window slot (y + i, x + j), channel quad lq
    int ab[4][2];            // [block][plane half] LDS byte address of this lane's fragment in plane 0 / 25 of the current buffer
    auto a_bases = [&](int cur) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int sy = (ln >> 2) + (b >> 1), sx = (ln & 3) + (b & 1);
            ab[b][0] = cur * (W7_VBUF_FLOATS * 4) + ((sy * 5 + sx) * W7_CK + ((lq ^ w7_swz_row(sy)) * 4)) * 4;
            ab[b][1] = ab[b][0] + W7_HALF * W7_PLANE_FLOATS * 4;
            HP3D_OPAQUE_V(ab[b][0]);
            HP3D_OPAQUE_V(ab[b][1]);
        }
    };
    auto a_fetch = [&](int t) {
        const int b = W7_SEQ_TAB.blk[t], pl = W7_SEQ_TAB.pl[t];
        const int base = ab[b][pl < W7_HALF ? 0 : 1], po = pl < W7_HALF ? pl : pl - W7_HALF;
        af[t & 3] = *(const f32x4*)((const char*)V + base + po * (W7_PLANE_FLOATS * 4));
    };
    // the fragments of an item are ONE ascending stream (chunk-major, products in kernel order, a chunk's 169 blocks contiguous): a running
    // scalar offset, bumped after every fetch.  (Written as 169 offsets from a per-chunk base the compiler hoisted all of them out of the chunk
    // loop: 434 spilled SGPRs, a v_readlane in front of every weight load.)
    int wsb = 0;
    auto b_fetch = [&](int t) {
        bq[t % W7_RING] = HP3D_BUFFER_LOAD16(wrsrc, wv_lane, wsb);
        HP3D_SADD(wsb, entry_stride_b);
    };

    for (int item = blockIdx.x, k = 0; item < nitems; item += (int)gridDim.x, ++k) {
        int cy, tblock;
        const int kz = SPLITK ? HP3D_READFIRSTLANE(item / per_split) : 0;
        item_decode(SPLITK ? item - kz * per_split : item, cy, tblock);
        cy = HP3D_READFIRSTLANE(cy);
        tblock = HP3D_READFIRSTLANE(tblock);
        const int c0 = SPLITK ? (kz * nchunks) / p.ksplit : 0, c1 = SPLITK ? ((kz + 1) * nchunks) / p.ksplit : nchunks;      // this item's chunks
        const int cyoff = (cy * (W7_COUTS / 16) + wave) * 1024;          // byte offset of this wave's cout group inside a product's block
        if (k) __syncthreads();                   // the previous item's epilogue has read its tile table; its last chunk's V buffer is free
        loader_setup(tblock);
        table_write(tblock, 0, kz);
        // ---- prologue: the first chunk's windows -> V[0], the first ring of weight fragments, accumulators = 0
#pragma unroll
        for (int e = 0; e < W7_NP; ++e) window_load(e, c0 * (W7_CK * 4));
        wsb = cyoff + c0 * (W7_SEQ * entry_stride_b);
#pragma unroll
        for (int t = 0; t < W7_RING; ++t) b_fetch(t);
#pragma unroll
        for (int pl = 0; pl < W7_NP; ++pl) M[pl] = f32x4{0.f, 0.f, 0.f, 0.f};
        transform_arith();
#pragma unroll
        for (int pl = 0; pl < W7_NP; ++pl) v_write(0, pl);
        __syncthreads();
        int cur = 0;

        for (int chunk = c0; chunk < c1; ++chunk) {
            const bool lastc = chunk + 1 == c1;
            const int wsoff = (lastc ? chunk : chunk + 1) * (W7_CK * 4);      // (the last chunk re-reads its own windows: harmless, never used)
            // (weight fragments requested past this item's last chunk are the next chunk's, or lie outside the buffer and read as 0: never used)
            a_bases(cur);
            a_fetch(0);
            a_fetch(1);
#pragma unroll
            for (int g = 0; g < W7_GROUPS; ++g) {
                const int t0 = 2 * g, t1 = 2 * g + 1;
                if (t1 < W7_SEQ) {
                    // two products on two accumulators, alternating (no MFMA waits for its predecessor), ONE memory instruction behind each pair:
                    // next group's A fragments | the weight fragments for the two ring slots this group releases
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        HP3D_SCHED_BARRIER();
                        HP3D_MFMA16_X2(M[W7_SEQ_TAB.pl[t0]], M[W7_SEQ_TAB.pl[t1]], af[t0 & 3][e], af[t1 & 3][e], bq[t0 % W7_RING][e], bq[t1 % W7_RING][e]);
                        HP3D_SCHED_BARRIER();
                        if (e == 0) { if (t0 + 2 < W7_SEQ) a_fetch(t0 + 2); }
                        else if (e == 1) { if (t1 + 2 < W7_SEQ) a_fetch(t1 + 2); }
                        else if (e == 2) {
                            if (g < W7_NP) window_load(W7_ISSUE.e[g], wsoff);            // next chunk's windows: one per group over the first 49, class by class
                        } else {
                            b_fetch(t0);          // product t0 + 13 of the stream (this chunk's, or the next chunk's first ones) into the slot t0 released
                            b_fetch(t1);
                        }
                    }
                } else {
                    HP3D_SCHED_BARRIER();
                    HP3D_MFMA16_X1(M[W7_SEQ_TAB.pl[t0]], af[t0 & 3], bq[t0 % W7_RING]);
                    HP3D_SCHED_BARRIER();
                    b_fetch(t0);
                }
                if (g == W7_TRANSFORM_AT) {
                    if (!lastc) transform_arith();
                } else if (g > W7_TRANSFORM_AT) {
                    // V of the next chunk: a few LDS writes behind each remaining group (one burst would park all four waves on the LDS port)
                    if (!lastc) {
#pragma unroll
                        for (int j = 0; j < W7_WRITES_PER_GROUP; ++j) {
                            const int pl = (g - W7_TRANSFORM_AT - 1) * W7_WRITES_PER_GROUP + j;
                            if (pl < W7_NP) v_write(cur ^ 1, pl);
                        }
                    }
                }
            }
            static_assert((W7_GROUPS - 1 - W7_TRANSFORM_AT) * W7_WRITES_PER_GROUP >= W7_NP, "every plane of V is written before the barrier");
            HP3D_SCHED_BARRIER();
            __syncthreads();             // V[cur ^ 1] complete, V[cur] free
            cur ^= 1;
        }

        // ---- epilogue: Y = A^T M A per (tile, cout), bias + leaky-ReLU, NHWC store ------------------------------------------------------
#ifndef HP3D_EMU
        // the accumulators were last written by MFMAs inside inline-asm statements, which the compiler's hazard recogniser cannot see into
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3");
#endif
        const int cout = cy * W7_COUTS + wave * 16 + ln;
        const float bias = SPLITK ? 0.f : p.bias[cout];                  // (raw sums: bias and activation happen in the reduce)
        const bool cok = cout < p.cout_store;
        const float slope = (!SPLITK && p.act) ? HP3D_LEAKY_SLOPE : 1.f;
        const hp3d_rsrc_t orsrc = HP3D_MAKE_RSRC(p.out, out_bytes);
        const int srow = p.Wo * p.out_cs * 4, scol = p.out_cs * 4;
        const bool full = HP3D_OPAQUE_SGPR((((p.Ho | p.Wo) & 3) == 0) ? 1 : 0) != 0;
        // two accumulator registers (tiles 4 lq + 2 rp, + 1 of the same cout) go through A^T . A as ONE packed value: an instruction costs the
        // f32 matrix pipe the same cycles whether it carries one float or two
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            f32x2 z[7][4];                               // A^T M: along the plane rows a
#pragma unroll
            for (int b = 0; b < 7; ++b)
                w7_at<f32x2>(f32x2{M[0 * 7 + b][2 * rp], M[0 * 7 + b][2 * rp + 1]}, f32x2{M[1 * 7 + b][2 * rp], M[1 * 7 + b][2 * rp + 1]},
                             f32x2{M[2 * 7 + b][2 * rp], M[2 * 7 + b][2 * rp + 1]}, f32x2{M[3 * 7 + b][2 * rp], M[3 * 7 + b][2 * rp + 1]},
                             f32x2{M[4 * 7 + b][2 * rp], M[4 * 7 + b][2 * rp + 1]}, f32x2{M[5 * 7 + b][2 * rp], M[5 * 7 + b][2 * rp + 1]},
                             f32x2{M[6 * 7 + b][2 * rp], M[6 * 7 + b][2 * rp + 1]}, z[b][0], z[b][1], z[b][2], z[b][3]);
            f32x2 yy[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w7_at<f32x2>(z[0][i], z[1][i], z[2][i], z[3][i], z[4][i], z[5][i], z[6][i], yy[i][0], yy[i][1], yy[i][2], yy[i][3]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x2 x = yy[i][j] + bias;
                    const f32x2 sx = slope * x;          // (slope 1 = a linear layer)
                    yy[i][j] = f32x2{fmaxf(x[0], sx[0]), fmaxf(x[1], sx[1])};
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = 4 * lq + 2 * rp + h;       // MFMA row = tile of the block
                const int off = tinfo[t];
                const int fl = tinfo[W7_TILES + t];
                const int vo = (cok && off >= 0) ? (off + cout) * 4 : OOR;
                auto store_tile = [&](auto full_tag) {
                    constexpr bool FULL = decltype(full_tag)::value;       // no edge selects: every store of the tile goes to `vo`
                    const int vr = fl & 15, vc = fl >> 4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int vrow = (FULL || i < vr) ? vo : OOR;
#pragma unroll
                        for (int j = 0; j < 4; ++j) HP3D_BUFFER_STORE4(orsrc, yy[i][j][h], (FULL || j < vc) ? vrow : OOR, i * srow + j * scol);
                    }
                };
                if (full) store_tile(std::true_type{});
                else store_tile(std::false_type{});
            }
        }
    }
}

}  // namespace

// U = G g G^T per (block, channel, cout), G of F(4,4) over {0, 1, -1, 2, -2, 1/2, inf}, evaluated in double and rounded once; packed
// [chunk = c / 16][product t (169: the non-zero (block, plane) pairs in kernel order)][Cout/16][q][n][e], channel c = 16 chunk + 4 q + e,
// cout = 16 co16 + n (zero padded).  chan_map as in wino_pack_weights (the concat-buffer permutation of conv6_1 / conv7_1).
// (+ W7_RING entries of zeros: the weight ring keeps fetching W7_RING fragments past an item's last chunk through the SCALAR offset, which the
//  raw-buffer range check does not cover -- for the layer's last chunk those reads must still land in memory this layer owns)
size_t wino7_packed_floats(int cin_pad, int cout_pad) { return (size_t)W7_SEQ * cin_pad * cout_pad + (size_t)W7_RING * (cout_pad / 16) * 256; }

void wino7_pack_weights(const float* g_hwio, int Cin, int Cout, int cin_pad, int cout_pad, const int* chan_map, float* dst) {
    const double G[7][4] = {{-1.0 / 2, 0, 0, 0},
                            {-1.0 / 3, -1.0 / 3, -1.0 / 3, -1.0 / 3},
                            {1.0 / 9, -1.0 / 9, 1.0 / 9, -1.0 / 9},
                            {1.0 / 36, 1.0 / 18, 1.0 / 9, 2.0 / 9},
                            {-1.0 / 60, 1.0 / 30, -1.0 / 15, 2.0 / 15},
                            {32.0 / 45, 16.0 / 45, 8.0 / 45, 4.0 / 45},
                            {0, 0, 0, 1}};
    const int CO16 = cout_pad / 16;
    memset(dst, 0, sizeof(float) * wino7_packed_floats(cin_pad, cout_pad));
    for (int c = 0; c < cin_pad; ++c) {
        const int rc = chan_map ? chan_map[c] : (c < Cin ? c : -1);
        if (rc < 0) continue;
        const int chunk = c >> 4, q = (c >> 2) & 3, e = c & 3;
        for (int co = 0; co < Cout; ++co) {
            for (int t = 0; t < W7_SEQ; ++t) {
                const int blk = W7_SEQ_TAB.blk[t], pl = W7_SEQ_TAB.pl[t];
                const int a = pl / 7, b = pl % 7, u0 = 4 * (blk >> 1), v0 = 4 * (blk & 1);
                double s = 0.0;
                for (int r = 0; r < 4; ++r) {
                    if (u0 + r >= 7 || G[a][r] == 0.0) continue;
                    double row = 0.0;
                    for (int cc = 0; cc < 4; ++cc)
                        if (v0 + cc < 7) row += (double)g_hwio[((size_t)((u0 + r) * 7 + (v0 + cc)) * Cin + rc) * Cout + co] * G[b][cc];
                    s += G[a][r] * row;
                }
                dst[((((size_t)chunk * W7_SEQ + t) * CO16 + (co >> 4)) * 4 + q) * 64 + (co & 15) * 4 + e] = (float)s;
            }
        }
    }
}

// 1: the layer can run here -- 7x7 / stride 1, Cin % 16 == 0, Cout % 64 == 0, offsets below 2^30 / 2^31 bytes; *items (may be NULL) = work items of the launch
// *ksplit (may be NULL): the channel split that fills the chip when the launch alone does not (1 = none; at most one split per 16-channel chunk)
int conv_wino7_eligible(int k, int stride, int Cin, int Cout, int Ho, int Wo, int B, int in_cs, int out_cs, long* items, int* ksplit) {
    if (items) *items = 0;
    if (ksplit) *ksplit = 1;
    if (k != 7 || stride != 1 || Cin % W7_CK || Cout % W7_COUTS) return 0;
    if ((long)B * Ho * Wo * in_cs * 4 >= (1L << 30) || (long)B * Ho * Wo * out_cs * 4 >= (1L << 31)) return 0;
    const long blocks = (long)B * (((Ho + 3) / 4 + 3) / 4) * (((Wo + 3) / 4 + 3) / 4);
    const long n = blocks * (Cout / W7_COUTS);
    if (items) *items = n;
    const int slots = hp3d_num_cus(), nchunks = Cin / W7_CK;
    if (ksplit && n * 8 < (long)slots * 5) {
        int ks = (int)(slots / n);
        if (ks > nchunks) ks = nchunks;
        if (ks >= 2 && (long)ks * B * Ho * Wo * Cout * 4 < (1L << 31)) *ksplit = ks;
    }
    return 1;
}

template <bool SPLITK>
static void wino7_launch_t(const ConvParams& p, hipStream_t s) {
    static bool attr_done[64] = {};
    auto k = conv_wino7_kernel<SPLITK>;
    if (hp3d_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, W7_SMEM_BYTES);
    const long items = (long)p.B * ((p.tiles_x + 3) / 4) * ((p.tiles_y + 3) / 4) * (p.Cout / W7_COUTS) * (SPLITK ? p.ksplit : 1);
    const int slots = hp3d_num_cus();                     // persistent grid: one workgroup per CU
    dim3 grid((unsigned)(items < slots ? items : slots));
    HP3D_LAUNCH(k, grid, dim3(256), W7_SMEM_BYTES, s, p);
}

// pin.ksplit > 1: pin.out must be the partial-sum scratch [ksplit][B*Ho*Wo][Cout] with out_cs = cout_store = Cout; the caller runs
// conv_splitk_reduce afterwards (bias + activation happen there).
int conv_wino7_launch(const ConvParams& pin, hipStream_t s) {
    const long kso = pin.ksplit > 1 ? pin.ksplit : 1;
    if ((long)pin.B * pin.H * pin.W * pin.in_cs * 4 >= (1L << 30) || kso * pin.B * pin.Ho * pin.Wo * pin.out_cs * 4 >= (1L << 31)) return -1;
    if (pin.Cout % W7_COUTS || pin.Cin % W7_CK) return -1;
    ConvParams p = pin;
    p.tiles_x = (p.Wo + 3) / 4;
    p.tiles_y = (p.Ho + 3) / 4;
    if (p.ksplit > 1) {
        if (p.ksplit > p.Cin / W7_CK || p.out_cs != p.Cout) return -1;
        wino7_launch_t<true>(p, s);
    } else {
        p.ksplit = 1;
        wino7_launch_t<false>(p, s);
    }
    return 0;
}

// wino4_shared.h -- the item-shape independent parts of conv_wino4.hip (items of 32 tiles x 64 couts, 16-channel steps): the F(4x4,3x3)
// transforms, the order in which a step's 36 window loads are issued, the tile / item geometry and the reduction of tail pieces, the latter two
// parameterised by the item shape.  (Split out in round 4 for a second item shape, conv_wino4w.hip: 16 tiles x 128 couts in 32-channel steps.  That
// kernel passed the whole GPU suite in round 5 and was then REMOVED: -2 % at the bench shape on that visit against +0.8 % in round 4 -- no
// reliable win for 480 lines; profiles/r05_tuning_notes.md.)  Device code only; everything sits in an anonymous namespace.
#pragma once
#include "hp3d_common.h"

namespace {

constexpr int W4_NP = 36;                          // planes of F(4x4,3x3)

// ISSUE ORDER of the 36 window loads of a step.  Window element (r, c) of every tile is the pixel (4 ty + r - 1, 4 tx + c - 1): the elements
// (r, c), (r + 4, c), (r, c + 4), (r + 4, c + 4) of neighbouring tiles are the SAME pixels (the 6x6 windows overlap by two), i.e. the same
// cache lines asked for by other lanes.  Issued in row-major order those requests lie up to 12 planes (3000 cycles) apart and the 32 KB L1,
// through which ~300 KB stream per step, has dropped the line in between: every window line is filled up to four times.  Issued CLASS BY
// CLASS ((r mod 4, c mod 4): 16 classes of 4 / 2 / 1 elements) the repeats follow within a plane or two and hit the line (or its pending
// fill).  (Row-major, round 3's order: 2266 against 2326 images/s.)
#define W4_ISSUE_ELEM(k) ((int[36]){0, 4, 24, 28, 1, 5, 25, 29, 2, 26, 3, 27, 6, 10, 30, 34, 7, 11, 31, 35, 8, 32, 9, 33, 12, 16, 13, 17, 14, 15, 18, 22, 19, 23, 20, 21}[(k)])

// B^T of F(4x4,3x3) applied to six values in place:
//   [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
template <typename T>
__device__ __forceinline__ void w4_bt_t(T& x0, T& x1, T& x2, T& x3, T& x4, T& x5) {
    // twelve operations (rows 1 / 2 = a +- b with a = x4 - 4 x2, b = x3 - 4 x1; rows 3 / 4 = c +- 2 f with c = x4 - x2, f = x3 - x1); until round 6 the sums and
    // differences of the pairs were formed first: fourteen
    const T t0 = (4.f * x0 + x4) - 5.f * x2;
    const T t5 = (4.f * x1 + x5) - 5.f * x3;
    const T a = x4 - 4.f * x2, b = x3 - 4.f * x1, c = x4 - x2, f = x3 - x1;
    x0 = t0;
    x1 = a + b;
    x2 = a - b;
    x3 = c + 2.f * f;
    x4 = c - 2.f * f;
    x5 = t5;
}
// (packed FMAs: scalar float transforms were re-measured on the final kernel in round 4, 2.3 % slower)
__device__ __forceinline__ void w4_bt(f32x2& x0, f32x2& x1, f32x2& x2, f32x2& x3, f32x2& x4, f32x2& x5) {
    w4_bt_t<f32x2>(x0, x1, x2, x3, x4, x5);
}
// A^T of F(4x4,3x3): [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]   (T = float, or f32x2: two accumulators per instruction)
template <typename T>
__device__ __forceinline__ void w4_at_t(T m0, T m1, T m2, T m3, T m4, T m5, T& y0, T& y1, T& y2, T& y3) {
    const T s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    y0 = (m0 + s12) + s34;
    y1 = d12 + 2.f * d34;
    y2 = s12 + 4.f * s34;
    y3 = (d12 + 8.f * d34) + m5;
}
__device__ __forceinline__ void w4_at(float m0, float m1, float m2, float m3, float m4, float m5, float& y0, float& y1, float& y2, float& y3) {
    w4_at_t<float>(m0, m1, m2, m3, m4, m5, y0, y1, y2, y3);
}

// tile / item geometry shared by a convolution kernel and its tail reduction; TILES x COUTS = the item shape
template <int TILES, int COUTS>
struct W4GeomT {
    int TXn, TYn, per_img, tile_blocks, ncy;
    __device__ __forceinline__ W4GeomT(const ConvParams& p)
        : TXn(p.tiles_x), TYn(p.tiles_y), per_img(p.tiles_x * p.tiles_y), tile_blocks((p.B * p.tiles_x * p.tiles_y + TILES - 1) / TILES),
          ncy(p.Cout / COUTS) {}
    // flat tile id -> (image, tile row, tile column): bands of four tile rows, column-major inside a band (a block of 32 consecutive
    // ids is a 4 x 8 tile patch where the grid allows, 16 ids a 4 x 4 patch), continuing into the next image
    __device__ __forceinline__ void tile_decode(int id, int& tb, int& tyy, int& txx) const {
        tb = id / per_img;
        const int r = id - tb * per_img;
        const int band = r / (4 * TXn), rem = r - band * 4 * TXn;
        const int rows = min(4, TYn - 4 * band);
        txx = rem / rows;
        tyy = band * 4 + rem - txx * rows;
    }
    // item index (within one channel split) -> (cout block, tile block)
    __device__ __forceinline__ void item_decode(int r, int& cy_, int& tb_) const {
        // XCD-affine order: workgroup ids go round-robin over the 8 XCDs; within an XCD consecutive items are the cout blocks of ONE
        // tile block, so its windows are fetched from the fabric once per XCD and re-read from that XCD's L2
        // (round 4: any tile-block count -- the whole groups of eight tile blocks in that order, the up to seven left over behind them, still
        //  cout block innermost.  Before, a count that is not a multiple of 8 fell back to the cout-block-major order: B = 24 at 320x320,
        //  300 tile blocks, ran 6 % slower on one stream than as two halves.)
        const int aff = (tile_blocks >> 3) * 8 * ncy;
        if (r < aff) {
            const int xcd = r & 7, j = r >> 3, tbq = j / ncy;
            cy_ = j - tbq * ncy;
            tb_ = tbq * 8 + xcd;
        } else {
            const int q = r - aff, tbi = q / ncy;
            cy_ = q - tbi * ncy;
            tb_ = (tile_blocks & ~7) + tbi;
        }
    }
};

// Tail pieces -> outputs: thread = (tail item, tile, output pixel [pooled: pooled pixel], cout quad); the slices are added in slice order
// (deterministic), then bias, leaky-ReLU (+ the 2x2 max of the tile's four pooling windows), float4 store.  CK = channels per step.
template <bool POOL, int TILES, int COUTS, int CK>
__device__ __forceinline__ void w4_tail_reduce_body(const ConvParams& p, long first, long stride) {      // (grid-stride loop bounds from the kernel)
    constexpr int PIECE_FLOATS = TILES * 16 * COUTS;
    const W4GeomT<TILES, COUTS> geo(p);
    const int nitems = geo.tile_blocks * geo.ncy, nfull = nitems - p.tail_items;
    constexpr int PX = POOL ? 4 : 16;
    const int Hs = POOL ? (p.Ho >> 1) : p.Ho, Ws = POOL ? (p.Wo >> 1) : p.Wo;
    const long total = (long)p.tail_items * TILES * PX * (COUTS / 4);
    for (long e = first; e < total; e += stride) {
        const int c4 = (int)(e % (COUTS / 4));
        long r = e / (COUTS / 4);
        const int px = (int)(r % PX); r /= PX;
        const int t = (int)(r % TILES);
        const int ti = (int)(r / TILES);
        int cy, tblock, img, ty, tx;
        geo.item_decode(nfull + ti, cy, tblock);
        geo.tile_decode(tblock * TILES + t, img, ty, tx);
        if (img >= p.B) continue;
        const int co = cy * COUTS + c4 * 4;
        // the pieces of tail item ti, in step order: workgroup w's run [w q, (w + 1) q) of the tail's item-steps meets the item's
        // [ti S, (ti + 1) S); it is the run's first piece (slot 2 w) when the run starts inside the item, else its second (slot 2 w + 1)
        const int S = p.Cin / CK, q = p.tail_q;
        const int w_lo = (ti * S) / q, w_hi = ((ti + 1) * S - 1) / q;
        const float* src = p.partial + (size_t)t * (16 * COUTS) + c4 * 4;
        auto slot_of = [&](int w) { return (size_t)(2 * w + (w * q >= ti * S ? 0 : 1)) * PIECE_FLOATS; };
        const f32x4 bias = *(const f32x4*)(p.bias + co);
        f32x4 res;
        int oy, ox;
        if (POOL) {
            const int pi = px >> 1, pj = px & 1;
            oy = 2 * ty + pi; ox = 2 * tx + pj;
            f32x4 mx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int pix = (2 * pi + (qd >> 1)) * 4 + 2 * pj + (qd & 1);
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                for (int w = w_lo; w <= w_hi; ++w) a += *(const f32x4*)(src + slot_of(w) + pix * COUTS);
#pragma unroll
                for (int j = 0; j < 4; ++j) mx[j] = qd == 0 ? a[j] : fmaxf(mx[j], a[j]);
            }
            res = mx + bias;                      // bias + activation after the max, like the fused epilogue (monotonic: same bits)
        } else {
            oy = 4 * ty + (px >> 2); ox = 4 * tx + (px & 3);
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            for (int w = w_lo; w <= w_hi; ++w) a += *(const f32x4*)(src + slot_of(w) + px * COUTS);
            res = a + bias;
        }
        if (oy >= Hs || ox >= Ws) continue;
        if (p.act) {
#pragma unroll
            for (int j = 0; j < 4; ++j) res[j] = fmaxf(res[j], HP3D_LEAKY_SLOPE * res[j]);
        }
        float* dst = p.out + ((size_t)(img * Hs + oy) * Ws + ox) * p.out_cs + co;
        if (co + 3 < p.cout_store) *(f32x4*)dst = res;
        else
            for (int j = 0; j < 4; ++j) if (co + j < p.cout_store) dst[j] = res[j];
    }
}

}  // namespace

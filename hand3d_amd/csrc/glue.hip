// glue.hip -- the HBM-bound / latency-bound stages between the MFMA convolutions.
// Everything here is compiled with -ffp-contract=off: the reference evaluates these
// expressions op by op in float32 (TF 1.3 CPU/GPU kernels), and the box / interpolation
// arithmetic feeds discontinuous decisions (row validity, floor/ceil), so no FMA contraction.
#include "hp3d_common.h"
#include <type_traits>

namespace {

__device__ __forceinline__ float leaky(float x) { return fmaxf(x, HP3D_LEAKY_SLOPE * x); }

// ---------------------------------------------------------------------------------------
// conv_naive: one thread per output element; HWIO weights as given by the caller.
// Debug cross-check for conv_mfma (hp3d_set_option conv_impl=naive) -- never a fallback.
HP3D_KERNEL(256)
void conv_naive_kernel(const float* x, int B, int H, int W, int Cin, int in_cs, const float* w, const float* bias,
                       int k, int stride, int Cout, int act, float* out, int out_cs, int Ho, int Wo,
                       int pad_t, int pad_l) {
    const long total = (long)B * Ho * Wo * Cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        long r = i / Cout;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float acc = 0.f;
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * stride - pad_t + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * stride - pad_l + kx;
                if (ix < 0 || ix >= W) continue;
                const float* xp = x + (((size_t)b * H + iy) * W + ix) * in_cs;
                const float* wp = w + ((size_t)(ky * k + kx) * Cin) * Cout + co;
                for (int c = 0; c < Cin; ++c) acc = fmaf(xp[c], wp[(size_t)c * Cout], acc);
            }
        }
        acc += bias[co];
        if (act) acc = leaky(acc);
        out[(((size_t)b * Ho + oy) * Wo + ox) * out_cs + co] = acc;
    }
}

// split-K epilogue of conv_mfma / the Winograd kernels: fixed-order sum over the K slices, then bias + leaky-ReLU.
// VEC = 4: a thread owns four consecutive couts (16-byte loads / stores; Cout, out_cs, cout_store multiples of 4, aligned pointers).
// Up to eight slices are requested before the first addition (same order of additions): one load per addition is a chain of
// `ksplit` dependent memory round trips -- 10 us for a 16-way split of a few hundred KB, which is what 25 of PoseNet2D's 56 launches
// at B = 1 were (round 5).
template <int VEC>
HP3D_KERNEL(256)
void conv_splitk_reduce_kernel(const float* partial, int ksplit, long npix, int Cout, const float* bias, int act,
                               float* out, int out_cs, int cout_store) {
    typedef typename std::conditional<VEC == 4, f32x4, float>::type vec_t;
    const int cvec = cout_store / VEC;
    const long total = npix * cvec;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % cvec) * VEC;
        const long pix = i / cvec;
        const float* src = partial + (size_t)pix * Cout + co;
        const size_t zs = (size_t)npix * Cout;
        vec_t v = {};
        int z = 0;
        for (; z + 8 <= ksplit; z += 8) {
            vec_t t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *(const vec_t*)(src + (size_t)(z + u) * zs);
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        if (z + 4 <= ksplit) {
            vec_t t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = *(const vec_t*)(src + (size_t)(z + u) * zs);
#pragma unroll
            for (int u = 0; u < 4; ++u) v += t[u];
            z += 4;
        }
        for (; z < ksplit; ++z) v += *(const vec_t*)(src + (size_t)z * zs);
        v += *(const vec_t*)(bias + co);
        if (act) {
            if constexpr (VEC == 4) { for (int j = 0; j < 4; ++j) v[j] = leaky(v[j]); }
            else v = leaky(v);
        }
        *(vec_t*)(out + pix * out_cs + co) = v;
    }
}

// the same followed by the 2x2 / 2 max-pool of the layer (NetworkOps.max_pool, utils/general.py:61-65): partial sums are
// [ksplit][B, H, W][Cout] at conv resolution, the output is [B, H/2, W/2] -- each of the four pixels is summed in split order,
// biased and activated exactly like the unpooled form, then the maximum is taken
HP3D_KERNEL(256)
void conv_splitk_reduce_pool_kernel(const float* partial, int ksplit, int B, int H, int W, int Cout, const float* bias, int act,
                                    float* out, int out_cs, int cout_store) {
    const int Hp = H / 2, Wp = W / 2;
    const long npix = (long)B * H * W, total = (long)B * Hp * Wp * cout_store;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % cout_store);
        long r = i / cout_store;
        const int ox = (int)(r % Wp); r /= Wp;
        const int oy = (int)(r % Hp);
        const int b = (int)(r / Hp);
        float m = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long pix = ((long)b * H + 2 * oy + (q >> 1)) * W + 2 * ox + (q & 1);
            float v = 0.f;
            for (int z = 0; z < ksplit; ++z) v += partial[((size_t)z * npix + pix) * Cout + co];
            v += bias[co];
            if (act) v = leaky(v);
            m = q == 0 ? v : fmaxf(m, v);
        }
        out[(((long)b * Hp + oy) * Wp + ox) * out_cs + co] = m;
    }
}

// ---------------------------------------------------------------------------------------
// 2x2/2 VALID max-pool (utils/general.py:61-65), standalone (the pipeline uses the fused epilogue)
HP3D_KERNEL(256)
void maxpool2_kernel(const float* x, int B, int H, int W, int C, int in_cs, float* out) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)B * Ho * Wo * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long r = i / C;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const float* p = x + (((size_t)b * H + 2 * oy) * W + 2 * ox) * in_cs + c;
        const float m0 = fmaxf(p[0], p[in_cs]);
        const float m1 = fmaxf(p[(size_t)W * in_cs], p[(size_t)W * in_cs + in_cs]);
        out[i] = fmaxf(m0, m1);
    }
}

// 8x8/8 average pool on sizes divisible by 8 (nets/PosePriorNetwork.py:61); row-major f32 sum
HP3D_KERNEL(256)
void avgpool8_kernel(const float* x, int B, int H, int W, int C, float* out, int out_cs) {
    const int Ho = H / 8, Wo = W / 8;
    const long total = (long)B * Ho * Wo * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long r = i / C;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float s = 0.f;
        for (int dy = 0; dy < 8; ++dy)
            for (int dx = 0; dx < 8; ++dx)
                s += x[(((size_t)b * H + oy * 8 + dy) * W + ox * 8 + dx) * C + c];
        out[(((size_t)b * Ho + oy) * Wo + ox) * out_cs + c] = s / 64.f;
    }
}

// TF 1.3 ResizeBilinear, align_corners=False
__device__ __forceinline__ void resize_coord(int o, float scale, int in_n, int& lo, int& hi, float& t) {
    const float src = (float)o * scale;
    lo = (int)floorf(src);
    hi = min(lo + 1, in_n - 1);
    t = src - (float)lo;
}

template <typename IDX>      // IDX = unsigned when the element count fits 32 bits (3x fewer index instructions)
HP3D_KERNEL(256)
void resize_bilinear_kernel(const float* x, int B, int H, int W, int C, int in_cs, int oh, int ow, float* out) {
    const float hscale = (float)H / (float)oh, wscale = (float)W / (float)ow;
    const IDX total = (IDX)B * oh * ow * C;
    for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * blockDim.x) {
        const int c = (int)(i % (IDX)C);
        IDX r = i / (IDX)C;
        const int ox = (int)(r % (IDX)ow); r /= (IDX)ow;
        const int oy = (int)(r % (IDX)oh);
        const int b = (int)(r / (IDX)oh);
        int y0, y1, x0, x1; float ty, tx;
        resize_coord(oy, hscale, H, y0, y1, ty);
        resize_coord(ox, wscale, W, x0, x1, tx);
        const float* xb = x + (size_t)b * H * W * in_cs + c;
        const float tl = xb[((size_t)y0 * W + x0) * in_cs], tr = xb[((size_t)y0 * W + x1) * in_cs];
        const float bl = xb[((size_t)y1 * W + x0) * in_cs], br = xb[((size_t)y1 * W + x1) * in_cs];
        const float top = tl + (tr - tl) * tx;
        const float bot = bl + (br - bl) * tx;
        out[i] = top + (bot - top) * ty;
    }
}

// The same arithmetic, one workgroup per output ROW: its two source rows (2 x W x C floats) and the 3 x ow column terms sit in LDS, every
// thread makes four consecutive values of the row's ow x C floats and stores them as 16 bytes.  For the 8x up-sampling of the key-point
// score maps (32 x 32 x 21 -> 256 x 256 x 21, 176 MB per 32 images: the element kernel above spends its time on three integer divisions
// and four gathered global loads per value, 1.6 TB/s) this is a streaming store.  Needs (ow x C) % 4 == 0 and 2 W C + 3 ow floats of LDS.
HP3D_KERNEL(256)
void resize_bilinear_rows_kernel(const float* x, int H, int W, int C, int in_cs, int oh, int ow, float* out, float inv_c) {
    HP3D_DYN_SMEM(sm);
    float* rows = sm;                             // [2][W][C]
    int* cx0 = (int*)(sm + 2 * W * C);            // [ow] x0 * C, [ow] x1 * C
    int* cx1 = cx0 + ow;
    float* ctx = (float*)(cx1 + ow);              // [ow] tx
    const int oy = blockIdx.x, b = blockIdx.y;
    const float hscale = (float)H / (float)oh, wscale = (float)W / (float)ow;
    int y0, y1; float ty;
    resize_coord(oy, hscale, H, y0, y1, ty);
    const float* xb = x + (size_t)b * H * W * in_cs;
    for (int i = threadIdx.x; i < W * C; i += blockDim.x) {
        const int px = (int)(((float)i + 0.5f) * inv_c), c = i - px * C;
        rows[i] = xb[((size_t)y0 * W + px) * in_cs + c];
        rows[W * C + i] = xb[((size_t)y1 * W + px) * in_cs + c];
    }
    for (int ox = threadIdx.x; ox < ow; ox += blockDim.x) {
        int x0, x1; float tx;
        resize_coord(ox, wscale, W, x0, x1, tx);
        cx0[ox] = x0 * C; cx1[ox] = x1 * C; ctx[ox] = tx;
    }
    __syncthreads();
    float* orow = out + ((size_t)b * oh + oy) * (size_t)ow * C;
    const float* r0 = rows;
    const float* r1 = rows + W * C;
    for (int q4 = threadIdx.x * 4; q4 < ow * C; q4 += blockDim.x * 4) {
        int ox = (int)(((float)q4 + 0.5f) * inv_c), c = q4 - ox * C;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int a0 = cx0[ox] + c, a1 = cx1[ox] + c;
            const float tx = ctx[ox];
            const float tl = r0[a0], tr = r0[a1], bl = r1[a0], br = r1[a1];
            const float top = tl + (tr - tl) * tx;
            const float bot = bl + (br - bl) * tx;
            v[e] = top + (bot - top) * ty;
            if (++c == C) { c = 0; ++ox; }
        }
        *(f32x4*)(orow + q4) = v;
    }
}

// Input pre-processing on device: uint8 image -> `x/255 - 0.5` (data/BinaryDbReader.py:182,
// run.py:59) -> tf.image.resize_images to the network size (eval_full.py:50, eval2d.py:53), fused:
// 4x less H2D traffic, no float image round trip.  Same float32 op order as the oracle (bit-exact).
HP3D_KERNEL(256)
void preprocess_u8_kernel(const unsigned char* img, int B, int H, int W, int oh, int ow, float* out) {
    const float hscale = (float)H / (float)oh, wscale = (float)W / (float)ow;
    const long total = (long)B * oh * ow * 3;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        long r = i / 3;
        const int ox = (int)(r % ow); r /= ow;
        const int oy = (int)(r % oh);
        const int b = (int)(r / oh);
        int y0, y1, x0, x1; float ty, tx;
        resize_coord(oy, hscale, H, y0, y1, ty);
        resize_coord(ox, wscale, W, x0, x1, tx);
        const unsigned char* ib = img + (size_t)b * H * W * 3 + c;
        const float tl = (float)ib[((size_t)y0 * W + x0) * 3] / 255.0f - 0.5f, tr = (float)ib[((size_t)y0 * W + x1) * 3] / 255.0f - 0.5f;
        const float bl = (float)ib[((size_t)y1 * W + x0) * 3] / 255.0f - 0.5f, br = (float)ib[((size_t)y1 * W + x1) * 3] / 255.0f - 0.5f;
        const float top = tl + (tr - tl) * tx;
        const float bot = bl + (br - bl) * tx;
        out[i] = (oh == H && ow == W) ? tl : top + (bot - top) * ty;
    }
}

// crop_image_from_xy -> tf.image.crop_and_resize (utils/general.py:163-196, App. B.4)
HP3D_KERNEL(256)
void crop_and_resize_kernel(const float* img, int B, int H, int W, int C, const float* center, const float* scale,
                            int crop, float* out) {
    const long total = (long)B * crop * crop;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % crop);
        long r = i / crop;
        const int y = (int)(r % crop);
        const int b = (int)(r / crop);
        // box arithmetic, float32 op by op (utils/general.py:182-190)
        const float cs = (float)crop / scale[b];
        const float half = floorf(cs / 2.0f);
        float y1 = center[b * 2 + 0] - half, y2 = y1 + cs;
        float x1 = center[b * 2 + 1] - half, x2 = x1 + cs;
        y1 = y1 / (float)H; y2 = y2 / (float)H; x1 = x1 / (float)W; x2 = x2 / (float)W;
        const float hs = (crop > 1) ? (y2 - y1) * (float)(H - 1) / (float)(crop - 1) : 0.f;
        const float ws = (crop > 1) ? (x2 - x1) * (float)(W - 1) / (float)(crop - 1) : 0.f;
        const float in_y = (crop > 1) ? y1 * (float)(H - 1) + (float)y * hs : 0.5f * (y1 + y2) * (float)(H - 1);
        const float in_x = (crop > 1) ? x1 * (float)(W - 1) + (float)x * ws : 0.5f * (x1 + x2) * (float)(W - 1);
        float* o = out + (size_t)i * C;
        const bool ok = in_y >= 0.f && in_y <= (float)(H - 1) && in_x >= 0.f && in_x <= (float)(W - 1);
        if (!ok) {
            for (int c = 0; c < C; ++c) o[c] = 0.f;
            continue;
        }
        const int ty0 = (int)floorf(in_y), ty1 = (int)ceilf(in_y);
        const int tx0 = (int)floorf(in_x), tx1 = (int)ceilf(in_x);
        const float ly = in_y - (float)ty0, lx = in_x - (float)tx0;
        const float* ib = img + (size_t)b * H * W * C;
        for (int c = 0; c < C; ++c) {
            const float tl = ib[((size_t)ty0 * W + tx0) * C + c], tr = ib[((size_t)ty0 * W + tx1) * C + c];
            const float bl = ib[((size_t)ty1 * W + tx0) * C + c], br = ib[((size_t)ty1 * W + tx1) * C + c];
            const float top = tl + (tr - tl) * lx;
            const float bot = bl + (br - bl) * lx;
            o[c] = top + (bot - top) * ly;
        }
    }
}

HP3D_KERNEL(256)
void copy_channels_kernel(const float* in, long npix, int C, int in_cs, float* out, int out_cs) {
    const long total = npix * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long p = i / C;
        out[p * out_cs + c] = in[p * in_cs + c];
    }
}

// float32 [npix, in_cs] channels 0..C-1 -> half [npix, out_cs] (score map fed back into the f16 concat buffer)
HP3D_KERNEL(256)
void cvt_channels_f16_kernel(const float* in, long npix, int C, int in_cs, hp3d_f16* out, int out_cs) {
    const long total = npix * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long p = i / C;
        out[p * out_cs + c] = (hp3d_f16)in[p * in_cs + c];
    }
}

// [npix, C] -> [npix, out_cs] zero padded
HP3D_KERNEL(256)
void pad_channels_kernel(const float* in, long npix, int C, float* out, int out_cs) {
    const long total = npix * out_cs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % out_cs);
        const long p = i / out_cs;
        out[i] = (c < C) ? in[p * C + c] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------
// Segmentation post-processing (utils/general.py:233-245): 2-class softmax in float32 with a
// correctly rounded exp (oracle/tf_ops.py:exp_f32_cr), detmap = round-half-even(fg), and the
// first arg-max of fg over the row-major flattened map as a 64-bit key
//    key = fg_bits << 32 | (0xFFFFFFFF - flat_index)       (fg >= 0 so its bits order like uints)
__device__ __forceinline__ float exp_cr(float x) { return (float)exp((double)x); }

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(k, off);
        k = (o > k) ? o : k;
    }
    return k;
}

// One atomicMax per WORKGROUP on the image's key (round 5: one per wave -- 256 serialised 64-bit atomics per image and address -- was two
// thirds of seg_upsample_softmax's 0.07 ms: without them the kernel takes 0.025, timing ablations `scripts/micro/r05_variants/seg_abl.sh`).
// The maximum does not depend on the order: deterministic as before.  Every thread of the workgroup must call it.
__device__ __forceinline__ void block_max_key(unsigned long long* key, unsigned long long best) {
    __shared__ unsigned long long s_best[16];
    best = wave_max_u64(best);
    const int wave = threadIdx.x >> 6, nwave = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) s_best[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nwave; ++i) best = s_best[i] > best ? s_best[i] : best;
        atomicMax(key, best);
    }
}

__device__ __forceinline__ void softmax_det_key(float l0, float l1, unsigned idx, float& fg, unsigned char& det,
                                                unsigned long long& key) {
    const float m = fmaxf(l0, l1);
    const float e0 = exp_cr(l0 - m), e1 = exp_cr(l1 - m);
    const float s = e0 + e1;
    fg = e1 / s;
    det = (unsigned char)(rintf(fg) == 1.0f);
    key = ((unsigned long long)__float_as_uint(fg) << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}

// small [B,hs,ws,cs] (channels 0,1) --x8 legacy bilinear--> hand_scoremap [B,H,W,2] (+ det, key)
HP3D_KERNEL(256)
void seg_upsample_softmax_kernel(const float* small, int B, int hs, int ws, int cs, int H, int W,
                                 float* large, unsigned char* det, float* fgout, unsigned long long* keys) {
    const int b = blockIdx.y;
    const float hscale = (float)hs / (float)H, wscale = (float)ws / (float)W;
    const int npx = H * W;
    unsigned long long best = 0ull;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
        const int oy = i / W, ox = i - oy * W;
        int y0, y1, x0, x1; float ty, tx;
        resize_coord(oy, hscale, hs, y0, y1, ty);
        resize_coord(ox, wscale, ws, x0, x1, tx);
        const float* sb = small + (size_t)b * hs * ws * cs;
        float l[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float tl = sb[((size_t)y0 * ws + x0) * cs + c], tr = sb[((size_t)y0 * ws + x1) * cs + c];
            const float bl = sb[((size_t)y1 * ws + x0) * cs + c], br = sb[((size_t)y1 * ws + x1) * cs + c];
            const float top = tl + (tr - tl) * tx;
            const float bot = bl + (br - bl) * tx;
            l[c] = top + (bot - top) * ty;
        }
        const size_t o = (size_t)b * npx + i;
        if (large) { large[o * 2] = l[0]; large[o * 2 + 1] = l[1]; }
        float fg; unsigned char d; unsigned long long key;
        softmax_det_key(l[0], l[1], (unsigned)i, fg, d, key);
        det[o] = d;
        if (fgout) fgout[o] = fg;
        best = key > best ? key : best;
    }
    block_max_key(&keys[b], best);
}

HP3D_KERNEL(256)
void seg_softmax_kernel(const float* large, int B, int H, int W, unsigned char* det, float* fgout,
                        unsigned long long* keys) {
    const int b = blockIdx.y;
    const int npx = H * W;
    unsigned long long best = 0ull;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
        const size_t o = (size_t)b * npx + i;
        float fg; unsigned char d; unsigned long long key;
        softmax_det_key(large[o * 2], large[o * 2 + 1], (unsigned)i, fg, d, key);
        det[o] = d;
        if (fgout) fgout[o] = fg;
        best = key > best ? key : best;
    }
    block_max_key(&keys[b], best);
}

// ---------------------------------------------------------------------------------------
// Seeded geodesic growth + bounding box (utils/general.py:247-328), one workgroup per image.
// O_0 = {seed};  O_{j+1} = det AND dilate21x21(O_j), j < max(H,W)//10 passes, bit-packed in LDS;
// stops early at a fix-point (exactly equivalent: the iteration is deterministic).
// Row r of the bitmap is WW = ceil(W/32) words + ONE zero guard word (pitch P = WW + 1): pixel x is bit x%32 of word x/32; the guard word is
// the right neighbour of the row's last word and the left neighbour of the next row's first, so the horizontal pass reads its neighbours
// without knowing its column (round 5: `w % WW` and `u / WW` per word and pass were a third of the pass's instructions).
constexpr int MG_R = 12;          // rows per thread in the vertical pass of mask_grow
HP3D_KERNEL(1024)
void mask_grow_kernel(const unsigned char* det, const unsigned long long* keys, int H, int W, int empty_fltmax,
                      float* mask_out, float* center, float* crop_size, float* scale, int* seed_out) {
    HP3D_DYN_SMEM(smem_f);
    const int WW = (W + 31) >> 5, P = WW + 1;
    const int NWORD = H * P;                     // words of a bitmap incl. the guard column
    unsigned* detb = (unsigned*)smem_f;
    unsigned* obj = detb + NWORD + 1;            // obj[-1] and obj[NWORD] exist and stay zero (neighbours of the first / last word)
    unsigned* tmp = obj + NWORD + 1;
    __shared__ int s_changed[2], s_rmin, s_rmax, s_cmin, s_cmax;

    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const unsigned char* d = det + (size_t)b * H * W;
    const unsigned idx = 0xFFFFFFFFu - (unsigned)(keys[b] & 0xFFFFFFFFull);
    const int sy = (int)(idx / (unsigned)W), sx = (int)(idx % (unsigned)W);

    for (int w = tid; w < NWORD; w += nthr) {
        const int y = w / P, wx = w - y * P;
        unsigned bits = 0;
        const unsigned char* row = d + (size_t)y * W + wx * 32;
        if (wx == WW) {
            // guard word
        } else if (wx * 32 + 32 <= W && ((W & 7) == 0)) {          // 4 aligned 8-byte loads instead of 32 byte loads
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned long long v = *(const unsigned long long*)(row + q * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if ((v >> (8 * k)) & 0xffull) bits |= (1u << (q * 8 + k));
            }
        } else {
            for (int k = 0; k < 32; ++k) {
                const int x = wx * 32 + k;
                if (x < W && row[k]) bits |= (1u << k);
            }
        }
        detb[w] = bits;
        obj[w] = (y == sy && wx == (sx >> 5)) ? (1u << (sx & 31)) : 0u;
    }
    if (tid == 0) { s_rmin = 0x7fffffff; s_rmax = -1; s_cmin = 0x7fffffff; s_cmax = -1; s_changed[0] = s_changed[1] = 0; obj[-1] = 0u; obj[NWORD] = 0u; }
    __syncthreads();

    // this thread's first unit of the vertical pass (MG_R rows of one word column), computed once; images beyond 1024 units pay the division
    const int nseg = (H + MG_R - 1) / MG_R, nunits = nseg * WW;
    const int u0seg = tid / WW, u0wx = tid - u0seg * WW;
    const int num_passes = max(H, W) / 10;   // max(s[1], s[2]) // (filter_size // 2)
    for (int pass = 0; pass < num_passes; ++pass) {
        // horizontal dilation, radius 10: three independent LDS reads; the guard words make the row ends (their own results are never read)
        for (int w = tid; w < NWORD; w += nthr) {
            const unsigned long long lo = obj[w - 1], mid = obj[w], hi = obj[w + 1];
            unsigned long long win = (mid << 16) | (lo >> 16) | (hi << 48);
            win = win | (win << 1) | (win >> 1);     // radius 1
            win = win | (win << 2) | (win >> 2);     // radius 3
            win = win | (win << 4) | (win >> 4);     // radius 7
            win = win | (win << 3) | (win >> 3);     // radius 10
            tmp[w] = (unsigned)(win >> 16);
        }
        __syncthreads();
        // (two "changed" flags, one barrier saved per pass: the other parity's flag is cleared HERE -- every thread has read it for the
        //  previous pass's exit test before it arrived at the barrier above, and its next writers come after the next two barriers)
        if (tid == 0) s_changed[(pass + 1) & 1] = 0;
        // vertical dilation, radius 10, AND det.  A thread owns MG_R consecutive rows of one word column: it reads the
        // MG_R + 20 rows it needs once and forms the 21-row ORs by doubling (2, 4, 8, 16 rows, then 16 + 4 + 1), ~10 ORs and
        // 2.7 LDS reads per output instead of 21 + 21
        int changed = 0;
        for (int u = tid, k = 0; u < nunits; u += nthr, ++k) {
            int seg = u0seg, wx = u0wx;
            if (k) { seg = u / WW; wx = u - seg * WW; }
            const int y0 = seg * MG_R;
            unsigned v[MG_R + 20];
            // all MG_R + 20 reads are issued unconditionally (rows outside the image read the unit's own first word and are
            // masked to zero afterwards): no branches, no serialised waits
            const int base = (y0 - 10) * P + wx;
#pragma unroll
            for (int i = 0; i < MG_R + 20; ++i) {
                const bool ok = (unsigned)(y0 - 10 + i) < (unsigned)H;
                const unsigned r = tmp[ok ? base + i * P : wx];
                v[i] = ok ? r : 0u;
            }
            unsigned a2[MG_R + 17], a4[MG_R + 5];
            {
                unsigned a1[MG_R + 19];
#pragma unroll
                for (int i = 0; i < MG_R + 19; ++i) a1[i] = v[i] | v[i + 1];
#pragma unroll
                for (int i = 0; i < MG_R + 17; ++i) a2[i] = a1[i] | a1[i + 2];
                unsigned a3[MG_R + 13];
#pragma unroll
                for (int i = 0; i < MG_R + 13; ++i) a3[i] = a2[i] | a2[i + 4];
#pragma unroll
                for (int i = 0; i < MG_R + 5; ++i) a4[i] = a3[i] | a3[i + 8];
            }
#pragma unroll
            for (int j = 0; j < MG_R; ++j) {
                const int y = y0 + j;
                if (y < H) {
                    const int w = y * P + wx;
                    const unsigned acc = (a4[j] | a2[j + 16] | v[j + 20]) & detb[w];     // rows y-10 .. y+10
                    if (acc != obj[w]) changed = 1;
                    // obj is only read through tmp in this phase -> safe to update in place (its guard words are never written)
                    obj[w] = acc;
                }
            }
        }
        if (changed) s_changed[pass & 1] = 1;
        __syncthreads();
        if (!s_changed[pass & 1]) break;
    }

    // bounding box (calc_center_bb): "x" = row index, "y" = column index
    int rmin = 0x7fffffff, rmax = -1, cmin = 0x7fffffff, cmax = -1;
    for (int w = tid; w < NWORD; w += nthr) {
        const unsigned v = obj[w];
        if (v) {
            const int y = w / P, wx = w - y * P;
            rmin = min(rmin, y); rmax = max(rmax, y);
            cmin = min(cmin, wx * 32 + (__ffs(v) - 1));
            cmax = max(cmax, wx * 32 + (31 - __clz(v)));
        }
    }
    if (rmax >= 0) {
        atomicMin(&s_rmin, rmin); atomicMax(&s_rmax, rmax);
        atomicMin(&s_cmin, cmin); atomicMax(&s_cmax, cmax);
    }
    __syncthreads();
    if (mask_out) {
        float* mo = mask_out + (size_t)b * H * W;
        for (int i = tid; i < H * W; i += nthr) {
            const int y = i / W, x = i - y * W;
            mo[i] = (obj[y * P + (x >> 5)] >> (x & 31)) & 1u ? 1.f : 0.f;
        }
    }
    if (tid == 0) {
        float cx, cy, sz;
        if (s_rmax >= 0) {
            const float xmin = (float)s_rmin, xmax = (float)s_rmax, ymin = (float)s_cmin, ymax = (float)s_cmax;
            cx = 0.5f * (xmax + xmin);
            cy = 0.5f * (ymax + ymin);
            sz = fmaxf(xmax - xmin, ymax - ymin);
        } else if (empty_fltmax) {   // Eigen-3.3 identities: centre finite (0,0), size -inf -> 100
            cx = 0.f; cy = 0.f; sz = 100.f;
        } else {                     // +-inf identities: NaN centre -> (160,160); size -> 100
            cx = 160.f; cy = 160.f; sz = 100.f;
        }
        center[b * 2 + 0] = cx;
        center[b * 2 + 1] = cy;
        if (crop_size) crop_size[b] = sz;
        const float best = sz * 1.25f;                       // nets/ColorHandPose3DNetwork.py:84
        scale[b] = fminf(fmaxf(256.0f / best, 0.25f), 5.0f);  // :85
        if (seed_out) { seed_out[b * 2] = sy; seed_out[b * 2 + 1] = sx; }
    }
}

// ---------------------------------------------------------------------------------------
// Fully connected: out[b][o] = sum_i x[b][i] * W[i][o] + bias[o]  (utils/general.py:112-136)
// Weight-streaming bound (4-9 MB of weights, a few MFLOP): split-K over many workgroups so the whole
// chip streams W once, then a fixed-order reduction (deterministic, no atomics).
//   pass 1: workgroup = 64 outputs x one 128-row K slice x up to 32 batch rows; 4 K-quarters per
//           workgroup, x slice staged transposed in LDS, W rows read coalesced (256 B per row).
//   pass 2: out = act(bias + sum over K slices in index order).
constexpr int FC_KCH = 128, FC_BT = 32;
// x = [x | x2]: columns 0..F1-1 come from x (row stride x_stride), columns F1..Cin-1 from x2 (row stride Cin - F1) -- the
// concat([flatten, hand_side]) of nets/ColorHandPose3DNetwork.py:262-263,297-298 without a copy (x2 = nullptr, F1 = Cin: plain).
// The thread's 32 weights are all requested BEFORE the first multiply-add (round 5: the loop used to load one weight per
// iteration behind a data-dependent exit, 32 dependent HBM / L2 round trips per workgroup = 12-25 us per launch for a few
// MFLOP; the sums are the same, in the same order: rows beyond Cin contribute fmaf(0, 0, acc) = acc).
// conv_s (round 6): 0 = a plain matrix x; S > 0 = the rows are the OUTPUT PIXELS of a 3x3 / stride-2 / SAME convolution of an [n,S,S,conv_c] map (S even:
// TensorFlow pads 0 before, 1 after), row = (image, oy, ox), column gk = (tap r * 3 + s) * conv_c + c -- i.e. the layer's HWIO filter IS the FC
// matrix.  The lifting towers' last stride-2 layers (8x8 -> 4x4 maps: 512 output pixels at B = 32) run here: as implicit GEMMs on 8x8-pixel tiles
// they were the slowest launches of the stage (ViewpointNet/conv_vp_2_2 47 us at 12.8 TFLOP/s).
HP3D_KERNEL(256)
void fc_partial_kernel(const float* x, int B, int Cin, int x_stride, const float* x2, int F1, const float* w, int Cout, float* part, int conv_s, int conv_c) {
    __shared__ float xs[FC_KCH][FC_BT + 4];        // [k][b], pitch 36 floats (16-B aligned rows)
    __shared__ float red[4][FC_BT][64];
    const int o = blockIdx.x * 64 + (threadIdx.x & 63);
    const int kq = threadIdx.x >> 6;               // K quarter: rows kq*32 .. kq*32+31 of the slice
    const int k0 = blockIdx.y * FC_KCH;
    const int b0 = blockIdx.z * FC_BT;
    float wv[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        const int gk = k0 + kq * 32 + u;
        wv[u] = (o < Cout && gk < Cin) ? w[(size_t)gk * Cout + o] : 0.f;
    }
    for (int i = threadIdx.x; i < FC_KCH * FC_BT; i += 256) {
        const int b = i / FC_KCH, kk = i - b * FC_KCH;       // coalesced along k
        const int gb = b0 + b, gk = k0 + kk;
        float v = 0.f;
        if (conv_s) {
            if (gb < B && gk < Cin) {
                const int so = conv_s >> 1, img = gb / (so * so), op = gb - img * so * so, oy = op / so, ox = op - oy * so;
                const int tap = gk / conv_c, c = gk - tap * conv_c, iy = 2 * oy + tap / 3, ix = 2 * ox + tap % 3;
                if (iy < conv_s && ix < conv_s) v = x[((size_t)(img * conv_s + iy) * conv_s + ix) * conv_c + c];
            }
        } else if (gb < B && gk < Cin) v = gk < F1 ? x[(size_t)gb * x_stride + gk] : x2[(size_t)gb * (Cin - F1) + (gk - F1)];
        xs[kk][b] = v;
    }
    __syncthreads();
    float acc[FC_BT];
#pragma unroll
    for (int j = 0; j < FC_BT; ++j) acc[j] = 0.f;
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        const int kk = kq * 32 + u;
#pragma unroll
        for (int j4 = 0; j4 < FC_BT / 4; ++j4) {
            const f32x4 xv = *(const f32x4*)&xs[kk][j4 * 4];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j4 * 4 + e] = fmaf(xv[e], wv[u], acc[j4 * 4 + e]);
        }
    }
#pragma unroll
    for (int j = 0; j < FC_BT; ++j) red[kq][j][threadIdx.x & 63] = acc[j];
    __syncthreads();
    // 256 threads reduce the 4 quarters for 32 x 64 outputs
    for (int i = threadIdx.x; i < FC_BT * 64; i += 256) {
        const int j = i >> 6, oo = i & 63;
        const int gb = b0 + j, go = blockIdx.x * 64 + oo;
        if (gb < B && go < Cout)
            part[((size_t)blockIdx.y * B + gb) * Cout + go] = (red[0][j][oo] + red[1][j][oo]) + (red[2][j][oo] + red[3][j][oo]);
    }
}

HP3D_KERNEL(256)
void fc_reduce_kernel(const float* part, int nslices, int B, int Cout, const float* bias, int act, float* out,
                      int out_stride) {
    const int total = B * Cout;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / Cout, o = i - b * Cout;
        // the slices are added in index order (deterministic); eight loads are in flight at a time -- one load per addition is a chain of
        // nslices (17 / 33) dependent memory round trips, 12-23 us for a few KB (round 5)
        float v = 0.f;
        int s = 0;
        for (; s + 8 <= nslices; s += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = part[((size_t)(s + u) * B + b) * Cout + o];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; s < nslices; ++s) v += part[((size_t)s * B + b) * Cout + o];
        v += bias[o];
        if (act) v = leaky(v);
        out[(size_t)b * out_stride + o] = v;
    }
}

// The tail of a lifting tower as ONE launch (round 6): the fixed-order reduction of the first FC layer's K slices (+ bias, leaky-ReLU), then the two small
// FC layers behind it (PosePrior: 512 -> 512 -> 63, nets/ColorHandPose3DNetwork.py:264-270; ViewpointNet: 256 -> 128 -> 3, :299-307).  As
// fc_partial + fc_reduce each, those were five launches of a few microseconds of work at the launch floor (18-20 us per layer); here a workgroup
// takes FCT_G images through all three steps with the activations in LDS.  Layer 1: thread = (4 consecutive outputs, one K part), the parts added in
// index order; layer 2: thread = (image, output).  Deterministic.
constexpr int FCT_G = 2, FCT_MAXC = 512;
HP3D_KERNEL(256)
void fc_tail_kernel(const float* part0, int ns0, int B, int C0, const float* bias0, int act0, const float* w1, const float* b1, int C1, int act1,
                    const float* w2, const float* b2, int C2, int act2, float* out, int out_stride) {
    __shared__ __attribute__((aligned(16))) float h0[FCT_G][FCT_MAXC];
    __shared__ __attribute__((aligned(16))) float h1[FCT_G][FCT_MAXC];
    __shared__ __attribute__((aligned(16))) float red[8 * FCT_G * 128];          // [K part][image][C1] (C1 * parts = 1024)
    const int tid = threadIdx.x, img0 = blockIdx.x * FCT_G;
    // layer 0: the K slices in index order (what fc_reduce_kernel does), eight loads in flight
    for (int i = tid; i < FCT_G * C0; i += 256) {
        const int g = i / C0, o = i - g * C0, gb = img0 + g;
        float v = 0.f;
        if (gb < B) {
            int sl = 0;
            for (; sl + 8 <= ns0; sl += 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = part0[((size_t)(sl + u) * B + gb) * C0 + o];
#pragma unroll
                for (int u = 0; u < 8; ++u) v += t[u];
            }
            for (; sl < ns0; ++sl) v += part0[((size_t)sl * B + gb) * C0 + o];
            v += bias0[o];
            if (act0) v = leaky(v);
        }
        h0[g][o] = v;
    }
    __syncthreads();
    // layer 1: C1 / 4 threads cover the outputs, 256 / (C1 / 4) K parts
    {
        const int no4 = C1 >> 2, kp = 256 / no4, o4 = (tid % no4) * 4, part = tid / no4;
        const int klen = (C0 + kp - 1) / kp, k0 = part * klen, k1 = min(C0, k0 + klen);
        f32x4 acc[FCT_G];
#pragma unroll
        for (int g = 0; g < FCT_G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (part < kp) {
            int k = k0;
            for (; k + 8 <= k1; k += 8) {
                f32x4 wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) wv[u] = *(const f32x4*)(w1 + (size_t)(k + u) * C1 + o4);
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int g = 0; g < FCT_G; ++g) {
                        const float xv = h0[g][k + u];
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[g][e] = fmaf(xv, wv[u][e], acc[g][e]);
                    }
            }
            for (; k < k1; ++k) {
                const f32x4 wv = *(const f32x4*)(w1 + (size_t)k * C1 + o4);
#pragma unroll
                for (int g = 0; g < FCT_G; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[g][e] = fmaf(h0[g][k], wv[e], acc[g][e]);
            }
#pragma unroll
            for (int g = 0; g < FCT_G; ++g) *(f32x4*)&red[(part * FCT_G + g) * C1 + o4] = acc[g];
        }
        __syncthreads();
        for (int i = tid; i < FCT_G * C1; i += 256) {
            const int g = i / C1, o = i - g * C1;
            float v = 0.f;
            for (int pp = 0; pp < kp; ++pp) v += red[(pp * FCT_G + g) * C1 + o];
            v += b1[o];
            if (act1) v = leaky(v);
            h1[g][o] = v;
        }
    }
    __syncthreads();
    // layer 2 (63 / 3 outputs): thread = (image, output), the K loop in index order
    for (int i = tid; i < FCT_G * C2; i += 256) {
        const int g = i / C2, o = i - g * C2, gb = img0 + g;
        if (gb >= B) continue;
        float v = 0.f;
        int k = 0;
        for (; k + 8 <= C1; k += 8) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = w2[(size_t)(k + u) * C2 + o];
#pragma unroll
            for (int u = 0; u < 8; ++u) v = fmaf(h1[g][k + u], wv[u], v);
        }
        for (; k < C1; ++k) v = fmaf(h1[g][k], w2[(size_t)k * C2 + o], v);
        v += b2[o];
        if (act2) v = leaky(v);
        out[(size_t)gb * out_stride + o] = v;
    }
}

// _get_rot_mat + _flip_right_hand + matmul (nets/ColorHandPose3DNetwork.py:240-245,311-334)
HP3D_KERNEL(64)
void lift_epilogue_kernel(const float* u, const float* can, const float* hand_side, int B, float* rot,
                          float* rel, int do_rot) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (do_rot) {
        const float ux_b = u[b * 3], uy_b = u[b * 3 + 1], uz_b = u[b * 3 + 2];
        const float un = sqrtf(((ux_b * ux_b + uy_b * uy_b) + uz_b * uz_b) + 1e-8f);
        const float st = sinf(un), ct = cosf(un), oc = 1.0f - ct;
        const float nf = 1.0f / un;
        const float ux = ux_b * nf, uy = uy_b * nf, uz = uz_b * nf;
        R[0] = ct + ux * ux * oc; R[1] = ux * uy * oc - uz * st; R[2] = ux * uz * oc + uy * st;
        R[3] = uy * ux * oc + uz * st; R[4] = ct + uy * uy * oc; R[5] = uy * uz * oc - ux * st;
        R[6] = uz * ux * oc - uy * st; R[7] = uz * uy * oc + ux * st; R[8] = ct + uz * uz * oc;
        if (rot) for (int i = 0; i < 9; ++i) rot[b * 9 + i] = R[i];
    }
    const bool right = do_rot && (hand_side[b * 2 + 1] > hand_side[b * 2]);   // argmax(hand_side)==1
    for (int k = 0; k < 21; ++k) {
        const float x = can[b * 63 + k * 3], y = can[b * 63 + k * 3 + 1];
        const float z = right ? -can[b * 63 + k * 3 + 2] : can[b * 63 + k * 3 + 2];
        for (int j = 0; j < 3; ++j) rel[b * 63 + k * 3 + j] = (x * R[j] + y * R[3 + j]) + z * R[6 + j];
    }
}

// bone_rel_trafo_inv (utils/relative_trafo.py:243-295; PosePriorNetwork variants 'local*',
// nets/PosePriorNetwork.py:70-75).  One thread per image walks the 5 finger chains root -> tip.
// The chain transform T (global -> bone frame) is rigid, T = [R | t], so the reference's
//   T_new = Trans_z(-len) * RotX(-ax) * RotY(-ay) * T_parent ;  x = matrix_inverse(T_new) * (0,0,0,1)^T
// is evaluated as R_new = M R, t_new = M t + (0,0,-len), x = -R_new^T t_new  (M = RotX(-ax) RotY(-ay)).
HP3D_KERNEL(64)
void bone_rel_inv_kernel(const float* rel, int B, float* xyz) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int chains[6][4] = {{0, -1, -1, -1}, {4, 3, 2, 1}, {8, 7, 6, 5}, {12, 11, 10, 9}, {16, 15, 14, 13}, {20, 19, 18, 17}};
    for (int ci = 0; ci < 6; ++ci) {
        float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tv[3] = {0, 0, 0};
        for (int k = 0; k < 4; ++k) {
            const int bone = chains[ci][k];
            if (bone < 0) break;
            const float len = rel[b * 63 + bone * 3], ax = -rel[b * 63 + bone * 3 + 1], ay = -rel[b * 63 + bone * 3 + 2];
            const float cx = cosf(ax), sx = sinf(ax), cy = cosf(ay), sy = sinf(ay);
            // M = RotX(ax) * RotY(ay), RotX = [1 0 0; 0 c -s; 0 s c], RotY = [c 0 s; 0 1 0; -s 0 c]
            const float M[9] = {cy, 0.f, sy, sx * sy, cx, -sx * cy, -cx * sy, sx, cx * cy};
            float Rn[9], tn[3];
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = (M[i * 3] * R[j] + M[i * 3 + 1] * R[3 + j]) + M[i * 3 + 2] * R[6 + j];
                tn[i] = (M[i * 3] * tv[0] + M[i * 3 + 1] * tv[1]) + M[i * 3 + 2] * tv[2];
            }
            tn[2] -= len;
            for (int i = 0; i < 9; ++i) R[i] = Rn[i];
            for (int i = 0; i < 3; ++i) tv[i] = tn[i];
            for (int j = 0; j < 3; ++j) xyz[b * 63 + bone * 3 + j] = -((R[j] * tv[0] + R[3 + j] * tv[1]) + R[6 + j] * tv[2]);
        }
    }
}

// detect_keypoints (utils/general.py:331-344): first arg-max per channel; one workgroup per (b,c)
// order-preserving key of a float for np.argmax's ordering: -0.0 and +0.0 are equal (the first one wins), and a NaN beats
// everything (np.argmax returns the first NaN)
__device__ __forceinline__ unsigned ord_f32(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0xFFFFFFFFu;       // NaN, either sign
    if ((u & 0x7FFFFFFFu) == 0u) u = 0u;                           // -0.0 -> +0.0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
HP3D_KERNEL(256)
void argmax2d_kernel(const float* x, int H, int W, int C, int cs, int* out_rc) {
    __shared__ unsigned long long red[4];
    const int c = blockIdx.x, b = blockIdx.y;
    const float* xb = x + (size_t)b * H * W * cs + c;
    unsigned long long best = 0ull;
    for (int i = threadIdx.x; i < H * W; i += blockDim.x) {
        const unsigned long long key = ((unsigned long long)ord_f32(xb[(size_t)i * cs]) << 32) |
                                       (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        best = key > best ? key : best;
    }
    best = wave_max_u64(best);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) best = red[i] > best ? red[i] : best;
        const unsigned idx = 0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull);
        out_rc[((size_t)b * C + c) * 2] = (int)(idx / (unsigned)W);
        out_rc[((size_t)b * C + c) * 2 + 1] = (int)(idx % (unsigned)W);
    }
}

// detect_keypoints + trafo_coords (utils/general.py:331-357) WITHOUT the 256 x 256 x 21 map in HBM: one workgroup per
// (channel, image) holds the h x w score map of its channel in LDS, evaluates tf.image.resize_images' arithmetic
// (resize_bilinear_kernel above, op for op) for every pixel of the oh x ow map and keeps the FIRST maximum in row-major
// order -- exactly np.argmax of the up-sampled map, ties and rounding included (the maximum of the small map times the
// up-sampling factor is NOT always it: a neighbour 1 ulp below the peak can round up to the peak value at an earlier
// interpolated position).  kp_crop [B,C,2] int32 (row, col); kp_image [B,C,2] float64 =
// (kp - crop_size // 2) / scale + center evaluated in double like NumPy does on float64 keypoints and float32 scale / centre.
HP3D_KERNEL(256)
void kp_detect_kernel(const float* sm, int h, int w, int C, int cs, int oh, int ow, const float* scale,
                      const float* center, int* kp_crop, double* kp_image) {
    __shared__ float src[64 * 64];
    __shared__ unsigned long long red[4];
    const int c = blockIdx.x, b = blockIdx.y;
    const float* xb = sm + (size_t)b * h * w * cs + c;
    for (int i = threadIdx.x; i < h * w; i += blockDim.x) src[i] = xb[(size_t)i * cs];
    __syncthreads();
    const float hscale = (float)h / (float)oh, wscale = (float)w / (float)ow;
    unsigned long long best = 0ull;
    for (int oy = threadIdx.x; oy < oh; oy += blockDim.x) {
        int y0, y1; float ty;
        resize_coord(oy, hscale, h, y0, y1, ty);
        for (int ox = 0; ox < ow; ++ox) {
            int x0, x1; float tx;
            resize_coord(ox, wscale, w, x0, x1, tx);
            const float tl = src[y0 * w + x0], tr = src[y0 * w + x1], bl = src[y1 * w + x0], br = src[y1 * w + x1];
            const float top = tl + (tr - tl) * tx;
            const float bot = bl + (br - bl) * tx;
            const float v = top + (bot - top) * ty;
            const unsigned long long key = ((unsigned long long)ord_f32(v) << 32) |
                                           (unsigned long long)(0xFFFFFFFFu - (unsigned)(oy * ow + ox));
            best = key > best ? key : best;
        }
    }
    best = wave_max_u64(best);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) best = red[i] > best ? red[i] : best;
        const unsigned idx = 0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull);
        const int row = (int)(idx / (unsigned)ow), col = (int)(idx % (unsigned)ow);
        const size_t o = ((size_t)b * C + c) * 2;
        if (kp_crop) { kp_crop[o] = row; kp_crop[o + 1] = col; }
        if (kp_image) {
            const double sc = (double)scale[b];
            kp_image[o] = ((double)row - (double)(oh / 2)) / sc + (double)center[b * 2];
            kp_image[o + 1] = ((double)col - (double)(ow / 2)) / sc + (double)center[b * 2 + 1];
        }
    }
}

inline int grid_for(long total, int block = 256, int cap = 256 * 16) {
    long g = (total + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

}  // namespace

// ---- launchers ---------------------------------------------------------------------------
void conv_naive_launch(const float* x, int B, int H, int W, int Cin, int in_cs, const float* w, const float* bias,
                       int k, int stride, int Cout, int act, float* out, int out_cs, int Ho, int Wo, int pad_t,
                       int pad_l, hipStream_t s) {
    HP3D_LAUNCH(conv_naive_kernel, dim3(grid_for((long)B * Ho * Wo * Cout)), dim3(256), 0, s, x, B, H, W, Cin, in_cs,
                w, bias, k, stride, Cout, act, out, out_cs, Ho, Wo, pad_t, pad_l);
}
void conv_splitk_reduce_launch(const float* partial, int ksplit, long npix, int Cout, const float* bias, int act,
                               float* out, int out_cs, int cout_store, hipStream_t s) {
    const bool v4 = !((Cout | out_cs | cout_store) & 3) && !(((uintptr_t)partial | (uintptr_t)bias | (uintptr_t)out) & 15);
    if (v4) HP3D_LAUNCH(conv_splitk_reduce_kernel<4>, dim3(grid_for(npix * (cout_store / 4))), dim3(256), 0, s, partial, ksplit, npix,
                        Cout, bias, act, out, out_cs, cout_store);
    else HP3D_LAUNCH(conv_splitk_reduce_kernel<1>, dim3(grid_for(npix * cout_store)), dim3(256), 0, s, partial, ksplit, npix,
                     Cout, bias, act, out, out_cs, cout_store);
}
void conv_splitk_reduce_pool_launch(const float* partial, int ksplit, int B, int H, int W, int Cout, const float* bias, int act,
                                    float* out, int out_cs, int cout_store, hipStream_t s) {
    HP3D_LAUNCH(conv_splitk_reduce_pool_kernel, dim3(grid_for((long)B * (H / 2) * (W / 2) * cout_store)), dim3(256), 0, s, partial, ksplit,
                B, H, W, Cout, bias, act, out, out_cs, cout_store);
}
void maxpool2_launch(const float* x, int B, int H, int W, int C, int in_cs, float* out, hipStream_t s) {
    HP3D_LAUNCH(maxpool2_kernel, dim3(grid_for((long)B * (H / 2) * (W / 2) * C)), dim3(256), 0, s, x, B, H, W, C,
                in_cs, out);
}
void avgpool8_launch(const float* x, int B, int H, int W, int C, float* out, int out_cs, hipStream_t s) {
    HP3D_LAUNCH(avgpool8_kernel, dim3(grid_for((long)B * (H / 8) * (W / 8) * C)), dim3(256), 0, s, x, B, H, W, C,
                out, out_cs);
}
void resize_bilinear_launch(const float* x, int B, int H, int W, int C, int in_cs, int oh, int ow, float* out,
                            hipStream_t s) {
    const long total = (long)B * oh * ow * C;
    const size_t row_lds = ((size_t)2 * W * C + (size_t)3 * ow) * 4;
    // (exact px = i / C from the float reciprocal needs i < 2^20 or so: W x C and ow x C are far below)
    if ((ow * C) % 4 == 0 && ((uintptr_t)out & 15) == 0 && row_lds <= 60 * 1024 && (long)ow * C < (1L << 20) && (long)W * C < (1L << 20) && B <= 65535 &&
        oh >= 4 * H) {
        HP3D_LAUNCH(resize_bilinear_rows_kernel, dim3(oh, B), dim3(256), row_lds, s, x, H, W, C, in_cs, oh, ow, out, 1.0f / (float)C);
        return;
    }
    if (total < (1L << 31)) {
        auto k32 = resize_bilinear_kernel<unsigned>;
        HP3D_LAUNCH(k32, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, x, B, H, W, C, in_cs, oh, ow, out);
    } else {
        auto k64 = resize_bilinear_kernel<unsigned long>;
        HP3D_LAUNCH(k64, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, x, B, H, W, C, in_cs, oh, ow, out);
    }
}
void preprocess_u8_launch(const unsigned char* img, int B, int H, int W, int oh, int ow, float* out, hipStream_t s) {
    HP3D_LAUNCH(preprocess_u8_kernel, dim3(grid_for((long)B * oh * ow * 3)), dim3(256), 0, s, img, B, H, W, oh, ow, out);
}
void crop_and_resize_launch(const float* img, int B, int H, int W, int C, const float* center, const float* scale,
                            int crop, float* out, hipStream_t s) {
    HP3D_LAUNCH(crop_and_resize_kernel, dim3(grid_for((long)B * crop * crop)), dim3(256), 0, s, img, B, H, W, C,
                center, scale, crop, out);
}
// Streams `n` floats through the caches and keeps nothing (the store below never executes for finite data): what it leaves behind is
// the buffer resident in the memory-side cache for the gather loads of the kernel that follows (option "first_touch").
HP3D_KERNEL(256)
void touch_kernel(const float* p, long n4, float* sink) {
    const f32x4* q = (const f32x4*)p;
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = q[i];
        acc += v[0] + v[1] + v[2] + v[3];
    }
    if (acc == 1.2345e-38f) *sink = acc;
}
void touch_launch(const float* p, size_t nfloats, float* sink, hipStream_t s) {
    const long n4 = (long)(nfloats / 4);
    if (n4 < 1 || ((uintptr_t)p & 15)) return;
    HP3D_LAUNCH(touch_kernel, dim3(grid_for(n4, 256, 256 * 8)), dim3(256), 0, s, p, n4, sink);
}
void copy_channels_launch(const float* in, int npix, int C, int in_cs, float* out, int out_cs, hipStream_t s) {
    HP3D_LAUNCH(copy_channels_kernel, dim3(grid_for((long)npix * C)), dim3(256), 0, s, in, (long)npix, C, in_cs, out,
                out_cs);
}
void cvt_channels_f16_launch(const float* in, int npix, int C, int in_cs, hp3d_f16* out, int out_cs, hipStream_t s) {
    HP3D_LAUNCH(cvt_channels_f16_kernel, dim3(grid_for((long)npix * C)), dim3(256), 0, s, in, (long)npix, C, in_cs, out,
                out_cs);
}
void pad_channels_launch(const float* in, int npix, int C, float* out, int out_cs, hipStream_t s) {
    HP3D_LAUNCH(pad_channels_kernel, dim3(grid_for((long)npix * out_cs)), dim3(256), 0, s, in, (long)npix, C, out,
                out_cs);
}
void seg_upsample_softmax_launch(const float* small, int B, int hs, int ws, int cs, int H, int W,
                                 float* scoremap_large, const MaskBuffers& mb, hipStream_t s) {
    (void)hipMemsetAsync(mb.argmax_key, 0, sizeof(unsigned long long) * B, s);
    const int gx = grid_for((long)H * W, 256, 64);
    HP3D_LAUNCH(seg_upsample_softmax_kernel, dim3(gx, B), dim3(256), 0, s, small, B, hs, ws, cs, H, W, scoremap_large,
                mb.det, mb.fg, mb.argmax_key);
}
void seg_softmax_launch(const float* scoremap_large, int B, int H, int W, const MaskBuffers& mb, hipStream_t s) {
    (void)hipMemsetAsync(mb.argmax_key, 0, sizeof(unsigned long long) * B, s);
    const int gx = grid_for((long)H * W, 256, 64);
    HP3D_LAUNCH(seg_softmax_kernel, dim3(gx, B), dim3(256), 0, s, scoremap_large, B, H, W, mb.det, mb.fg,
                mb.argmax_key);
}
// three bitmaps of H rows x (ceil(W / 32) + 1 guard) words, + the two end guards of the growing one
size_t mask_grow_lds_bytes(int H, int W) { return ((size_t)3 * H * ((W + 31) / 32 + 1) + 2) * sizeof(unsigned); }
void mask_grow_launch(const MaskBuffers& mb, int B, int H, int W, int empty_fltmax, float* mask_out, float* center,
                      float* crop_size, float* scale, int* seed, hipStream_t s) {
    const size_t smem = mask_grow_lds_bytes(H, W);
    static bool attr_done[64] = {};
    if (hp3d_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute((const void*)mask_grow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    HP3D_LAUNCH(mask_grow_kernel, dim3(B), dim3(1024), smem, s, (const unsigned char*)mb.det,
                (const unsigned long long*)mb.argmax_key, H, W, empty_fltmax, mask_out, center, crop_size, scale, seed);
}
size_t fc_scratch_floats(int B, int Cin, int Cout) { return (size_t)((Cin + FC_KCH - 1) / FC_KCH) * B * Cout; }
void fc_launch(const float* x, int B, int Cin, int x_stride, const float* w, const float* bias, int Cout, int act,
               float* out, int out_stride, float* scratch, hipStream_t s, const float* x2, int F1) {
    const int ns = (Cin + FC_KCH - 1) / FC_KCH;
    HP3D_LAUNCH(fc_partial_kernel, dim3((Cout + 63) / 64, ns, (B + FC_BT - 1) / FC_BT), dim3(256), 0, s, x, B, Cin,
                x_stride, x2, x2 ? F1 : Cin, w, Cout, scratch, 0, 0);
    HP3D_LAUNCH(fc_reduce_kernel, dim3(grid_for((long)B * Cout)), dim3(256), 0, s, (const float*)scratch, ns, B, Cout,
                bias, act, out, out_stride);
}
// the K slices only (the reduction happens in the consumer: fc_tail_launch); returns the slice count
int fc_partial_launch(const float* x, int B, int Cin, int x_stride, const float* w, int Cout, float* scratch, hipStream_t s, const float* x2, int F1) {
    const int ns = (Cin + FC_KCH - 1) / FC_KCH;
    HP3D_LAUNCH(fc_partial_kernel, dim3((Cout + 63) / 64, ns, (B + FC_BT - 1) / FC_BT), dim3(256), 0, s, x, B, Cin,
                x_stride, x2, x2 ? F1 : Cin, w, Cout, scratch, 0, 0);
    return ns;
}
int fc_tail_eligible(int C0, int C1, int C2) { return C0 <= FCT_MAXC && C2 >= 1 && C1 <= FCT_MAXC && C1 >= 4 && C1 % 4 == 0 && 256 % (C1 / 4) == 0 && (256 / (C1 / 4)) <= 8; }
void fc_tail_launch(const float* part0, int ns0, int B, int C0, const float* bias0, int act0, const float* w1, const float* b1, int C1, int act1,
                    const float* w2, const float* b2, int C2, int act2, float* out, int out_stride, hipStream_t s) {
    HP3D_LAUNCH(fc_tail_kernel, dim3((B + FCT_G - 1) / FCT_G), dim3(256), 0, s, part0, ns0, B, C0, bias0, act0, w1, b1, C1, act1, w2, b2, C2, act2, out, out_stride);
}
// a 3x3 / stride-2 / SAME convolution of an [n,S,S,C] map (S even) as split-K GEMM over its S/2 x S/2 x n output pixels; w = the HWIO filter;
// scratch: fc_scratch_floats(n * (S/2)^2, 9 C, Cout)
void conv_s2_gemm_launch(const float* x, int n, int S, int C, const float* w_hwio, const float* bias, int Cout, int act, float* out, float* scratch, hipStream_t s) {
    const int rows = n * (S / 2) * (S / 2), K = 9 * C, ns = (K + FC_KCH - 1) / FC_KCH;
    HP3D_LAUNCH(fc_partial_kernel, dim3((Cout + 63) / 64, ns, (rows + FC_BT - 1) / FC_BT), dim3(256), 0, s, x, rows, K, 0, (const float*)nullptr, K, w_hwio, Cout, scratch, S, C);
    HP3D_LAUNCH(fc_reduce_kernel, dim3(grid_for((long)rows * Cout)), dim3(256), 0, s, (const float*)scratch, ns, rows, Cout, bias, act, out, Cout);
}
void lift_epilogue_launch(const float* u, const float* coord_can, const float* hand_side, int B, float* rot,
                          float* coord_rel, int do_flip_rot, hipStream_t s) {
    HP3D_LAUNCH(lift_epilogue_kernel, dim3((B + 63) / 64), dim3(64), 0, s, u, coord_can, hand_side, B, rot, coord_rel,
                do_flip_rot);
}
void bone_rel_inv_launch(const float* rel, int B, float* xyz, hipStream_t s) {
    HP3D_LAUNCH(bone_rel_inv_kernel, dim3((B + 63) / 64), dim3(64), 0, s, rel, B, xyz);
}
void kp_detect_launch(const float* sm, int B, int h, int w, int C, int cs, int oh, int ow, const float* scale,
                      const float* center, int* kp_crop, double* kp_image, hipStream_t s) {
    HP3D_LAUNCH(kp_detect_kernel, dim3(C, B), dim3(256), 0, s, sm, h, w, C, cs, oh, ow, scale, center, kp_crop, kp_image);
}
void argmax2d_launch(const float* x, int B, int H, int W, int C, int cs, int* out_rc, hipStream_t s) {
    HP3D_LAUNCH(argmax2d_kernel, dim3(C, B), dim3(256), 0, s, x, H, W, C, cs, out_rc);
}

// Micro-benchmark + semantics check for the row-piece LDS-DMA that conv_wino4.hip's patch staging uses (round 4).
//   hipcc --offload-arch=gfx950 -O3 dma_rows.hip -o dma_rows && ./dma_rows
// 1. semantics: `buffer_load_dwordx4 ... lds` under a partial EXEC mask: masked-off lanes leave LDS untouched; lanes whose
//    voffset fails the descriptor's range check (negative / past num_records) write ZEROS; num_records = 0 zero-fills a whole row.
// 2. cost: a step of 288 v_mfma_f32_16x16x4_f32 per wave (one wave per SIMD, as in conv_wino4) with a ring of weight loads, plus
//    per step K row pieces (one per plane) of NL active lanes from a cold / hot region, against K x 2 plain 8-byte loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
extern __shared__ __attribute__((aligned(16))) float S[];

__global__ __launch_bounds__(256, 1) void sem_kernel(const float* in, float* out, int cs, int W, int hi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 64 * 4; i += 256) S[hi + i] = -7.f;
    __syncthreads();
    // wave w: row w of `in` ([4][W][cs] floats), 41 active lanes, x0 = -1 (lane 0..3 -> pixel -1: negative offset)
    const int x0 = wave == 3 ? W - 8 : -1;           // wave 3: the row's right end (pixels >= W out of range)
    const int nrec = wave == 2 ? 0 : W * cs * 4;     // wave 2: an invalid row
    rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(in + (size_t)wave * W * cs), 0, nrec, 0x00020000);
    const int voff = (x0 + (lane >> 2)) * cs * 4 + (lane & 3) * 16;
    if (lane < 41) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(S + hi + wave * 256), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * 64 * 4; i += 256) out[i] = S[hi + i];
}

template <int MODE>    // 0: no window traffic, 1: K row pieces by LDS-DMA (a descriptor per row, partial EXEC), 2: 2K plain 8-byte loads per lane consumed at once,
                       // 3: K lean pieces (one descriptor, 64 lanes, per-lane offsets held in registers, channel step in soffset), 5: round 3's window stream (2 x 8 bytes per plane, consumed under plane 29)
__global__ __launch_bounds__(256, 1) void cost_kernel(const float* in, const float* w, float* out, int steps, int K, int NL, int cs, int W,
                                                      int rows_per_wg, int hot) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 64 << 20, 0x00020000);
    rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 1u << 30, 0x00020000);
    constexpr int R = 18;
    f32x4 ring[R];
    f32x4 acc[8];
    f32x2 d[8];
    for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0, 0, 0, 0}; d[i] = f32x2{0, 0}; }
    const int wl = lane * 16 + wave * 1024;
    for (int t = 0; t < R; ++t) ring[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, wl, t * 4096, 0));
    const int voff = ((lane >> 2) - 1) * cs * 4 + (lane & 3) * 16;
    const size_t wgrow = (size_t)(hot ? 0 : blockIdx.x) * rows_per_wg;
    int pvo[20];
    if (MODE == 3) {
#pragma unroll
        for (int i = 0; i < 20; ++i) { const int u = i * 64 + lane, row = u / 41, xq = u - row * 41; pvo[i] = row * W * cs * 4 + ((xq >> 2)) * cs * 4 + (xq & 3) * 16; }
    }
    f32x2 win[36];
    if (MODE == 5) for (int i = 0; i < 36; ++i) win[i] = f32x2{0, 0};
    for (int s = 0; s < steps; ++s) {
        const int rowbase = (int)(wgrow + (size_t)(hot ? 0 : s * 24));
#pragma unroll
        for (int pl = 0; pl < 36; ++pl) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[(pl & 3) * 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[pl % R][e], ring[pl % R][e], acc[(pl & 3) * 2], 0, 0, 0);
                acc[(pl & 3) * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[pl % R][e], ring[pl % R][3 - e], acc[(pl & 3) * 2 + 1], 0, 0, 0);
            }
            ring[pl % R] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, wl, ((s * 36 + pl + R) & 1023) * 4096, 0));
            if (MODE == 1 && pl < K) {
                const int row = rowbase + wave * 24 + pl;
                rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(in + (size_t)row * W * cs + (s & 3) * 16), 0, W * cs * 4, 0x00020000);
                if (lane < NL) __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, LDSP(S + wave * 8192 + pl * 164), 16, voff, 0, 0, 0);
            }
            if (MODE == 3 && pl < K) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ir, LDSP(S + 20480 + wave * 6144 + pl * 256), 16, pvo[pl < 20 ? pl : 19], (int)(((size_t)rowbase + wave * 24) * W * cs * 4) + (s & 3) * 64, 0, 0);
            }
            if (MODE == 5 && pl < 18) {
                const int row = rowbase + wave * 24 + pl;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    win[pl * 2 + j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ir, ((lane >> 3) * 4 + j) * cs * 4 + (lane & 7) * 8, (int)((size_t)row * W * cs * 4) + (s & 3) * 64, 0));
            }
            if (MODE == 5 && pl == 29) {
#pragma unroll
                for (int i = 0; i < 36; ++i) d[i & 7] += win[i];
            }
            if (MODE == 2 && pl < K) {
                const int row = rowbase + wave * 24 + pl;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ir, ((lane >> 3) * 4 + j) * cs * 4 + (lane & 7) * 8, (int)((size_t)row * W * cs * 4) + (s & 3) * 64, 0));
                    d[(pl * 2 + j) & 7] += v;
                }
            }
        }
        if (MODE == 1 || MODE == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (MODE == 1) { d[0][0] += S[threadIdx.x + (s & 15)]; }
        if (MODE == 3) { d[0][0] += S[20480 + threadIdx.x + (s & 15)]; }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    f32x4 t = {0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) { t += acc[i]; t[0] += d[i][0] + d[i][1]; }
    *(f32x4*)(out + (blockIdx.x * 256 + threadIdx.x) * 4) = t;
}

template <int MODE>
static void run(const char* name, const float* in, const float* w, float* out, int K, int NL, int cs, int W, int hot) {
    const int steps = 64, grid = 256;
    hipFuncSetAttribute((const void*)cost_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(cost_kernel<MODE>, dim3(grid), dim3(256), 140 * 1024, 0, in, w, out, steps, K, NL, cs, W, steps * 24, hot);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double us_step = best * 1e3 / 5 / steps;
    printf("%-44s K=%2d NL=%2d %s  %.3f us/step (MFMA-only bound %.3f us at 2.4 GHz)\n", name, K, NL, hot ? "hot " : "cold", us_step, 9216 / 2400.0);
}

int main() {
    const int cs = 256, W = 16;
    float *in, *w, *out;
    const size_t rows = (size_t)256 * 64 * 24 + 128;        // every workgroup and step its own rows (6.4 GB: cold = from HBM)
    hipMalloc(&in, rows * W * cs * 4); hipMalloc(&w, 64 << 20); hipMalloc(&out, 256 * 256 * 16);
    hipMemset(w, 0, 64 << 20);
    std::vector<float> h((size_t)4 * W * cs);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1.f + (float)i;
    hipMemset(in, 0, rows * W * cs * 4);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)sem_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    int bad = 0;
    for (int hi = 0; hi <= 36000; hi += 36000) {          // LDS byte offset 0 and 144000: the DMA base (M0) beyond 64 KB
    hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(256), 150 * 1024, 0, in, out, cs, W, hi);
    std::vector<float> o(4 * 256);
    hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
    for (int wv = 0; wv < 4; ++wv)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                const float got = o[wv * 256 + l * 4 + e];
                const int x0 = wv == 3 ? W - 8 : -1, px = x0 + (l >> 2);
                float want;
                if (l >= 41) want = -7.f;                                  // masked-off lane: LDS untouched
                else if (wv == 2 || px < 0 || px >= W) want = 0.f;         // out of range: zero
                else want = 1.f + (float)(((size_t)wv * W + px) * cs + (l & 3) * 4 + e);
                if (got != want) { if (bad < 10) printf("SEMANTICS wave %d lane %d e %d: got %g want %g\n", wv, l, e, got, want); ++bad; }
            }
    printf("semantics at LDS offset %d: %s (%d mismatches so far)\n", hi * 4, bad ? "FAILED" : "ok", bad);
    }
    run<0>("MFMA + weight ring only", in, w, out, 0, 0, cs, W, 0);
    for (int hot = 0; hot < 2; ++hot) {
        run<1>("LDS-DMA row pieces", in, w, out, 18, 41, cs, W, hot);
        run<1>("LDS-DMA row pieces", in, w, out, 18, 64, cs, W, hot);
        run<1>("LDS-DMA row pieces", in, w, out, 24, 41, cs, W, hot);
        run<1>("LDS-DMA row pieces", in, w, out, 36, 41, cs, W, hot);
        run<2>("plain 8-byte loads, 2 per plane", in, w, out, 18, 0, cs, W, hot);
        run<3>("lean LDS-DMA pieces (64 lanes)", in, w, out, 12, 64, cs, W, hot);
        run<3>("lean LDS-DMA pieces (64 lanes)", in, w, out, 18, 64, cs, W, hot);
        run<3>("lean LDS-DMA pieces (64 lanes)", in, w, out, 24, 64, cs, W, hot);
        run<5>("round-3 window stream (36 x 8 B)", in, w, out, 18, 0, cs, W, hot);
    }
    return bad != 0;
}

// Micro-benchmark (round 6): are back-to-back independent MFMAs of one wave slower than MFMAs with a short instruction in between?
// (mfma32_fill.hip: v_mfma_f32_32x32x16_f16 alone 15.8 ns per MFMA, with two dependent v_mul behind each 14.2.)  Here per instruction kind:
// N x s_nop 0 behind every MFMA, N = 0..4, accumulators round-robin over ACCS tuples, one wave per SIMD, all CUs.
// hipcc --offload-arch=gfx950 -O3 -w mfma_b2b.hip -o mfma_b2b
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int N, int ACCS>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float a0) {
    f32x16 acc16[ACCS];
    f32x4 acc4[ACCS];
    for (int i = 0; i < ACCS; ++i) { for (int r = 0; r < 16; ++r) acc16[i][r] = 0.f; acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    f32x4 a = {a0 + threadIdx.x, a0, 1.f, 2.f}, b = {a0, 3.f, 4.f, a0 * 2};
    float af = a0 + threadIdx.x, bf = a0 * 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            if (KIND == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc16[u % ACCS]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc4[u % ACCS]) : "v"(af), "v"(bf));
            if (KIND == 2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc16[u % ACCS]) : "v"(af), "v"(bf));
            if (KIND == 3) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[u % ACCS]) : "v"(a), "v"(b));
            if (KIND == 4) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc4[u % ACCS]) : "v"(af), "v"(bf));      // accumulators in AGPRs (conv_wino4)
#pragma unroll
            for (int v = 0; v < N; ++v) asm volatile("s_nop 0");
        }
    }
    float s = 0.f;
    for (int i = 0; i < ACCS; ++i) { for (int r = 0; r < 16; ++r) s += acc16[i][r]; s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int N, int ACCS>
void run(const char* name, float* d) {
    const int iters = 1000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f, best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, N, ACCS>), dim3(grid), dim3(256), 0, 0, d, iters, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    printf("%-28s accumulators %d  s_nop x%d  %7.3f ms = %6.2f ns per MFMA\n", name, ACCS, N, best, best * 1e6 / (iters * 32.0));
}

template <int KIND, int ACCS>
void sweep(const char* name, float* d) {
    run<KIND, 0, ACCS>(name, d); run<KIND, 1, ACCS>(name, d); run<KIND, 2, ACCS>(name, d); run<KIND, 3, ACCS>(name, d); run<KIND, 4, ACCS>(name, d);
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    sweep<0, 4>("v_mfma_f32_32x32x16_f16", d);
    sweep<1, 8>("v_mfma_f32_16x16x4_f32", d);
    sweep<1, 2>("v_mfma_f32_16x16x4_f32", d);
    sweep<4, 8>("v_mfma_f32_16x16x4_f32 agpr", d);
    sweep<2, 4>("v_mfma_f32_32x32x2_f32", d);
    sweep<3, 8>("v_mfma_f32_16x16x32_f16", d);
    return 0;
}

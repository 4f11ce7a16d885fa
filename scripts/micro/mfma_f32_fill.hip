// Micro-benchmark (round 6): conv_wino4's step is 288 v_mfma_f32_16x16x4_f32 + 96 v_pk_fma_f32 + 72 v_pk_add_f32 (the F(4x4,3x3) input transform).
// mfma32_fill.hip found packed-f32 VALU NOT hidden under a half-precision MFMA of the same wave.  Here the float32 MFMA: N fillers behind each
// MFMA -- plain v_fma_f32 / v_add_f32 against v_pk_fma_f32 / v_pk_add_f32 doing the same arithmetic in half the instructions.  One wave per SIMD.
// hipcc --offload-arch=gfx950 -O3 -w mfma_f32_fill.hip -o mfma_f32_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float a0) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float af = a0 + threadIdx.x, bf = a0 * 3;
    f32x2 x[8];
    for (int i = 0; i < 8; ++i) x[i] = f32x2{a0 * (i + 1) + threadIdx.x, a0 * i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[u & 7]) : "v"(af), "v"(bf));
#pragma unroll
            for (int v = 0; v < N; ++v) {
                const int j = (u * N + v) & 7, j2 = (j + 3) & 7, j3 = (j + 5) & 7;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[j][v & 1]) : "v"(x[j2][0]), "v"(x[j3][1]), "v"(x[j][v & 1]));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(x[j]) : "v"(x[j2]), "v"(x[j3]), "v"(x[j]));
                if (KIND == 2) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[j][v & 1]) : "v"(x[j2][0]), "v"(x[j3][1]));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x[j]) : "v"(x[j2]), "v"(x[j3]));
                if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(x[j]) : "v"(x[j2]), "v"(x[j3]));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + x[i][0] + x[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int N>
void run(const char* name, float* d) {
    const int iters = 1000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f, best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, N>), dim3(grid), dim3(256), 0, 0, d, iters, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    printf("%-14s x%d behind each v_mfma_f32_16x16x4_f32  %7.3f ms = %6.2f ns per MFMA\n", name, N, best, best * 1e6 / (iters * 32.0));
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    run<0, 0>("(none)", d);
    run<0, 1>("v_fma_f32", d); run<0, 2>("v_fma_f32", d); run<0, 3>("v_fma_f32", d); run<0, 4>("v_fma_f32", d); run<0, 6>("v_fma_f32", d); run<0, 8>("v_fma_f32", d);
    run<1, 1>("v_pk_fma_f32", d); run<1, 2>("v_pk_fma_f32", d); run<1, 3>("v_pk_fma_f32", d); run<1, 4>("v_pk_fma_f32", d);
    run<2, 2>("v_add_f32", d); run<2, 4>("v_add_f32", d); run<2, 6>("v_add_f32", d);
    run<3, 1>("v_pk_add_f32", d); run<3, 2>("v_pk_add_f32", d); run<3, 3>("v_pk_add_f32", d);
    run<4, 1>("v_pk_mul_f32", d); run<4, 2>("v_pk_mul_f32", d);
    return 0;
}

// Micro-benchmark (round 6, conv_h16_first_kernel): one wave per SIMD issuing v_mfma_f32_32x32x16_f16 on four independent accumulators with N filler
// instructions behind each MFMA: how many VALU (independent / a dependent chain), LDS reads or LDS writes issue under the 32 matrix-pipe cycles of the
// SAME wave's MFMA, and what the rest costs.  Second part: the same with TWO waves per SIMD (the other wave's fillers under this wave's MFMAs).
// hipcc --offload-arch=gfx950 -O3 mfma32_fill.hip -o mfma32_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int N, int WPS>
__global__ __launch_bounds__(256 * WPS, 1) void k(float* out, int iters, float a0) {
    extern __shared__ float lds[];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 a = {a0 + threadIdx.x, a0, 1.f, 2.f}, b = {a0, 3.f, 4.f, a0 * 2};
    float x[8];
    f32x4 q[4];
    for (int i = 0; i < 8; ++i) x[i] = a0 * (i + 1) + threadIdx.x;
    for (int i = 0; i < 4; ++i) q[i] = f32x4{a0, a0, a0, a0};
    lds[threadIdx.x] = a0;
    const unsigned laddr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
    __syncthreads();
    constexpr int kind = KIND;
    constexpr bool do_mfma = true, do_fill = true;       // (a role split per wave needs a branch around every MFMA, and an MFMA ignores EXEC: not measurable this way)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (do_mfma) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
            if (do_fill) {
#pragma unroll
                for (int v = 0; v < N; ++v) {
                    const int j = (u * N + v) & 7;
                    if (kind == 0) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[j]) : "v"(x[j]), "v"(x[(j + 3) & 7]));           // independent VALU
                    if (kind == 1) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[0]) : "v"(x[0]), "v"(x[1]));                     // dependent chain
                    if (kind == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(q[j & 3]) : "v"(laddr + (j & 3) * 1024));            // LDS reads (no wait)
                    if (kind == 3) asm volatile("ds_write_b64 %0, %1" :: "v"(laddr / 2), "v"(*(double*)&q[j & 3]) : "memory");     // LDS writes
                    if (kind == 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x[j]) : "v"(x[j]), "v"(x[(j + 1) & 7]));
                    if (kind == 5) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(*(double*)&x[2 * (j & 3)]) : "v"(*(double*)&x[2 * (j & 3)]), "v"(*(double*)&x[2 * ((j + 1) & 3)]));
                    if (kind == 6) asm volatile("s_nop 0");
                }
            }
        }
        if (kind == 2 || kind == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int i = 0; i < 4; ++i) s += q[i][0] + q[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND, int N, int WPS>
void run(const char* name, float* d) {
    const int iters = 2000, grid = 256;
    hipFuncSetAttribute((const void*)k<KIND, N, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f, best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, N, WPS>), dim3(grid), dim3(256 * WPS), 100 * 1024, 0, d, iters, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    const double mf = (double)iters * 16;
    printf("%-34s waves/SIMD %d  N=%d  %7.3f ms = %6.1f ns per MFMA slot\n", name, WPS, N, best, best * 1e6 / mf);
}

int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    run<0, 0, 1>("mfma only", d);
    run<0, 2, 1>("independent v_mul", d); run<0, 4, 1>("independent v_mul", d); run<0, 6, 1>("independent v_mul", d); run<0, 8, 1>("independent v_mul", d); run<0, 12, 1>("independent v_mul", d);
    run<1, 2, 1>("dependent v_mul chain", d); run<1, 4, 1>("dependent v_mul chain", d); run<1, 6, 1>("dependent v_mul chain", d); run<1, 8, 1>("dependent v_mul chain", d);
    run<4, 4, 1>("v_cvt_pk_f16_f32", d); run<4, 8, 1>("v_cvt_pk_f16_f32", d);
    run<5, 4, 1>("v_pk_mul_f32", d); run<5, 8, 1>("v_pk_mul_f32", d);
    run<6, 4, 1>("s_nop 0", d); run<6, 8, 1>("s_nop 0", d);
    run<2, 1, 1>("ds_read_b128", d); run<2, 2, 1>("ds_read_b128", d); run<2, 4, 1>("ds_read_b128", d);
    run<3, 1, 1>("ds_write_b64", d); run<3, 2, 1>("ds_write_b64", d); run<3, 4, 1>("ds_write_b64", d);
    // two waves per SIMD, both doing MFMA + fillers
    run<0, 0, 2>("mfma only", d); run<0, 4, 2>("independent v_mul", d); run<0, 8, 2>("independent v_mul", d); run<1, 8, 2>("dependent v_mul chain", d);
    return 0;
}

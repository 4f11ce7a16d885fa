// Micro-benchmark: what VALU work in the shadow of v_mfma_f32_32x32x2_f32 costs (per MFMA: N x v_add_f32 vs
// N x v_pk_add_f32 vs ds_read / s_add).  hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND, int N>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    __shared__ float lds[4096];
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    f32x2 y[8]; for (int i = 0; i < 8; ++i) y[i] = f32x2{0.f, 0.f};
    f32x2 inc = {a, b};
    lds[threadIdx.x] = a; __syncthreads();
    int sacc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < N; ++v) {
                const int j = (u * N + v) & 7;
                if (KIND == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(y[j].x) : "v"(a));
                if (KIND == 1) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(y[j]) : "v"(inc));
                if (KIND == 2) asm volatile("v_pk_add_f32 %0, %1, %0 neg_lo:[1,0] neg_hi:[1,0]" : "+v"(y[j]) : "v"(inc));
                if (KIND == 3) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sacc));
                if (KIND == 4) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"((threadIdx.x & 63) * 4)); asm volatile("s_waitcnt lgkmcnt(8)"); (void)t; }
                if (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(y[j].x) : "v"(a));
                if (KIND == 6) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(y[j]) : "v"(inc));
            }
        }
    }
    float s = sacc;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += y[i].x + y[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND, int N>
void run(const char* name, float* d) {
    int iters = 1000;
    dim3 grid(512);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k<KIND, N>), grid, dim3(256), 0, 0, d, iters, 1.f, 2.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    double mfma = 20.0 * grid.x * 4 * (double)iters * 16;
    double cyc_per_mfma = (ms * 1e-3) * 2.4e9 / (mfma / 1024.0);   // per SIMD at 2.4 GHz, 2 waves/SIMD share
    printf("%-28s N=%d  %.3f of peak   %.1f cycles/MFMA (+%.1f per extra instr)\n", name, N, 64.0 / cyc_per_mfma, cyc_per_mfma, N ? (cyc_per_mfma - 64.8) / N : 0.0);
}
int main() {
    float* d; hipMalloc(&d, 512 * 256 * 4);
    run<0, 0>("none", d);
    run<0, 2>("v_add_f32", d); run<0, 4>("v_add_f32", d);
    run<1, 2>("v_pk_add_f32", d); run<1, 4>("v_pk_add_f32", d);
    run<2, 4>("v_pk_add_f32 neg", d);
    run<6, 4>("v_pk_mul_f32", d);
    run<5, 4>("v_mov_b32", d);
    run<3, 4>("s_add_i32", d);
    run<4, 4>("ds_read_b32", d);
    return 0;
}

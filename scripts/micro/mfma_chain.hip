// Micro-benchmark: rate of v_mfma_f32_32x32x2_f32 for 1 or 2 dependent accumulator chains per wave, 1 or 2 waves
// per SIMD, with / without VALU adds in the MFMA shadow.  hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS, int VALU>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    f32x16 y; for (int i = 0; i < 16; ++i) y[i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
            if (VALU) {
#pragma unroll
                for (int v = 0; v < VALU; ++v) y[(u * VALU + v) & 15] += a;
            }
        }
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    for (int i = 0; i < 16; ++i) s += y[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS, int VALU>
void run(const char* name, int blocks_per_cu, float* d) {
    int iters = 2000 / CHAINS;
    dim3 grid(256 * blocks_per_cu);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k<CHAINS, VALU>), grid, dim3(256), 0, 0, d, iters, 1.f, 2.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double mfma = 20.0 * grid.x * 4 * (double)iters * 16 * CHAINS;
        double tf = mfma * 4096 / (ms * 1e-3) / 1e12;
        if (rep == 2) printf("%-34s wg/CU=%d  %.1f TFLOP/s  (%.3f of 157.3)  %.2f ms\n", name, blocks_per_cu, tf, tf / 157.3, ms);
    }
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<1, 0>("1 chain", 1, d);  run<1, 0>("1 chain", 2, d);
    run<2, 0>("2 chains", 1, d); run<2, 0>("2 chains", 2, d);
    run<4, 0>("4 chains", 1, d); run<4, 0>("4 chains", 2, d);
    run<1, 2>("1 chain + 2 VALU/mfma", 1, d); run<1, 2>("1 chain + 2 VALU/mfma", 2, d);
    run<1, 4>("1 chain + 4 VALU/mfma", 1, d); run<1, 4>("1 chain + 4 VALU/mfma", 2, d);
    run<1, 8>("1 chain + 8 VALU/mfma", 2, d);
    run<2, 4>("2 chains + 4 VALU/mfma", 1, d); run<2, 4>("2 chains + 4 VALU/mfma", 2, d);
    return 0;
}

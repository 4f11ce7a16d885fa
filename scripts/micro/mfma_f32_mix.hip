// Micro-benchmark (round 6): what an instruction BETWEEN float32 MFMAs costs.  mfma_f32_fill.hip: one v_fma_f32 behind every v_mfma_f32_16x16x4_f32
// lengthens the MFMA slot from 14.5 to 19.8 ns -- the float32 matrix instruction and the VALU are not independent pipes for one wave.
// conv_wino4's step keeps its 168 packed transform instructions in ONE block already, but 42 v_add_u32, ~95 ds_read, 72 buffer_load and ~100 scalar
// instructions sit between MFMA pairs.  Here: 64 MFMAs per iteration (8 accumulator tuples in AGPRs, pairs as in the kernel) and 32 fillers of one kind,
// either ONE behind every MFMA pair (SPREAD) or all 32 in one block behind the 64 MFMAs (BLOCK).  One wave per SIMD, all CUs.
// hipcc --offload-arch=gfx950 -O3 -w mfma_f32_mix.hip -o mfma_f32_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__device__ __forceinline__ void filler(int j, f32x2* x, unsigned* u, f32x4* q, unsigned laddr, const float* g, int& sreg) {
    if (KIND == 1) asm volatile("v_add_u32 %0, %1, %2" : "=v"(u[j & 7]) : "v"(u[(j + 3) & 7]), "v"(u[(j + 5) & 7]));
    if (KIND == 2) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[j & 7][0]) : "v"(x[(j + 3) & 7][0]), "v"(x[(j + 5) & 7][1]), "v"(x[j & 7][0]));
    if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(x[j & 7]) : "v"(x[(j + 3) & 7]), "v"(x[(j + 5) & 7]), "v"(x[j & 7]));
    if (KIND == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(q[j & 3]) : "v"(laddr + (j & 3) * 1024));
    if (KIND == 5) asm volatile("ds_read_b64 %0, %1" : "=v"(x[j & 7]) : "v"(laddr / 2 + (j & 3) * 512));
    if (KIND == 6) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[j & 3]) : "v"(g + (threadIdx.x & 63) * 4 + (j & 3) * 256));
    if (KIND == 7) asm volatile("s_nop 0");
    if (KIND == 8) asm volatile("s_add_u32 %0, %0, 4" : "+s"(sreg) :: "scc");
    if (KIND == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(u[j & 7]) : "v"(u[(j + 3) & 7]));
    if (KIND == 10) asm volatile("ds_write_b64 %0, %1" :: "v"(laddr / 2 + (j & 3) * 512), "v"(x[j & 7]) : "memory");
}

template <int KIND, int MODE>    // MODE 0: no fillers; 1: SPREAD (one per MFMA pair); 2: BLOCK (32 behind the 64 MFMAs); 3: 4 blocks of 8 (behind every 16 MFMAs)
__global__ __launch_bounds__(256, 1) void k(float* out, const float* g, int iters, float a0) {
    extern __shared__ float lds[];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float af = a0 + threadIdx.x, bf = a0 * 3;
    f32x2 x[8]; unsigned u[8]; f32x4 q[4];
    for (int i = 0; i < 8; ++i) { x[i] = f32x2{a0 * (i + 1) + threadIdx.x, a0 * i}; u[i] = threadIdx.x + i; }
    for (int i = 0; i < 4; ++i) q[i] = f32x4{a0, a0, a0, a0};
    lds[threadIdx.x] = a0;
    const unsigned laddr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
    int sreg = 0;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int pr = 0; pr < 32; ++pr) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n\tv_mfma_f32_16x16x4_f32 %1, %2, %3, %1" : "+a"(acc[(2 * pr) & 7]), "+a"(acc[(2 * pr + 1) & 7]) : "v"(af), "v"(bf));
            if (MODE == 1) filler<KIND>(pr, x, u, q, laddr, g, sreg);
            if (MODE == 3 && (pr & 7) == 7) {
#pragma unroll
                for (int j = 0; j < 8; ++j) filler<KIND>(pr + j, x, u, q, laddr, g, sreg);
            }
        }
        if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 32; ++j) filler<KIND>(j, x, u, q, laddr, g, sreg);
        }
        if (KIND == 4 || KIND == 5 || KIND == 10) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (KIND == 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float s = (float)sreg;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + x[i][0] + x[i][1] + (float)u[i];
    for (int i = 0; i < 4; ++i) s += q[i][0] + q[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int MODE>
float run(float* d, const float* g) {
    const int iters = 1000, grid = 256;
    hipFuncSetAttribute((const void*)k<KIND, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f, best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, MODE>), dim3(grid), dim3(256), 64 * 1024, 0, d, g, iters, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    return best * 1e6 / iters;      // ns per iteration (64 MFMAs + 32 fillers)
}

template <int KIND>
void row(const char* name, float* d, const float* g, float base) {
    const float s = run<KIND, 1>(d, g), b = run<KIND, 2>(d, g), b8 = run<KIND, 3>(d, g);
    printf("%-22s spread %7.1f ns (+%5.2f per filler)   4 blocks of 8 %7.1f (+%5.2f)   one block %7.1f (+%5.2f)\n", name, s, (s - base) / 32, b8, (b8 - base) / 32, b, (b - base) / 32);
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    float* g; hipMalloc(&g, 1 << 20); hipMemset(g, 0, 1 << 20);
    const float base = run<7, 0>(d, g);
    printf("64 x v_mfma_f32_16x16x4_f32 alone: %.1f ns (%.2f ns per MFMA)\n", base, base / 64);
    row<1>("v_add_u32", d, g, base); row<9>("v_mov_b32", d, g, base); row<2>("v_fma_f32", d, g, base); row<3>("v_pk_fma_f32", d, g, base);
    row<4>("ds_read_b128", d, g, base); row<5>("ds_read_b64", d, g, base); row<10>("ds_write_b64", d, g, base); row<6>("global_load_dwordx4", d, g, base);
    row<7>("s_nop 0", d, g, base); row<8>("s_add_u32", d, g, base);
    return 0;
}

// Micro-benchmark (round 6): one conv_wino4 "plane" = 4 pairs of v_mfma_f32_16x16x4_f32 + 2 ds_read_b128 (A fragments) + 2 window loads (v_add_u32 +
// buffer_load_dwordx2) + 1 filter load (buffer_load_dwordx4), in different arrangements of the non-MFMA instructions between the pairs.
// mfma_f32_mix.hip: a lone VALU or VMEM instruction between float32 MFMAs costs 4.4 ns, in blocks 1.6 (VALU) / 7.5 (VMEM); LDS and scalar ones are free.
//   A (the kernel today): MM dd MM vb MM vb MM b        B: MM dd MM vvbbb MM MM        C: MM dd MM vvbb MM b MM MM(=pair 4 first)   D: MM dd MM b MM b MM b (no v_add)
//   E: MM dd MM vbvb MM MM b                             F: MM ddvb MM vb MM b MM       G: no VMEM / VALU at all (MM dd MM MM MM)
// hipcc --offload-arch=gfx950 -O3 -w mfma_f32_plane.hip -o mfma_f32_plane
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PAIR(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n\tv_mfma_f32_16x16x4_f32 %1, %2, %3, %1" : "+a"(acc[(2 * (i)) & 7]), "+a"(acc[(2 * (i) + 1) & 7]) : "v"(af), "v"(bf))
#define DD(j) do { asm volatile("ds_read_b128 %0, %1" : "=v"(q[(j) & 3]) : "v"(laddr + ((j) & 3) * 1024)); asm volatile("ds_read_b128 %0, %1 offset:512" : "=v"(q[((j) + 1) & 3]) : "v"(laddr + ((j) & 3) * 1024)); } while (0)
#define VADD(j) asm volatile("v_add_u32 %0, %1, %2" : "=v"(va[(j) & 1]) : "v"(ro[(j) % 6]), "v"(co[((j) / 6) % 6]))
#define WLOAD(j) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(w[(j) & 7]) : "v"(va[(j) & 1]), "s"(rs), "s"(so))
#define WLOAD_NOADD(j) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(w[(j) & 7]) : "v"(ro[(j) % 6]), "s"(rs), "s"(so))
#define BLOAD(j) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(bq[(j) % 6]) : "v"(bv), "s"(rs), "s"(so2))

template <int ARR>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* g, int iters, float a0) {
    extern __shared__ float lds[];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float af = a0 + threadIdx.x, bf = a0 * 3;
    f32x4 q[4], bq[6]; f32x2 w[8];
    unsigned ro[6], co[6], va[2] = {0, 0};
    for (int i = 0; i < 6; ++i) { ro[i] = (threadIdx.x & 63) * 8 + i * 2048; co[i] = i * 512 + (threadIdx.x >> 6) * 16384; }
    for (int i = 0; i < 4; ++i) q[i] = f32x4{a0, a0, a0, a0};
    for (int i = 0; i < 6; ++i) bq[i] = f32x4{a0, a0, a0, a0};
    for (int i = 0; i < 8; ++i) w[i] = f32x2{a0, a0};
    const unsigned bv = (threadIdx.x & 63) * 16;
    lds[threadIdx.x] = a0;
    const unsigned laddr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, 1u << 20, 0x00020000);
    int so = 0, so2 = 65536;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int pl = 0; pl < 8; ++pl) {
            const int j = 2 * pl;
            if (ARR == 0) { PAIR(0); DD(j); PAIR(1); VADD(j); WLOAD(j); PAIR(2); VADD(j + 1); WLOAD(j + 1); PAIR(3); BLOAD(pl); }
            if (ARR == 1) { PAIR(0); DD(j); PAIR(1); VADD(j); VADD(j + 1); WLOAD(j); WLOAD(j + 1); BLOAD(pl); PAIR(2); PAIR(3); }
            if (ARR == 2) { PAIR(0); DD(j); PAIR(1); VADD(j); VADD(j + 1); WLOAD(j); WLOAD(j + 1); PAIR(2); BLOAD(pl); PAIR(3); }
            if (ARR == 3) { PAIR(0); DD(j); PAIR(1); WLOAD_NOADD(j); PAIR(2); WLOAD_NOADD(j + 1); PAIR(3); BLOAD(pl); }
            if (ARR == 4) { PAIR(0); DD(j); PAIR(1); VADD(j); WLOAD(j); VADD(j + 1); WLOAD(j + 1); PAIR(2); PAIR(3); BLOAD(pl); }
            if (ARR == 5) { PAIR(0); DD(j); VADD(j); WLOAD(j); PAIR(1); VADD(j + 1); WLOAD(j + 1); PAIR(2); BLOAD(pl); PAIR(3); }
            if (ARR == 6) { PAIR(0); DD(j); PAIR(1); PAIR(2); PAIR(3); }
            if (ARR == 7) { PAIR(0); DD(j); PAIR(1); PAIR(2); PAIR(3); BLOAD(pl); }                                   // the filter load alone
            if (ARR == 8) { PAIR(0); DD(j); PAIR(1); WLOAD_NOADD(j); WLOAD_NOADD(j + 1); BLOAD(pl); PAIR(2); PAIR(3); }   // three VMEM in one group, no v_add
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + w[i][0] + w[i][1];
    for (int i = 0; i < 4; ++i) s += q[i][0] + q[i][3];
    for (int i = 0; i < 6; ++i) s += bq[i][0] + bq[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)va[0] + (float)va[1];
}

template <int ARR>
void run(const char* name, float* d, const float* g) {
    const int iters = 2000, grid = 256;
    hipFuncSetAttribute((const void*)k<ARR>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f, best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<ARR>), dim3(grid), dim3(256), 64 * 1024, 0, d, g, iters, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    printf("%-44s %7.1f ns per plane (8 MFMAs = %.1f ns at 14.0)\n", name, best * 1e6 / (iters * 8.0), 8 * 14.0);
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    float* g; hipMalloc(&g, 1 << 20); hipMemset(g, 0, 1 << 20);
    run<6>("G: MM dd MM MM MM", d, g);
    run<7>("   MM dd MM MM MM b", d, g);
    run<0>("A: MM dd MM vb MM vb MM b   (today)", d, g);
    run<1>("B: MM dd MM vvbbb MM MM", d, g);
    run<2>("C: MM dd MM vvbb MM b MM", d, g);
    run<4>("E: MM dd MM vbvb MM MM b", d, g);
    run<5>("F: MM ddvb MM vb MM b MM", d, g);
    run<3>("D: MM dd MM b MM b MM b     (no v_add)", d, g);
    run<8>("   MM dd MM bbb MM MM       (no v_add)", d, g);
    return 0;
}

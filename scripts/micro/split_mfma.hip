// Micro-benchmark for the split-operand Winograd kernel (round 6): one wave per SIMD (the 512-register regime of conv_wino4 / conv_wino4s)
//   part 1: v_mfma_f32_16x16x32_bf16 pairs with N VALU instructions of the bf16x3 split (v_cvt_pk_bf16_f32, v_lshlrev, v_and, v_sub_f32)
//           behind each MFMA: how many issue under the 16 matrix-pipe cycles, and what the rest costs;
//   part 2: the filter stream: every wave pulls 1.5 KB fragments (b128 + b64 per lane) from a region all workgroups share, R loads in
//           flight: bytes per clock and CU from L2 (3.5 MB region), from the memory-side cache (56 MB) -- the U stream's ceiling.
// hipcc --offload-arch=gfx950 -O3 split_mfma.hip -o split_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void k_mfma(float* out, long long* cyc, int iters, float a0) {
    extern __shared__ float lds[];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a = {a0 + threadIdx.x, a0, 1.f, 2.f}, b = {a0, 3.f, 4.f, a0 * 2};
    float x[8], r[8];
    unsigned pk[8];
    for (int i = 0; i < 8; ++i) { x[i] = a0 * (i + 1) + threadIdx.x; r[i] = 0.f; pk[i] = 0; }
    lds[threadIdx.x] = a0;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[u & 7]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < N; ++v) {
                const int j = (u * N + v) & 7, j2 = (j + 1) & 7;
                if (KIND == 0) {         // the split's mix, in its proportions: cvt_pk, shl, and, sub, sub
                    const int w = (u * N + v) % 5;
                    if (w == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[j]) : "v"(x[j]), "v"(x[j2]));
                    if (w == 1) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(r[j]) : "v"(pk[j]));
                    if (w == 2) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(r[j2]) : "v"(pk[j]));
                    if (w == 3) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(x[j]) : "v"(x[j]), "v"(r[j]));
                    if (w == 4) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(x[j2]) : "v"(x[j2]), "v"(r[j2]));
                }
                if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[j]) : "v"(x[j]), "v"(x[j2]));
                if (KIND == 2) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r[j]) : "v"(x[j]), "v"(x[j2]));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*(f32x2*)&r[2 * (j & 3)]) : "v"(*(f32x2*)&x[2 * (j & 3)]), "v"(*(f32x2*)&x[2 * ((j + 1) & 3)]));
                if (KIND == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(r[j]) : "v"(x[j2]));
                if (KIND == 5) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(pk[j]) : "v"(x[j]), "v"(x[j2]), "s"(0x07060302));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + x[i] + r[i] + __uint_as_float(pk[i]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int N>
void run_mfma(const char* name, float* d, long long* dc) {
    const int iters = 2000, grid = 256;
    hipFuncSetAttribute((const void*)k_mfma<KIND, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    std::vector<long long> hc(grid * 4);
    double best = 1e30;
    float ms = 0.f, bestms = 1e30f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_mfma<KIND, N>), dim3(grid), dim3(256), 120 * 1024, 0, d, dc, iters, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hc.data(), dc, hc.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto c : hc) s += (double)c;
        best = std::min(best, s / hc.size());
        bestms = std::min(bestms, ms);
    }
    // readcyclecounter = s_memtime: a constant 100 MHz clock on this chip, so use wall time for cycles at the clock the chip ran
    const double mf = (double)iters * 16;
    printf("%-22s N=%d  %8.1f ticks/MFMA   wall %7.3f ms = %6.1f ns/MFMA  (%.1f cycles at 2.1 GHz; +%.1f per filler)\n", name, N, best / mf, bestms,
           bestms * 1e6 / mf, bestms * 1e6 / mf * 2.1, N ? (bestms * 1e6 / mf * 2.1 - 16.5) / N : 0.0);
}

// part 2: the filter stream
template <int R>
__global__ __launch_bounds__(256, 1) void k_stream(const char* src, unsigned region, float* out, int iters, int stagger) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, region, 0x00020000);
    f32x4 q[R]; f32x2 q2[R];
    float s = 0.f;
    // fragment sequence of this wave: 1536 B every 6144 B (the four waves interleave), wrapping inside the region; workgroups start
    // `stagger` fragments apart
    unsigned o = (unsigned)(((unsigned long long)blockIdx.x * stagger * 6144ull + wave * 1536u) % region);
    auto next = [&]() { const unsigned r = o; o += 6144u; if (o + 1536u > region) o = wave * 1536u; return __builtin_amdgcn_readfirstlane(r); };
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const unsigned oo = next();
        q[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, oo, 0));
        q2[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, 1024 + lane * 8, oo, 0));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            s += q[i][0] + q[i][3] + q2[i][1];
            const unsigned oo = next();
            q[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, oo, 0));
            q2[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, 1024 + lane * 8, oo, 0));
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) s += q[i][0] + q2[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
}

template <int R>
void run_stream(const char* src, unsigned region, float* d, int stagger) {
    const int iters = 4000 / R, grid = 256;
    hipFuncSetAttribute((const void*)k_stream<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f, best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_stream<R>), dim3(grid), dim3(256), 120 * 1024, 0, src, region, d, iters, stagger);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    const double bytes = (double)grid * 4 * (iters + 1) * R * 1536.0;
    printf("stream region %6.1f MB  R=%2d stagger %3d: %7.3f ms  %6.2f TB/s  = %5.1f B/clk/CU at 2.1 GHz (%.0f GB/s per CU)\n", region / 1048576.0, R, stagger, best,
           bytes / best * 1e-9, bytes / (best * 1e-3) / 256 / 2.1e9, bytes / (best * 1e-3) / 256 * 1e-9);
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    long long* dc; hipMalloc(&dc, 256 * 4 * 8);
    run_mfma<2, 0>("mfma only", d, dc);
    run_mfma<0, 1>("split mix", d, dc); run_mfma<0, 2>("split mix", d, dc); run_mfma<0, 3>("split mix", d, dc);
    run_mfma<0, 4>("split mix", d, dc); run_mfma<0, 6>("split mix", d, dc); run_mfma<0, 8>("split mix", d, dc);
    run_mfma<1, 2>("v_cvt_pk_bf16_f32", d, dc); run_mfma<1, 4>("v_cvt_pk_bf16_f32", d, dc);
    run_mfma<2, 2>("v_sub_f32", d, dc); run_mfma<2, 4>("v_sub_f32", d, dc); run_mfma<2, 8>("v_sub_f32", d, dc);
    run_mfma<3, 2>("v_pk_add_f32", d, dc); run_mfma<3, 4>("v_pk_add_f32", d, dc);
    run_mfma<4, 4>("v_mov_b32", d, dc);
    run_mfma<5, 2>("v_perm_b32", d, dc); run_mfma<5, 4>("v_perm_b32", d, dc);
    const unsigned big = 56u << 20;
    char* src; hipMalloc(&src, big); hipMemset(src, 1, big);
    for (unsigned region : {3u * 1024 * 1024 + 512 * 1024, 7u << 20, 14u << 20, 56u << 20}) {
        run_stream<4>(src, region, d, 0); run_stream<8>(src, region, d, 0); run_stream<12>(src, region, d, 0);
        run_stream<8>(src, region, d, 1); run_stream<8>(src, region, d, 37);
    }
    return 0;
}

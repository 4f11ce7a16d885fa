// write_bw.hip -- what a store-only kernel can reach on this GPU (the ceiling conv_first is priced against).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/write_bw scripts/micro/write_bw.hip && scripts/micro/write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void fill16(f32x4* p, size_t n, float v) {          // 16 B per lane, lane-contiguous
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = f32x4{v, v, v, v};
}
__global__ void fill4(float* p, size_t n, float v) {            // 4 B per lane, lane-contiguous
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill16_strided(f32x4* p, size_t n, float v) {  // 16 B per lane, 32 B contiguous per lane pair, 256 B stride
    // lane l of a wave writes chunk (l >> 5) of row (l & 31): the pattern of 16-byte stores from D[cout][pixel] registers
    const size_t rows = n / 16;
    for (size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64; w * 32 < rows; w += (size_t)gridDim.x * (blockDim.x / 64)) {
        const int l = threadIdx.x & 63;
        for (int a = 0; a < 8; ++a) p[(w * 32 + (l & 31)) * 16 + a * 2 + (l >> 5)] = f32x4{v, v, v, v};
    }
}
__global__ void copy16(const f32x4* s, f32x4* d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
int main() {
    const size_t bytes = 720ull << 20;
    void *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch, double moved) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %8.3f ms  %7.1f GB/s\n", name, ms / 20, moved / (ms / 20 * 1e-3) / 1e9);
    };
    for (int g : {1024, 4096, 16384}) {
        printf("grid %d x 256\n", g);
        time("fill 16 B/lane coalesced", [&] { fill16<<<g, 256>>>((f32x4*)a, bytes / 16, 1.f); }, (double)bytes);
        time("fill 4 B/lane coalesced", [&] { fill4<<<g, 256>>>((float*)a, bytes / 4, 1.f); }, (double)bytes);
        time("fill 16 B/lane, 32 B per row+instr", [&] { fill16_strided<<<g, 256>>>((f32x4*)a, bytes / 16, 1.f); }, (double)bytes);
        time("copy 16 B/lane (read + write)", [&] { copy16<<<g, 256>>>((const f32x4*)a, (f32x4*)b, bytes / 16); }, 2.0 * bytes);
    }
    hipMemset(a, 0, bytes);
    return 0;
}

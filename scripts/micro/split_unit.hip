// Unit check of conv_wino4s.hip's product step on the GPU: C[16x16] = sum_k A[16][k] B[k][16] over 16 channels through the bf16 x3 split and
// the three v_mfma_f32_16x16x32_bf16 of a step, against float64.   hipcc --offload-arch=gfx950 -O3 -I../../hand3d_amd/csrc split_unit.hip
#include "hp3d_common.h"
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstring>
static unsigned short bf16_rne(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static float bf16_f32(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
template <int mode>
__global__ void k(const float* A, const unsigned* Bpk, float* C) {
    const int lane = threadIdx.x & 63, ln = lane & 15, lq = lane >> 4;
    // A: [16 rows][16 channels] float32; this lane: row ln, channels 4 lq .. + 3
    f32x4 x = *(const f32x4*)(A + ln * 16 + lq * 4);
    unsigned p1a = hp3d_cvt_pk_bf16(x[0], x[1]), p1b = hp3d_cvt_pk_bf16(x[2], x[3]);
    float r0 = x[0] - __builtin_bit_cast(float, p1a << 16), r1 = x[1] - __builtin_bit_cast(float, p1a & 0xffff0000u);
    float r2 = x[2] - __builtin_bit_cast(float, p1b << 16), r3 = x[3] - __builtin_bit_cast(float, p1b & 0xffff0000u);
    unsigned p2a = hp3d_cvt_pk_bf16(r0, r1), p2b = hp3d_cvt_pk_bf16(r2, r3);
    r0 -= __builtin_bit_cast(float, p2a << 16); r1 -= __builtin_bit_cast(float, p2a & 0xffff0000u);
    r2 -= __builtin_bit_cast(float, p2b << 16); r3 -= __builtin_bit_cast(float, p2b & 0xffff0000u);
    unsigned p3a = hp3d_cvt_pk_bf16(r0, r1), p3b = hp3d_cvt_pk_bf16(r2, r3);
    u32x4 a10 = {p2a, p2b, p1a, p1b}, a02 = {p1a, p1b, p3a, p3b};
    // B fragments: [lane][6 dwords]: U1 (2), U0 (2), U2 (2)
    const unsigned* bp = Bpk + lane * 6;
    f32x4 b10 = {__builtin_bit_cast(float, bp[0]), __builtin_bit_cast(float, bp[1]), __builtin_bit_cast(float, bp[2]), __builtin_bit_cast(float, bp[3])};
    f32x4 b02 = {b10[2], b10[3], __builtin_bit_cast(float, bp[4]), __builtin_bit_cast(float, bp[5])};
    f32x4 acc0, acc1;
    // mode = 16 * nops + fillers: `fillers` independent MFMAs (other accumulators) and `nops` wait states between the dependent ones
    f32x4 other[8];
    for (int i = 0; i < 8; ++i) other[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int fillers = mode & 15, nops = mode >> 4;
#define FILL() do { \
        if (fillers >= 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(other[0]) : "v"(a10), "v"(b10)); \
        if (fillers >= 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(other[1]) : "v"(a10), "v"(b10)); \
        if (fillers >= 3) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(other[2]) : "v"(a10), "v"(b10)); \
        if (fillers >= 4) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(other[3]) : "v"(a10), "v"(b10)); \
        if (fillers >= 5) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(other[4]) : "v"(a10), "v"(b10)); \
        if (fillers >= 6) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(other[5]) : "v"(a10), "v"(b10)); \
        if (nops == 1) asm volatile("s_nop 0"); if (nops == 2) asm volatile("s_nop 1"); if (nops == 3) asm volatile("s_nop 2"); \
        if (nops == 4) asm volatile("s_nop 3"); if (nops == 5) asm volatile("s_nop 4"); if (nops == 6) asm volatile("s_nop 5"); \
        if (nops == 7) asm volatile("s_nop 6"); if (nops == 8) asm volatile("s_nop 7"); } while (0)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&a"(acc0) : "v"(a10), "v"(b10));
    FILL();
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a10), "v"(b02));
    FILL();
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a02), "v"(b10));
    acc1 = acc0;
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3");
    for (int r = 0; r < 4; ++r) { C[(4 * lq + r) * 16 + ln] = acc0[r]; C[256 + (4 * lq + r) * 16 + ln] = acc1[r]; }
}
template <int mode>
void run(float* dA, unsigned* dB, float* dC, const std::vector<float>& A, const std::vector<float>& B) {
    hipMemset(dC, 0xff, 2048);
    hipLaunchKernelGGL(k<mode>, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<float> C(512);
    hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost);
    double worst = 0, worst32 = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0; float s32 = 0;
            for (int kk = 0; kk < 16; ++kk) { s += (double)A[i * 16 + kk] * B[kk * 16 + j]; s32 = fmaf(A[i * 16 + kk], B[kk * 16 + j], s32); }
            worst = fmax(worst, fabs(C[i * 16 + j] - s)); worst = fmax(worst, fabs(C[256 + i * 16 + j] - s));
            worst32 = fmax(worst32, fabs(s32 - s));
        }
    printf("fillers %d nops %d: max |C - float64| %.3e   (float32 fmaf chain: %.3e)   C[0][0..3] %g %g %g %g\n", mode & 15, mode >> 4, worst, worst32, C[0], C[1], C[2], C[3]);
}
int main() {
    std::vector<float> A(256), B(256);
    srand(1);
    for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 8.f;
    for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;     // B[k][n]
    std::vector<unsigned> Bpk(64 * 6);
    for (int lane = 0; lane < 64; ++lane) {
        const int n = lane & 15, q = lane >> 4;
        unsigned short pc[4][3];
        for (int e = 0; e < 4; ++e) {
            float x = B[(4 * q + e) * 16 + n];
            pc[e][0] = bf16_rne(x); float r = x - bf16_f32(pc[e][0]); pc[e][1] = bf16_rne(r); r -= bf16_f32(pc[e][1]); pc[e][2] = bf16_rne(r);
        }
        unsigned short* d = (unsigned short*)&Bpk[lane * 6];
        for (int e = 0; e < 4; ++e) { d[e] = pc[e][1]; d[4 + e] = pc[e][0]; d[8 + e] = pc[e][2]; }
    }
    float *dA, *dC; unsigned* dB;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 64 * 24); hipMalloc(&dC, 2048);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, Bpk.data(), 64 * 24, hipMemcpyHostToDevice);
    run<0>(dA, dB, dC, A, B); run<1>(dA, dB, dC, A, B); run<2>(dA, dB, dC, A, B); run<3>(dA, dB, dC, A, B); run<4>(dA, dB, dC, A, B); run<6>(dA, dB, dC, A, B);
    run<16>(dA, dB, dC, A, B); run<32>(dA, dB, dC, A, B); run<48>(dA, dB, dC, A, B); run<64>(dA, dB, dC, A, B); run<128>(dA, dB, dC, A, B);
    run<17>(dA, dB, dC, A, B); run<33>(dA, dB, dC, A, B); run<18>(dA, dB, dC, A, B);
    return 0;
}

// How many issue slots must lie between a VALU write of an MFMA's SrcA register and the MFMA (gfx950)?  One asm block per case: A is set up
// long before, ONE register of it is rewritten N slots in front of v_mfma_f32_16x16x32_bf16; the result tells which value the MFMA read.
// hipcc --offload-arch=gfx950 -O3 mfma_operand_hazard.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CASE(NAME, PRODUCER, GAP)                                                                                         \
    __global__ void NAME(float* out, unsigned xold, unsigned xnew, unsigned one) {                                        \
        f32x4 acc;                                                                                                         \
        u32x4 b = {one, one, one, one};                                                                                    \
        asm volatile("v_mov_b32 v40, %[xo]\n\tv_mov_b32 v41, %[xo]\n\tv_mov_b32 v42, %[xo]\n\tv_mov_b32 v43, %[xo]\n\t"   \
                     "s_nop 7\n\ts_nop 7\n\t" PRODUCER "\n\t" GAP                                                          \
                     "v_mfma_f32_16x16x32_bf16 %[acc], v[40:43], %[b], 0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3"                 \
                     : [acc] "=&a"(acc) : [xo] "v"(xold), [xn] "v"(xnew), [b] "v"(b) : "v40", "v41", "v42", "v43");       \
        out[threadIdx.x] = acc[0];                                                                                         \
    }
CASE(mov0, "v_mov_b32 v41, %[xn]", "")
CASE(mov1, "v_mov_b32 v41, %[xn]", "s_nop 0\n\t")
CASE(mov2, "v_mov_b32 v41, %[xn]", "s_nop 1\n\t")
CASE(mov3, "v_mov_b32 v41, %[xn]", "s_nop 2\n\t")
CASE(movv1, "v_mov_b32 v41, %[xn]", "v_mov_b32 v44, %[xo]\n\t")
CASE(movv2, "v_mov_b32 v41, %[xn]", "v_mov_b32 v44, %[xo]\n\tv_mov_b32 v45, %[xo]\n\t")
CASE(cvt0, "v_cvt_pk_bf16_f32 v41, %[xn], %[xn]", "")
CASE(cvt1, "v_cvt_pk_bf16_f32 v41, %[xn], %[xn]", "s_nop 0\n\t")
CASE(cvt2, "v_cvt_pk_bf16_f32 v41, %[xn], %[xn]", "s_nop 1\n\t")
CASE(cvt3, "v_cvt_pk_bf16_f32 v41, %[xn], %[xn]", "s_nop 2\n\t")
CASE(cvtv1, "v_cvt_pk_bf16_f32 v41, %[xn], %[xn]", "v_mov_b32 v44, %[xo]\n\t")
CASE(cvtv2, "v_cvt_pk_bf16_f32 v41, %[xn], %[xn]", "v_mov_b32 v44, %[xo]\n\tv_mov_b32 v45, %[xo]\n\t")
template <typename K> void run(const char* name, K k, float* d, unsigned xo, unsigned xn, float want_new, float want_old) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, xo, xn, 0x3f803f80u);
    float h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("%-6s lane0 row sum %g  -> %s\n", name, h[0], h[0] == want_new ? "NEW value read (ok)" : h[0] == want_old ? "OLD value read (hazard)" : "?");
}
int main() {
    float* d; hipMalloc(&d, 256);
    // bf16 1.0 = 0x3f80, 2.0 = 0x4000.  A row = 8 bf16 per lane group x 4 groups; B all ones: sum over k of A.
    // old: every A element 1.0 -> sum 32; new: register 1 (k slots 2,3 of every lane group) = 2.0 -> 32 + 8 = 40
    const unsigned one = 0x3f803f80u, two = 0x40004000u;
    run("mov0", mov0, d, one, two, 40.f, 32.f); run("mov1", mov1, d, one, two, 40.f, 32.f); run("mov2", mov2, d, one, two, 40.f, 32.f); run("mov3", mov3, d, one, two, 40.f, 32.f);
    run("movv1", movv1, d, one, two, 40.f, 32.f); run("movv2", movv2, d, one, two, 40.f, 32.f);
    // cvt_pk of (2.0f, 2.0f): pass the float bits 0x40000000 as xn
    run("cvt0", cvt0, d, one, 0x40000000u, 40.f, 32.f); run("cvt1", cvt1, d, one, 0x40000000u, 40.f, 32.f); run("cvt2", cvt2, d, one, 0x40000000u, 40.f, 32.f);
    run("cvt3", cvt3, d, one, 0x40000000u, 40.f, 32.f); run("cvtv1", cvtv1, d, one, 0x40000000u, 40.f, 32.f); run("cvtv2", cvtv2, d, one, 0x40000000u, 40.f, 32.f);
    return 0;
}

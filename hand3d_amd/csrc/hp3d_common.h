// hp3d_common.h -- shared declarations of the gfx950 engine (kernels + executor).
//
// Two build modes:
//   * default          : hipcc --offload-arch=gfx950 (the product, libhp3d.so)
//   * -DHP3D_EMU       : g++ with tests/emu/hp3d_emu.h -- a fiber-per-lane CPU interpreter of the
//                        same kernel sources, used ONLY by the CPU test-suite to check index math
//                        (tile maps, weight packing, masks) without a GPU.  Never shipped.
#pragma once

#ifdef HP3D_EMU
#include "hp3d_emu.h"
#else
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 hp3d_f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((vector_size(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// two float32 -> two bfloat16 (round to nearest even), `lo` in bits 0..15
static __device__ __forceinline__ unsigned hp3d_cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// wave-level rendezvous for data exchanged through LDS inside ONE wave
#define HP3D_WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define HP3D_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) float name[]
#define HP3D_LAUNCH(kern, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__)
#define HP3D_MFMA_32x32x2(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// v_mfma_f32_16x16x4_f32: A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k, D reg r of lane l: column l & 15, row 4 (l >> 4) + r
#define HP3D_MFMA_16x16x4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// 16-B fragments (held as f32x4) reinterpreted as 8 halves: lane l carries k = 8*(l>>5) .. 8*(l>>5)+7
#define HP3D_MFMA_32x32x16_F16(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (a)), __builtin_bit_cast(f16x8, (b)), (c), 0, 0, 0)
// the same instruction as a statement with the accumulator in VGPRs (the epilogue's VALU reads it there) and the A operand (REG_A = "a": an AGPR
// tuple, "v": a VGPR tuple) read where it lives: conv_h16_first_kernel keeps 72 filter fragments = 288 registers resident, 256 of them in
// AGPRs, and hipcc would otherwise copy every AGPR-parked fragment to VGPRs in front of its MFMA (434 v_accvgpr_read per item).  Operands come from registers
// loaded once per kernel (A) and from LDS (B: the compiler's s_waitcnt in front of the statement covers it); _FIRST starts the chain from
// the inline constant 0 (no register clearing in front of an MFMA hipcc cannot see into).  HP3D_MFMA_RESULT_FENCE*: a VALU read of an MFMA result
// needs up to 19 wait states behind a 16-pass MFMA, and hipcc does not count them across an asm statement.
#define HP3D_MFMA_32x32x16_F16_ACC(REG_A, acc, a, b) \
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : REG_A(a), "v"(b))
#define HP3D_MFMA_32x32x16_F16_ACC_FIRST(REG_A, acc, a, b) \
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : REG_A(a), "v"(b))
// max of two floats neither of which is a signalling NaN (matrix-core sums): fmaxf() costs a canonicalising v_max_f32 x, x in front
static __device__ __forceinline__ float hp3d_vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// (the accumulators pass THROUGH the statement: without the data dependence hipcc moved the first VALU read in front of the s_nop pair)
#define HP3D_MFMA_RESULT_FENCE2(r0, r1) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(r0), "+v"(r1))
#define HP3D_MFMA_RESULT_FENCE4(r0, r1, r2, r3) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3))
// ... both operands in VGPRs (conv1_1 inside conv_h16_first_kernel; hipcc's own choice for that chain was AGPR accumulators + 32 v_accvgpr_read,
// and it evicted resident filter fragments to make room)
#define HP3D_MFMA_32x32x16_F16_V(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define HP3D_MFMA_32x32x16_F16_V_FIRST(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b))
// four dependent v_mfma_f32_32x32x2_f32 on one accumulator tuple (an accumulate chain), skipped as a whole when the wave-uniform
// `skip` is non-zero.  The branch lives INSIDE the statement so that the compiler sees one opaque read-modify-write of the
// accumulator: a source-level `if` around 16 MFMAs makes hipcc merge two versions of a 16-register tuple at the join and it
// answers with AGPR <-> scratch copies (profiles/r01_tuning_notes.md, "7x7 mode: skip ...").  Operands come from LDS / buffer
// loads (the compiler's s_waitcnt in front of the statement covers them); the accumulator's previous writer and next reader
// are MFMAs of the same chain or lie a whole step away.
#define HP3D_MFMA4_UNLESS(acc, a4, b4, skip)                                                                       \
    asm volatile("s_cmp_lg_u32 %9, 0\n\t"                                                                          \
                 "s_cbranch_scc1 .Lhp3d_skip%=\n\t"                                                                \
                 "v_mfma_f32_32x32x2_f32 %0, %1, %5, %0\n\t"                                                       \
                 "v_mfma_f32_32x32x2_f32 %0, %2, %6, %0\n\t"                                                       \
                 "v_mfma_f32_32x32x2_f32 %0, %3, %7, %0\n\t"                                                       \
                 "v_mfma_f32_32x32x2_f32 %0, %4, %8, %0\n"                                                          \
                 ".Lhp3d_skip%=:"                                                                                   \
                 : "+a"(acc)                                                                                        \
                 : "v"((a4)[0]), "v"((a4)[1]), "v"((a4)[2]), "v"((a4)[3]), "v"((b4)[0]), "v"((b4)[1]), "v"((b4)[2]),   \
                   "v"((b4)[3]), "s"(skip)                                                                           \
                 : "scc")
// the same for conv_wino2.hip's plane: eight v_mfma_f32_16x16x4_f32 (two accumulator tuples alternating, four k groups), skipped as
// a whole when `skip` is non-zero.  Its accumulators live in arch VGPRs (the kernel has no AGPRs), hence "+v".
#define HP3D_MFMA16_2x4_UNLESS(acc0, acc1, a0, a1, b4, skip)                                                        \
    asm volatile("s_cmp_lg_u32 %14, 0\n\t"                                                                         \
                 "s_cbranch_scc1 .Lhp3d_skip%=\n\t"                                                                \
                 "v_mfma_f32_16x16x4_f32 %0, %2, %10, %0\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %1, %6, %10, %1\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %0, %3, %11, %0\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %1, %7, %11, %1\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %0, %4, %12, %0\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %1, %8, %12, %1\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %0, %5, %13, %0\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %1, %9, %13, %1\n"                                                         \
                 ".Lhp3d_skip%=:"                                                                                   \
                 : "+v"(acc0), "+v"(acc1)                                                                           \
                 : "v"((a0)[0]), "v"((a0)[1]), "v"((a0)[2]), "v"((a0)[3]), "v"((a1)[0]), "v"((a1)[1]), "v"((a1)[2]),   \
                   "v"((a1)[3]), "v"((b4)[0]), "v"((b4)[1]), "v"((b4)[2]), "v"((b4)[3]), "s"(skip)                     \
                 : "scc")
// conv_wino4.hip's plane-step with the accumulators' register class PINNED by the constraint (REG = "a": AGPRs, "v": arch VGPRs): the
// kernel holds 72 accumulator tuples = 288 registers, more than the 256 AGPRs; left to itself hipcc shuttles ~10 tuples per step
// between the two files (80 v_accvgpr moves in the MFMA stream).  Planes 0..31 are pinned to AGPRs, planes 32..35 to VGPRs.
#define HP3D_MFMA16_PLANE_UNLESS(REG, acc0, acc1, a0, a1, b4, skip)                                                  \
    asm volatile("s_cmp_lg_u32 %14, 0\n\t"                                                                         \
                 "s_cbranch_scc1 .Lhp3d_skip%=\n\t"                                                                \
                 "v_mfma_f32_16x16x4_f32 %0, %2, %10, %0\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %1, %6, %10, %1\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %0, %3, %11, %0\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %1, %7, %11, %1\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %0, %4, %12, %0\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %1, %8, %12, %1\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %0, %5, %13, %0\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %1, %9, %13, %1\n"                                                         \
                 ".Lhp3d_skip%=:"                                                                                   \
                 : "+" REG(acc0), "+" REG(acc1)                                                                     \
                 : "v"((a0)[0]), "v"((a0)[1]), "v"((a0)[2]), "v"((a0)[3]), "v"((a1)[0]), "v"((a1)[1]), "v"((a1)[2]),   \
                   "v"((a1)[3]), "v"((b4)[0]), "v"((b4)[1]), "v"((b4)[2]), "v"((b4)[3]), "s"(skip)                     \
                 : "scc")
// the same plane as four statements of two MFMAs (both tile halves of one k quad e): between them the kernel places the plane's loads ONE
// PER GAP, so that each issues under the 64 matrix-core cycles of a pair instead of all of them queueing at the plane boundary
// (profiles/r04_tuning_notes.md section 4: the boundary gap cost 40-80 idle cycles per plane)
#define HP3D_MFMA16_PAIR_UNLESS(REG, acc0, acc1, a0e, a1e, be, skip)                                                \
    asm volatile("s_cmp_lg_u32 %5, 0\n\t"                                                                          \
                 "s_cbranch_scc1 .Lhp3d_skip%=\n\t"                                                                \
                 "v_mfma_f32_16x16x4_f32 %0, %2, %4, %0\n\t"                                                       \
                 "v_mfma_f32_16x16x4_f32 %1, %3, %4, %1\n"                                                          \
                 ".Lhp3d_skip%=:"                                                                                   \
                 : "+" REG(acc0), "+" REG(acc1)                                                                     \
                 : "v"(a0e), "v"(a1e), "v"(be), "s"(skip)                                                           \
                 : "scc")
#define HP3D_MFMA16_PAIR(REG, acc0, acc1, a0e, a1e, be)                                                             \
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %4, %0\n\t"                                                       \
                 "v_mfma_f32_16x16x4_f32 %1, %3, %4, %1"                                                             \
                 : "+" REG(acc0), "+" REG(acc1)                                                                     \
                 : "v"(a0e), "v"(a1e), "v"(be))
#define HP3D_MFMA16_PAIR_FIRST(REG, acc0, acc1, a0e, a1e, be)                                                       \
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %4, 0\n\t"                                                        \
                 "v_mfma_f32_16x16x4_f32 %1, %3, %4, 0"                                                              \
                 : "=&" REG(acc0), "=&" REG(acc1)                                                                   \
                 : "v"(a0e), "v"(a1e), "v"(be))
// conv_wino4s.hip: v_mfma_f32_16x16x32_bf16 on both tile halves with one B fragment (A, B: 4 registers = 8 bfloat16, lane l carries
// k = 8 (l >> 4) .. + 7; D as the f32 16x16x4 form); accumulators pinned by REG like conv_wino4.hip's
#define HP3D_MFMA16B_PAIR(REG, acc0, acc1, a0, a1, b)                                                               \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %4, %0\n\t"                                                     \
                 "v_mfma_f32_16x16x32_bf16 %1, %3, %4, %1"                                                           \
                 : "+" REG(acc0), "+" REG(acc1)                                                                     \
                 : "v"(a0), "v"(a1), "v"(b))
#define HP3D_MFMA16B_PAIR_FIRST(REG, acc0, acc1, a0, a1, b)                                                         \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %4, 0\n\t"                                                      \
                 "v_mfma_f32_16x16x32_bf16 %1, %3, %4, 0"                                                            \
                 : "=&" REG(acc0), "=&" REG(acc1)                                                                   \
                 : "v"(a0), "v"(a1), "v"(b))
// two wait states in front of an asm MFMA whose A / B operand a VALU instruction has just written (the hazard recogniser cannot see the MFMA)
#define HP3D_MFMA_OPERAND_FENCE() asm volatile("s_nop 1")
// conv_wino7.hip: two products on two accumulators (one k quad each) -- neither MFMA waits for its predecessor; accumulators pinned to AGPRs
#define HP3D_MFMA16_X2(acc0, acc1, a0e, a1e, b0e, b1e)                                                              \
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %4, %0\n\t"                                                       \
                 "v_mfma_f32_16x16x4_f32 %1, %3, %5, %1"                                                             \
                 : "+a"(acc0), "+a"(acc1)                                                                           \
                 : "v"(a0e), "v"(a1e), "v"(b0e), "v"(b1e))
// ... and the odd product at the end of a chunk: four dependent MFMAs on one accumulator
#define HP3D_MFMA16_X1(acc, a4, b4)                                                                                 \
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %5, %0\n\t"                                                       \
                 "v_mfma_f32_16x16x4_f32 %0, %2, %6, %0\n\t"                                                       \
                 "v_mfma_f32_16x16x4_f32 %0, %3, %7, %0\n\t"                                                       \
                 "v_mfma_f32_16x16x4_f32 %0, %4, %8, %0"                                                             \
                 : "+a"(acc)                                                                                        \
                 : "v"((a4)[0]), "v"((a4)[1]), "v"((a4)[2]), "v"((a4)[3]), "v"((b4)[0]), "v"((b4)[1]), "v"((b4)[2]), "v"((b4)[3]))
// the first step of an item: the accumulators start from the inline constant 0 (never skipped)
#define HP3D_MFMA16_PLANE_FIRST(REG, acc0, acc1, a0, a1, b4)                                                         \
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %10, 0\n\t"                                                       \
                 "v_mfma_f32_16x16x4_f32 %1, %6, %10, 0\n\t"                                                       \
                 "v_mfma_f32_16x16x4_f32 %0, %3, %11, %0\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %1, %7, %11, %1\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %0, %4, %12, %0\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %1, %8, %12, %1\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %0, %5, %13, %0\n\t"                                                      \
                 "v_mfma_f32_16x16x4_f32 %1, %9, %13, %1"                                                            \
                 : "=&" REG(acc0), "=&" REG(acc1)                                                                   \
                 : "v"((a0)[0]), "v"((a0)[1]), "v"((a0)[2]), "v"((a0)[3]), "v"((a1)[0]), "v"((a1)[1]), "v"((a1)[2]),   \
                   "v"((a1)[3]), "v"((b4)[0]), "v"((b4)[1]), "v"((b4)[2]), "v"((b4)[3]))
// a wave-uniform int held in an SGPR that the compiler cannot trace back to a VGPR (it does propagate
// __builtin_amdgcn_readfirstlane's argument into an inline-asm "s" operand and then fails to assemble)
static __device__ __forceinline__ int hp3d_opaque_sgpr(int uniform_value) {
    int s;
    asm volatile("s_nop 0\n\tv_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(uniform_value));
    return s;
}
#define HP3D_OPAQUE_SGPR(x) hp3d_opaque_sgpr(x)
// s += inc on the scalar unit, opaque to the optimiser (a running offset that must not be re-derived as base + i * stride for every i)
#define HP3D_SADD(s, inc) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s) : "s"(inc) : "scc")
// in-launch hand-off between workgroups (cdna_hip_programming.md Guideline 16, R1 in its counter form): the producers store their
// payload write-through (HP3D_BUFFER_STORE4_SC1) and drain it (vmcnt(0) in every wave, then a barrier), one lane takes a ticket
// with a relaxed agent-scope atomic; the last arriver acquires at agent scope (one lane, then a barrier) before plain loads.
// (A release FENCE instead of write-through stores is correct too but writes back the whole L2 of the XCD per workgroup:
//  measured 140 us per launch with 512 workgroups, profiles/r03_tuning_notes.md.)
#define HP3D_ACQUIRE_AGENT() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#define HP3D_TICKET_AGENT(ptr) __hip_atomic_fetch_add((ptr), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define HP3D_STORE_RELAXED_AGENT(ptr, v) __hip_atomic_store((ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define HP3D_KERNEL(nthr) __global__ __launch_bounds__(nthr)
#define HP3D_KERNEL2(nthr, waves_per_simd) __global__ __launch_bounds__(nthr, waves_per_simd)
#define HP3D_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// keeps a per-lane value in a register as-is (the compiler may not re-derive it from other values)
#define HP3D_OPAQUE_V(x) asm volatile("" : "+v"(x))
// true the first time it is called for (this call site's flag array, current device): kernel function attributes
// (dynamic LDS size) are per device, and one process may hold contexts on several devices
static inline bool hp3d_first_use_on_device(bool (&done)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    if (done[dev]) return false;
    done[dev] = true;
    return true;
}
static inline int hp3d_num_cus() {          // CUs of the current device (cached per device)
    static int cache[64] = {};
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cache[dev]) return cache[dev];
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return cache[dev] = n;
}
// scheduling groups inside one sched-barrier region: "next, n instructions of this kind"
#define HP3D_SG_VALU 0x2
#define HP3D_SG_MFMA 0x8
#define HP3D_SG_VMEM_READ 0x20
#define HP3D_SG_DS_READ 0x100
#define HP3D_SCHED_GROUP(kind, n) __builtin_amdgcn_sched_group_barrier((kind), (n), 0)
#define HP3D_READFIRSTLANE(x) __builtin_amdgcn_readfirstlane(x)
#define HP3D_WAIT_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// buffer-addressed LDS-DMA: 16 B per lane from (rsrc base + per-lane voff + scalar soff) to
// (wave-uniform LDS base) + lane*16; out-of-range offsets return 0 (hardware bounds check).
// The builtins exist in the device pass only; hipcc's host pass (which just needs the kernel stubs)
// sees inert stand-ins.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t hp3d_rsrc_t;
#define HP3D_MAKE_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, (bytes), 0x00020000)
#define HP3D_BUFFER_LDS16(rsrc, lds_wave_base, voff, soff, lane)                                   \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(lds_wave_base), 16, \
                                             (voff), (soff), 0, 0)
// 16 B per lane from (rsrc base + per-lane voff + scalar soff) straight into VGPRs
#define HP3D_BUFFER_LOAD16(rsrc, voff, soff) \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((rsrc), (voff), (soff), 0))
// the same with the non-temporal hint (nt): a stream that is read once and must not displace the lines other loads re-use
#define HP3D_BUFFER_LOAD16_NT(rsrc, voff, soff) \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((rsrc), (voff), (soff), 2))
// 8 B per lane (same addressing / range check)
#define HP3D_BUFFER_LOAD8(rsrc, voff, soff) \
    __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64((rsrc), (voff), (soff), 0))
// 16 B per lane at agent scope (sc1): the consuming side of an in-launch hand-off whose payload was stored write-through
#define HP3D_BUFFER_LOAD16_SC1(rsrc, voff, soff) \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((rsrc), (voff), (soff), 16))
// 4 B per lane from (rsrc base + per-lane voff + scalar soff); out-of-range offsets read 0
#define HP3D_BUFFER_LOAD4(rsrc, voff, soff) \
    __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32((rsrc), (voff), (soff), 0))
// 4 B per lane to (rsrc base + per-lane voff + scalar soff); out-of-range offsets are dropped by the hardware
#define HP3D_BUFFER_STORE4(rsrc, val, voff, soff) \
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)(val)), (rsrc), (voff), (soff), 0)
// the same store with the non-temporal hint (nt)
#define HP3D_BUFFER_STORE4_NT(rsrc, val, voff, soff) \
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)(val)), (rsrc), (voff), (soff), 2)
// the same store written through to memory (sc1): payload of an in-launch hand-off to a workgroup on another XCD -- no
// L2 write-back fence needed afterwards (cdna_hip_programming.md Guideline 16, R1)
#define HP3D_BUFFER_STORE4_SC1(rsrc, val, voff, soff) \
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)(val)), (rsrc), (voff), (soff), 16)
#define HP3D_BUFFER_STORE16_SC1(rsrc, val4, voff, soff) \
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, (val4)), (rsrc), (voff), (soff), 16)
// 2 B per lane (a half); same addressing / range check
#define HP3D_BUFFER_STORE2(rsrc, half_val, voff, soff) \
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (hp3d_f16)(half_val)), (rsrc), (voff), (soff), 0)
// 16 B per lane (same addressing / range check)
#define HP3D_BUFFER_STORE16(rsrc, val4, voff, soff) \
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, (val4)), (rsrc), (voff), (soff), 0)
#else
typedef int hp3d_rsrc_t;
#define HP3D_MAKE_RSRC(ptr, bytes) 0
#define HP3D_BUFFER_LDS16(rsrc, lds_wave_base, voff, soff, lane) ((void)(rsrc), (void)(lds_wave_base), (void)(voff), (void)(soff), (void)(lane))
#define HP3D_BUFFER_LOAD16(rsrc, voff, soff) ((void)(rsrc), (void)(voff), (void)(soff), f32x4{0.f, 0.f, 0.f, 0.f})
#define HP3D_BUFFER_LOAD8(rsrc, voff, soff) ((void)(rsrc), (void)(voff), (void)(soff), f32x2{0.f, 0.f})
#define HP3D_BUFFER_LOAD16_NT(rsrc, voff, soff) ((void)(rsrc), (void)(voff), (void)(soff), f32x4{0.f, 0.f, 0.f, 0.f})
#define HP3D_BUFFER_LOAD16_SC1(rsrc, voff, soff) ((void)(rsrc), (void)(voff), (void)(soff), f32x4{0.f, 0.f, 0.f, 0.f})
#define HP3D_BUFFER_STORE4(rsrc, val, voff, soff) ((void)(rsrc), (void)(val), (void)(voff), (void)(soff))
#define HP3D_BUFFER_STORE4_SC1(rsrc, val, voff, soff) ((void)(rsrc), (void)(val), (void)(voff), (void)(soff))
#define HP3D_BUFFER_STORE4_NT(rsrc, val, voff, soff) ((void)(rsrc), (void)(val), (void)(voff), (void)(soff))
#define HP3D_BUFFER_STORE16_SC1(rsrc, val4, voff, soff) ((void)(rsrc), (void)(val4), (void)(voff), (void)(soff))
#define HP3D_BUFFER_STORE16(rsrc, val4, voff, soff) ((void)(rsrc), (void)(val4), (void)(voff), (void)(soff))
#define HP3D_BUFFER_STORE2(rsrc, half_val, voff, soff) ((void)(rsrc), (void)(half_val), (void)(voff), (void)(soff))
#define HP3D_BUFFER_LOAD4(rsrc, voff, soff) ((void)(rsrc), (void)(voff), (void)(soff), 0.f)
#endif
#endif

#include <stdint.h>

#define HP3D_LEAKY_SLOPE 0.01f

// ---------------------------------------------------------------------------------------
// MFMA implicit-GEMM convolution (conv_mfma.hip).  NHWC float32, channel-padded to 32.
struct ConvParams {
    const float* in;      // [B,H,W,in_cs]; the Cin (multiple of 32) channels start at `in`
    const float* wpk;     // packed weights [tap][Cin/8][Cout/32][h:2][n:32][j:4]
    const float* bias;    // [Cout] (padded couts: 0)
    float* out;           // [B,Ho',Wo',out_cs] (Ho',Wo' halved when pooling); channel 0 at `out`
    int B, H, W;          // input extent
    int Ho, Wo;           // conv output extent (before the fused 2x2 max-pool)
    int Cin, in_cs;       // iterated channels (multiple of 32), input channel stride
    int Cout, out_cs;     // padded couts (multiple of 32), output channel stride
    int cout_store;       // couts actually written (<= Cout)
    int pad_t, pad_l;     // TF SAME "before" padding
    int tiles_x, tiles_y; // spatial tiles per image
    int act;              // HP3D_ACT_*
    int ksplit;           // >1: split-K over grid.z, raw partial sums go to `partial`
    float* partial;       // [ksplit][B*Ho*Wo][Cout]
    int f16;              // 1: half-precision operands (channel counts/strides above are in 4-byte units = f16 pairs)
    int out_f32;          // f16 mode only: store float32 (score-map heads) instead of halves
    int im2col;           // 1: `in` is a raw [B,H,W,3] image, the A tile is built as a 3x3 im2col row (conv1_1)
    int nsub;             // conv_wino: 1 = 3x3 filter; 9 = 7x7 filter as 3x3 blocks of its zero-extended 9x9 form
    // conv_h16 fused first block (conv1_1 -> conv1_2 + pool): `in` is the raw [B,H,W,3] float32 image and these are conv1_1's
    // packed weights (mode 1, as conv_first reads them) and bias; nullptr otherwise
    const float* wpk1 = nullptr;
    const float* bias1 = nullptr;
    int first_form = 0;       // 0 = by size, 1 = two workgroups per CU with a filter ring, 2 = one per CU with conv1_2's filters resident in registers
    // conv_wino4.hip tail pieces (set by conv_wino4_launch from `partial` / `partial_cap`): the last, under-filled round of items
    // (tail_items of them) is shared out in runs of tail_q channel steps per workgroup, raw sums to `partial`; 0 = off
    int tail_items = 0, tail_q = 0;
    size_t partial_cap = 0;   // floats available behind `partial` (conv_wino4 tail pieces need conv_wino4_tail_floats())
};

// ---- launchers implemented in the .hip files (all stream-ordered, no sync) -------------
struct ConvPlan {      // chosen by conv_mfma_plan() from the layer geometry
    int th, tw, bn;    // spatial tile (output pixels) and cout tile
    int variant;       // index into the instantiation table
    int ksplit;        // >1: split the Cin chunks over grid.z (under-filled chip)
};
int conv_mfma_plan(int k, int stride, int Ho, int Wo, int Cin, int Cout, int pool, int B, ConvPlan* plan);
// out[pix][co] = act(bias[co] + sum_z partial[z][pix][co]) for co < cout_store
void conv_splitk_reduce_launch(const float* partial, int ksplit, long npix, int Cout, const float* bias, int act,
                               float* out, int out_cs, int cout_store, hipStream_t s);
// the same + the layer's 2x2 / 2 max-pool: partial = [ksplit][B,H,W][Cout], out = [B,H/2,W/2,out_cs]
void conv_splitk_reduce_pool_launch(const float* partial, int ksplit, int B, int H, int W, int Cout, const float* bias, int act,
                                    float* out, int out_cs, int cout_store, hipStream_t s);
int conv_mfma_launch(const ConvParams& p, int k, int stride, int pool, const ConvPlan& plan, hipStream_t s);
const char* conv_mfma_variant_name(int k, int stride, int pool, const ConvPlan& plan);

// Winograd F(2x2,3x3) form of the 3x3/stride-1 layers with Cout % 128 == 0 (conv_wino.hip)
void wino_pack_weights(const float* g_hwio, int k, int Cin, int Cout, int cin_pad, int cout_pad, const int* chan_map, float* dst);
// half-precision 3x3 trunk layers on their own kernel (conv_h16.hip): returns per-wave cout blocks (1, 2, 4) or 0
int conv_h16_eligible(int mode, int k, int stride, int cin_units, int Cout, int Ho, int Wo, int B, int out_f32, int out_cs);
int conv_h16_launch(const ConvParams& p, int pool, hipStream_t s);
// conv1_1 (3 -> 64) computed per patch inside conv1_2 (64 -> 64, pooled): p.in = image, p.wpk1 / p.bias1 = conv1_1
int conv_h16_fused12_launch(const ConvParams& p, hipStream_t s);
int conv_first_eligible(int k, int stride, int Cin, int Cout, int B, int H, int W, int out_cs, int f16);
int conv_first_launch(const ConvParams& p, hipStream_t s, int balanced = 1);
int conv_wino_eligible(int mode, int k, int stride, int Cin, int Cout, int Ho, int Wo, int B, int in_cs, int out_cs,
                       int pool, int* ksplit);
size_t wino_packed_floats(int k, int cin_pad, int cout_pad);
// the same transform on two workgroups per CU (conv_wino2.hip): v_mfma_f32_16x16x4_f32, 32 tiles x 64 couts x 16-channel steps
void wino2_pack_weights(const float* g_hwio, int k, int Cin, int Cout, int cin_pad, int cout_pad, const int* chan_map, float* dst);
int conv_wino2_eligible(int k, int stride, int Cin, int Cout, int Ho, int Wo, int B, int in_cs, int out_cs, int pool, int* ksplit);
int conv_wino2_launch(const ConvParams& p, int pool, hipStream_t s);
// conv_wino4.hip: Winograd F(4x4,3x3) (36 planes, one workgroup per CU); packed filters [36][step][Cout/16][q][n][e]
size_t wino4_packed_floats(int k, int cin_pad, int cout_pad);
void wino4_pack_weights(const float* g_hwio, int k, int Cin, int Cout, int cin_pad, int cout_pad, const int* chan_map, float* dst);
int conv_wino4_eligible(int k, int stride, int Cin, int Cout, int Ho, int Wo, int B, int in_cs, int out_cs, int pool, int* ksplit);
int conv_wino4_launch(const ConvParams& p, int pool, hipStream_t s);
// conv_wino4s.hip (round 6): the same F(4x4,3x3) with the plane products on v_mfma_f32_16x16x32_bf16 over three bfloat16 pieces per
// operand (six products, f32 accumulate); filters pre-split, 6 bytes per value: [36][step][Cout/64][4]{[U1|U0] 16 B x 64 lanes, U2 8 B x 64 lanes}
size_t wino4s_packed_bytes(int cin_pad, int cout_pad);
void wino4s_pack_weights(const float* g_hwio, int Cin, int Cout, int cin_pad, int cout_pad, const int* chan_map, void* dst);
int conv_wino4s_eligible(int k, int stride, int Cin, int Cout, int Ho, int Wo, int B, int in_cs, int out_cs, int pool, int* filled);
int conv_wino4s_launch(const ConvParams& p, int pool, hipStream_t s);
size_t conv_wino4s_tail_floats();
// scratch (floats) the tail pieces of any conv_wino4 launch can need: two 32-tile x 64-cout blocks of raw 4x4 outputs per CU
size_t conv_wino4_tail_floats();
// conv_wino7.hip: the 7x7 layers as Winograd F(4x4,4x4) over the filter's four 4x4-tap blocks (49 planes, V shared by the blocks);
// packed filters [chunk Cin/16][169 non-zero (block, plane) products][Cout/16][q][n][e]
size_t wino7_packed_floats(int cin_pad, int cout_pad);
void wino7_pack_weights(const float* g_hwio, int Cin, int Cout, int cin_pad, int cout_pad, const int* chan_map, float* dst);
int conv_wino7_eligible(int k, int stride, int Cin, int Cout, int Ho, int Wo, int B, int in_cs, int out_cs, long* items, int* ksplit);
int conv_wino7_launch(const ConvParams& p, hipStream_t s);
// conv_pw2.hip: two 1x1 convolutions as one launch (the head pairs of both trunks): out = act2((act1(x W1 + b1)) W2 + b2); x [npix, >= 128 ch],
// W1 / W2 in conv_mfma.hip's packed order (pack_conv, k = 1), hidden width H a multiple of 128, 32 padded output channels
struct Pw2Params {
    const float* in; int in_cs; long npix;
    const float* w1; const float* b1; int H; int act1;
    const float* w2; const float* b2; int act2;
    float* out; int out_cs; int cout_store;
};
int conv_pw2_eligible(int cin_pad, int hidden_pad, int cout2_pad, long npix, int in_cs, int out_cs);
int conv_pw2_launch(const Pw2Params& p, hipStream_t s);
// > 0: conv_wino4_launch would share the last round of this layer out as tail pieces (given the scratch); channel steps per workgroup
int conv_wino4_tail_plan(int Cin, int Cout, int Ho, int Wo, int B, int* tail_items);
int conv_wino_launch(const ConvParams& p, int pool, hipStream_t s);

// debug cross-check (one thread per output element, obviously-correct loops)
void conv_naive_launch(const float* x, int B, int H, int W, int Cin, int in_cs, const float* w_hwio_like,
                       const float* bias, int k, int stride, int Cout, int act, float* out, int out_cs,
                       int Ho, int Wo, int pad_t, int pad_l, hipStream_t s);

void maxpool2_launch(const float* x, int B, int H, int W, int C, int in_cs, float* out, hipStream_t s);
void avgpool8_launch(const float* x, int B, int H, int W, int C, float* out, int out_cs, hipStream_t s);
void resize_bilinear_launch(const float* x, int B, int H, int W, int C, int in_cs,
                            int oh, int ow, float* out, hipStream_t s);
void preprocess_u8_launch(const unsigned char* img, int B, int H, int W, int oh, int ow, float* out, hipStream_t s);
void crop_and_resize_launch(const float* img, int B, int H, int W, int C, const float* center,
                            const float* scale, int crop, float* out, hipStream_t s);

struct MaskBuffers {           // per-call scratch owned by the executor
    unsigned long long* argmax_key;  // [B]
    unsigned char* det;              // [B,H,W]
    float* fg;                       // [B,H,W] (optional output for tests; may be null)
};
// seg logits (small map, channel stride cs) -> hand_scoremap [B,H,W,2], det bytes, arg-max keys
void seg_upsample_softmax_launch(const float* small, int B, int hs, int ws, int cs, int H, int W,
                                 float* scoremap_large, const MaskBuffers& mb, hipStream_t s);
// same from an already-large scoremap [B,H,W,2] (hp3d_mask_from_scoremap)
void seg_softmax_launch(const float* scoremap_large, int B, int H, int W, const MaskBuffers& mb, hipStream_t s);
// geodesic growth + bbox + centre / crop size / scale.  mask_out (float [B,H,W]) may be null.
void mask_grow_launch(const MaskBuffers& mb, int B, int H, int W, int empty_fltmax, float* mask_out,
                      float* center, float* crop_size, float* scale, int* seed, hipStream_t s);

size_t fc_scratch_floats(int B, int Cin, int Cout);   // split-K partials [ceil(Cin/128)][B][Cout]
// x2 / F1: input columns F1 .. Cin-1 come from x2 (row stride Cin - F1): concat([flatten, hand_side]) without a copy; nullptr = plain
void fc_launch(const float* x, int B, int Cin, int x_stride, const float* w, const float* bias, int Cout,
               int act, float* out, int out_stride, float* scratch, hipStream_t s, const float* x2 = nullptr, int F1 = 0);
// round 6: the K slices of an FC layer only (the consumer reduces them); the tail of a lifting tower -- reduce of layer 0 + two small FC layers -- as one launch;
// a 3x3 / stride-2 layer on an 8x8 map as split-K GEMM over its output pixels (w = the HWIO filter)
int fc_partial_launch(const float* x, int B, int Cin, int x_stride, const float* w, int Cout, float* scratch, hipStream_t s, const float* x2 = nullptr, int F1 = 0);
int fc_tail_eligible(int C0, int C1, int C2);
void fc_tail_launch(const float* part0, int ns0, int B, int C0, const float* bias0, int act0, const float* w1, const float* b1, int C1, int act1,
                    const float* w2, const float* b2, int C2, int act2, float* out, int out_stride, hipStream_t s);
void conv_s2_gemm_launch(const float* x, int n, int S, int C, const float* w_hwio, const float* bias, int Cout, int act, float* out, float* scratch, hipStream_t s);
// u = (ux,uy,uz) [B,3], coord_can [B,63], hand_side [B,2] -> rot [B,9], coord_rel [B,63]
void lift_epilogue_launch(const float* u, const float* coord_can, const float* hand_side, int B,
                          float* rot, float* coord_rel, int do_flip_rot, hipStream_t s);
void bone_rel_inv_launch(const float* rel, int B, float* xyz, hipStream_t s);   // [B,21,3] local -> xyz
// first arg-max of the up-sampled (oh x ow, legacy bilinear) score map per channel + trafo_coords; h*w <= 4096
void kp_detect_launch(const float* sm, int B, int h, int w, int C, int cs, int oh, int ow, const float* scale,
                      const float* center, int* kp_crop, double* kp_image, hipStream_t s);
void argmax2d_launch(const float* x, int B, int H, int W, int C, int cs, int* out_rc, hipStream_t s);
void copy_channels_launch(const float* in, int npix, int C, int in_cs, float* out, int out_cs, hipStream_t s);
size_t mask_grow_lds_bytes(int H, int W);
void touch_launch(const float* p, size_t nfloats, float* sink, hipStream_t s);
void cvt_channels_f16_launch(const float* in, int npix, int C, int in_cs, hp3d_f16* out, int out_cs, hipStream_t s);
void pad_channels_launch(const float* in, int npix, int C, float* out, int out_cs, hipStream_t s);

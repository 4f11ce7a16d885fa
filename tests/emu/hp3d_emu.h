// hp3d_emu.h -- CPU interpreter shim for the HIP kernel sources (TEST INFRASTRUCTURE ONLY).
//
// Compiling hand3d_amd/csrc/*.hip with `g++ -x c++ -DHP3D_EMU` against this header yields
// libhp3d_emu.so: the same kernels and the same executor, but every GPU thread is a fiber on one
// host thread, __syncthreads()/wave collectives are cooperative rendezvous, and
// v_mfma_f32_32x32x2_f32 is evaluated from its documented lane->element maps as an fmaf chain.
// It lets `pytest -m "not gpu"` check tile maps, weight packing, halo/zero-fill, masks and the
// executor's wiring on small shapes without a GPU.  It is never shipped or loaded by the product.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

typedef float f32x4 __attribute__((vector_size(16)));
typedef float f32x2 __attribute__((vector_size(8)));
typedef float f32x16 __attribute__((vector_size(64)));
typedef _Float16 hp3d_f16;
typedef _Float16 f16x8 __attribute__((vector_size(16)));
typedef _Float16 f16x4 __attribute__((vector_size(8)));
typedef unsigned u32x4 __attribute__((vector_size(16)));
typedef unsigned u32x2 __attribute__((vector_size(8)));
// two float32 -> two bfloat16, round to nearest even (v_cvt_pk_bf16_f32); `lo` in bits 0..15
static inline unsigned hp3d_emu_bf16_rne(float f) {
    unsigned u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;      // NaN stays NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
static inline unsigned hp3d_cvt_pk_bf16(float lo, float hi) { return hp3d_emu_bf16_rne(lo) | (hp3d_emu_bf16_rne(hi) << 16); }
// wave-level LDS rendezvous: lanes are fibers here and need a real yield point (use only where the trip count is uniform
// over the workgroup)
#define HP3D_WAVE_LDS_SYNC() __syncthreads()

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct EmuIdx { unsigned x, y, z; };
extern EmuIdx hp3d_emu_threadIdx, hp3d_emu_blockIdx, hp3d_emu_blockDim, hp3d_emu_gridDim;
#define threadIdx hp3d_emu_threadIdx
#define blockIdx hp3d_emu_blockIdx
#define blockDim hp3d_emu_blockDim
#define gridDim hp3d_emu_gridDim

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define HP3D_KERNEL(nthr)
#define HP3D_KERNEL2(nthr, w)
#define HP3D_SCHED_BARRIER() ((void)0)
#define HP3D_OPAQUE_V(x) ((void)0)
static inline bool hp3d_first_use_on_device(bool (&done)[64]) { const bool f = !done[0]; done[0] = true; return f; }
static inline int hp3d_num_cus() { return 3; }     // small on purpose: persistent kernels walk several items per workgroup
#define HP3D_SG_VALU 0x2
#define HP3D_SG_MFMA 0x8
#define HP3D_SG_VMEM_READ 0x20
#define HP3D_SG_DS_READ 0x100
#define HP3D_SCHED_GROUP(kind, n) ((void)0)
#define HP3D_READFIRSTLANE(x) (x)
#define HP3D_OPAQUE_SGPR(x) (x)
#define HP3D_SADD(s, inc) ((s) += (inc))
#define HP3D_WAIT_VMCNT0() ((void)0)
// workgroups run one after another on the interpreter: the hand-off protocol reduces to its arithmetic
#define HP3D_ACQUIRE_AGENT() ((void)0)
#define HP3D_TICKET_AGENT(ptr) ((*(ptr))++)
#define HP3D_STORE_RELAXED_AGENT(ptr, v) (*(ptr) = (v))
struct hp3d_rsrc_t { const char* base; unsigned bytes; };
#define HP3D_MAKE_RSRC(ptr, bytes) hp3d_rsrc_t{(const char*)(ptr), (unsigned)(bytes)}
static inline void hp3d_emu_buffer_lds16(hp3d_rsrc_t r, float* lds_wave_base, unsigned off, int lane) {
    if (off < r.bytes && off + 16u <= r.bytes) memcpy(lds_wave_base + lane * 4, r.base + off, 16);      // (off + 16 may wrap: negative offsets)
    else memset(lds_wave_base + lane * 4, 0, 16);
}
#define HP3D_BUFFER_LDS16(rsrc, lds_wave_base, voff, soff, lane) \
    hp3d_emu_buffer_lds16((rsrc), (float*)(lds_wave_base), (unsigned)(voff) + (unsigned)(soff), (lane))
// (hardware: the raw-buffer range check covers the VECTOR offset only -- a scalar offset that leaves the buffer reads whatever lies there.
//  The interpreter does the same (with voff + soff checked together no test could see conv_wino7's ring running past its filters);
//  one concession to a host process: a read that would leave the buffer through soff returns zeros instead of touching foreign memory, and is
//  COUNTED -- hp3d_emu_soff_overreads, asserted 0 by tests/test_emu_kernels.py for buffers without slack)
extern unsigned long hp3d_emu_soff_overreads;
static inline f32x4 hp3d_emu_buffer_load16(hp3d_rsrc_t r, unsigned voff, unsigned soff) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (voff < r.bytes && voff + 16u <= r.bytes) {
        if ((unsigned long)voff + soff + 16u <= r.bytes) memcpy(&v, r.base + voff + soff, 16);
        else ++hp3d_emu_soff_overreads;
    }
    return v;
}
#define HP3D_BUFFER_LOAD16(rsrc, voff, soff) hp3d_emu_buffer_load16((rsrc), (unsigned)(voff), (unsigned)(soff))
#define HP3D_BUFFER_LOAD16_NT(rsrc, voff, soff) hp3d_emu_buffer_load16((rsrc), (unsigned)(voff), (unsigned)(soff))
#define HP3D_BUFFER_LOAD16_SC1(rsrc, voff, soff) hp3d_emu_buffer_load16((rsrc), (unsigned)(voff), (unsigned)(soff))
static inline float hp3d_emu_buffer_load4(hp3d_rsrc_t r, unsigned voff, unsigned soff) {
    float v = 0.f;
    if (voff < r.bytes && voff + soff + 4u <= r.bytes) memcpy(&v, r.base + voff + soff, 4);
    return v;
}
#define HP3D_BUFFER_LOAD4(rsrc, voff, soff) hp3d_emu_buffer_load4((rsrc), (unsigned)(voff), (unsigned)(soff))
static inline f32x2 hp3d_emu_buffer_load8(hp3d_rsrc_t r, unsigned voff, unsigned soff) {
    f32x2 v = {0.f, 0.f};
    if (voff < r.bytes && voff + soff + 8u <= r.bytes) memcpy(&v, r.base + voff + soff, 8);
    return v;
}
#define HP3D_BUFFER_LOAD8(rsrc, voff, soff) hp3d_emu_buffer_load8((rsrc), (unsigned)(voff), (unsigned)(soff))
static inline void hp3d_emu_buffer_store4(hp3d_rsrc_t r, float v, unsigned voff, unsigned soff) {
    if (voff < r.bytes && voff + soff + 4u <= r.bytes) memcpy((char*)r.base + voff + soff, &v, 4);   // hardware: range check on voff
}
#define HP3D_BUFFER_STORE4(rsrc, val, voff, soff) hp3d_emu_buffer_store4((rsrc), (val), (unsigned)(voff), (unsigned)(soff))
#define HP3D_BUFFER_STORE4_NT(rsrc, val, voff, soff) hp3d_emu_buffer_store4((rsrc), (val), (unsigned)(voff), (unsigned)(soff))
#define HP3D_BUFFER_STORE4_SC1(rsrc, val, voff, soff) hp3d_emu_buffer_store4((rsrc), (val), (unsigned)(voff), (unsigned)(soff))
static inline void hp3d_emu_buffer_store2(hp3d_rsrc_t r, hp3d_f16 v, unsigned voff, unsigned soff) {
    if (voff < r.bytes && voff + soff + 2u <= r.bytes) memcpy((char*)r.base + voff + soff, &v, 2);
}
#define HP3D_BUFFER_STORE2(rsrc, half_val, voff, soff) hp3d_emu_buffer_store2((rsrc), (hp3d_f16)(half_val), (unsigned)(voff), (unsigned)(soff))
static inline void hp3d_emu_buffer_store16(hp3d_rsrc_t r, f32x4 v, unsigned voff, unsigned soff) {
    if (voff < r.bytes && voff + soff + 16u <= r.bytes) memcpy((char*)r.base + voff + soff, &v, 16);
}
#define HP3D_BUFFER_STORE16(rsrc, val4, voff, soff) hp3d_emu_buffer_store16((rsrc), (val4), (unsigned)(voff), (unsigned)(soff))
#define HP3D_BUFFER_STORE16_SC1(rsrc, val4, voff, soff) hp3d_emu_buffer_store16((rsrc), (val4), (unsigned)(voff), (unsigned)(soff))
#define HP3D_GLDS16(gptr, lds_wave_base, lane) memcpy((float*)(lds_wave_base) + (lane) * 4, (gptr), 16)
extern float* hp3d_emu_smem;
#define HP3D_DYN_SMEM(name) float* name = hp3d_emu_smem

typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

void hp3d_emu_run(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
#define HP3D_LAUNCH(kern, grid, block, shmem, stream, ...) \
    hp3d_emu_run(grid, block, shmem, [=]() { kern(__VA_ARGS__); })

void hp3d_emu_syncthreads();
void hp3d_emu_yield();
// LDS-counter hand-off inside a workgroup (conv_wino4.hip): one lane of a wave adds 1; a waiter lets the other fibers run until the count is reached
void hp3d_emu_wave_sync();
#define __syncthreads hp3d_emu_syncthreads
f32x16 hp3d_emu_mfma_32x32x2(float a, float b, f32x16 c);
#define HP3D_MFMA_32x32x2(a, b, c) hp3d_emu_mfma_32x32x2((a), (b), (c))
f32x4 hp3d_emu_mfma_16x16x4(float a, float b, f32x4 c);
#define HP3D_MFMA_16x16x4(a, b, c) hp3d_emu_mfma_16x16x4((a), (b), (c))
f32x16 hp3d_emu_mfma_32x32x16_f16(f32x4 a, f32x4 b, f32x16 c);
#define HP3D_MFMA_32x32x16_F16(a, b, c) hp3d_emu_mfma_32x32x16_f16((a), (b), (c))
#define HP3D_MFMA_32x32x16_F16_ACC(REG_A, acc, a, b) ((acc) = hp3d_emu_mfma_32x32x16_f16((a), (b), (acc)))
#define HP3D_MFMA_32x32x16_F16_ACC_FIRST(REG_A, acc, a, b)                                       \
    do {                                                                                         \
        const f32x16 _z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; \
        (acc) = hp3d_emu_mfma_32x32x16_f16((a), (b), _z);                                        \
    } while (0)
static inline float hp3d_vmax(float a, float b) { return fmaxf(a, b); }
#define HP3D_MFMA_RESULT_FENCE2(r0, r1) ((void)0)
#define HP3D_MFMA_RESULT_FENCE4(r0, r1, r2, r3) ((void)0)
#define HP3D_MFMA_32x32x16_F16_V(acc, a, b) HP3D_MFMA_32x32x16_F16_ACC("v", acc, a, b)
#define HP3D_MFMA_32x32x16_F16_V_FIRST(acc, a, b) HP3D_MFMA_32x32x16_F16_ACC_FIRST("v", acc, a, b)
// skip must be uniform over the wave (every fiber takes the same branch, so the collective MFMA stays collective)
#define HP3D_MFMA4_UNLESS(acc, a4, b4, skip)                                                        \
    do {                                                                                           \
        if (!(skip))                                                                               \
            for (int _e = 0; _e < 4; ++_e) (acc) = hp3d_emu_mfma_32x32x2((a4)[_e], (b4)[_e], (acc)); \
    } while (0)
#define HP3D_MFMA16_2x4_UNLESS(acc0, acc1, a0, a1, b4, skip)                      \
    do {                                                                          \
        if (!(skip))                                                              \
            for (int _e = 0; _e < 4; ++_e) {                                      \
                (acc0) = hp3d_emu_mfma_16x16x4((a0)[_e], (b4)[_e], (acc0));       \
                (acc1) = hp3d_emu_mfma_16x16x4((a1)[_e], (b4)[_e], (acc1));       \
            }                                                                     \
    } while (0)
#define HP3D_MFMA16_PLANE_UNLESS(REG, acc0, acc1, a0, a1, b4, skip) HP3D_MFMA16_2x4_UNLESS(acc0, acc1, a0, a1, b4, skip)
#define HP3D_MFMA16_PAIR_UNLESS(REG, acc0, acc1, a0e, a1e, be, skip)                  \
    do {                                                                              \
        if (!(skip)) {                                                                \
            (acc0) = hp3d_emu_mfma_16x16x4((a0e), (be), (acc0));                      \
            (acc1) = hp3d_emu_mfma_16x16x4((a1e), (be), (acc1));                      \
        }                                                                             \
    } while (0)
#define HP3D_MFMA16_PAIR(REG, acc0, acc1, a0e, a1e, be) HP3D_MFMA16_PAIR_UNLESS(REG, acc0, acc1, a0e, a1e, be, 0)
#define HP3D_MFMA16_PAIR_FIRST(REG, acc0, acc1, a0e, a1e, be)                         \
    do {                                                                              \
        const f32x4 _z = {0.f, 0.f, 0.f, 0.f};                                        \
        (acc0) = _z; (acc1) = _z;                                                     \
        HP3D_MFMA16_PAIR_UNLESS(REG, acc0, acc1, a0e, a1e, be, 0);                    \
    } while (0)
#define HP3D_MFMA16_X2(acc0, acc1, a0e, a1e, b0e, b1e)                                \
    do {                                                                              \
        (acc0) = hp3d_emu_mfma_16x16x4((a0e), (b0e), (acc0));                         \
        (acc1) = hp3d_emu_mfma_16x16x4((a1e), (b1e), (acc1));                         \
    } while (0)
#define HP3D_MFMA16_X1(acc, a4, b4)                                                   \
    do {                                                                              \
        for (int _e = 0; _e < 4; ++_e) (acc) = hp3d_emu_mfma_16x16x4((a4)[_e], (b4)[_e], (acc)); \
    } while (0)
#define HP3D_MFMA16_PLANE_FIRST(REG, acc0, acc1, a0, a1, b4)                      \
    do {                                                                          \
        const f32x4 _z = {0.f, 0.f, 0.f, 0.f};                                    \
        (acc0) = _z; (acc1) = _z;                                                 \
        HP3D_MFMA16_2x4_UNLESS(acc0, acc1, a0, a1, b4, 0);                        \
    } while (0)
#define HP3D_MFMA_OPERAND_FENCE() ((void)0)
f32x4 hp3d_emu_mfma_16x16x32_bf16(u32x4 a, u32x4 b, f32x4 c);
#define HP3D_MFMA16B_PAIR(REG, acc0, acc1, a0, a1, b)                                   \
    do {                                                                              \
        (acc0) = hp3d_emu_mfma_16x16x32_bf16((a0), (u32x4)(b), (acc0));               \
        (acc1) = hp3d_emu_mfma_16x16x32_bf16((a1), (u32x4)(b), (acc1));               \
    } while (0)
#define HP3D_MFMA16B_PAIR_FIRST(REG, acc0, acc1, a0, a1, b)                             \
    do {                                                                              \
        const f32x4 _z = {0.f, 0.f, 0.f, 0.f};                                        \
        (acc0) = _z; (acc1) = _z;                                                     \
        HP3D_MFMA16B_PAIR(REG, acc0, acc1, a0, a1, b);                                \
    } while (0)
unsigned long long hp3d_emu_shfl_xor_u64(unsigned long long v, int mask);
inline unsigned long long __shfl_xor(unsigned long long v, int mask) { return hp3d_emu_shfl_xor_u64(v, mask); }

template <typename T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
using std::max;
using std::min;

// ---- minimal HIP runtime (host memory stands in for device memory) ----
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipFuncSetAttribute(const void* f, int attr, int v);

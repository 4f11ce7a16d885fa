// Micro-benchmark for NEXT round's question (profiles/r04_tuning_notes.md section 5): can one workgroup hand a 74 KB block (conv_wino4's
// transformed input V of one step) to workgroups on OTHER CUs of the same XCD once per step (~5.7 us), inside one launch?
//   hipcc --offload-arch=gfx950 -O3 xcd_handoff.hip -o xcd_handoff && ./xcd_handoff
// 256 persistent workgroups (one per CU: 100 KB of LDS each).  Workgroup ids are dealt round-robin over the 8 XCDs, so ids w and w + 8 share
// an XCD (checked: every workgroup records its XCC_ID).  Group g = 4 consecutive same-XCD workgroups: member 0 PRODUCES a block per step into
// a ring of 4 slots (agent-scope write-through stores, then a release of the step counter), members 1..3 CONSUME it (poll the counter with
// agent-scope loads, read the block with agent-scope loads, check every word, post their own progress for the producer's back-pressure).
// Every wait is bounded: a timeout sets an error flag and leaves (a workgroup that is not resident must not hang the GPU).
// Reported: us per step for the consumers against a no-hand-off baseline (everybody reads a constant block), and the check result.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int BLOCK_BYTES = 73728;                  // 36 planes x 32 tiles x 16 channels x 4 B
constexpr int PIECES = BLOCK_BYTES / (256 * 16);    // 18 x 16 B per thread
constexpr int RING = 4;
extern __shared__ float lds[];

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODE>   // 0: baseline (no hand-off: consumers read slot 0 of their group, written before the launch), 1: hand-off per step
__global__ __launch_bounds__(256, 1) void handoff(float* ring, unsigned* prod_step, unsigned* cons_step, unsigned* xcc, unsigned* err,
                                                  unsigned long long* mismatches, int steps) {
    const int w = blockIdx.x, tid = threadIdx.x;
    const int xcd = w & 7, j = w >> 3;              // j-th workgroup of this XCD
    const int grp = (j >> 2) * 8 + xcd, member = j & 3;
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[w] = id & 0xf;
    }
    float* gring = ring + (size_t)grp * RING * (BLOCK_BYTES / 4);
    rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)gring, 0, RING * BLOCK_BYTES, 0x00020000);
    unsigned* ps = prod_step + grp * 32;            // (own cache line per group)
    unsigned* cs = cons_step + (grp * 4) * 32;
    float acc = 0.f;
    unsigned long long bad = 0;
    for (int s = 1; s <= steps; ++s) {
        const int slot = MODE ? (s % RING) : 0;
        if (MODE == 1 && member == 0) {
            // back-pressure: slot s % RING was read by every consumer in step s - RING
            if (tid == 0 && s > RING) {
                for (int m = 1; m < 4; ++m) {
                    unsigned spins = 0;
                    while (ld_agent(cs + m * 32) + RING < (unsigned)s) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1u << 18)) { atomicExch(err, 1u); break; }
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const float v = (float)(s * 1000 + i);
                f32x4 q = {v, v + 0.25f, (float)tid, (float)grp};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, q), rs,
                                                       (i * 256 + tid) * 16, slot * BLOCK_BYTES, 16 /* sc1: write-through at agent scope */);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(ps, (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 0 || member != 0) {
            if (MODE == 1) {
                if (tid == 0) {
                    unsigned spins = 0;
                    while (ld_agent(ps) < (unsigned)s) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1u << 18)) { atomicExch(err, 2u); break; }
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (i * 256 + tid) * 16, slot * BLOCK_BYTES, MODE ? 16 : 0));
                acc += q[0] + q[1];
                if (MODE == 1) {
                    const float v = (float)(s * 1000 + i);
                    bad += (q[0] != v) + (q[1] != v + 0.25f) + (q[2] != (float)tid) + (q[3] != (float)grp);
                }
            }
            if (MODE == 1) {
                __syncthreads();
                if (tid == 0) __hip_atomic_store(cs + member * 32, (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    lds[tid] = acc;
    if (bad) atomicAdd(mismatches, bad);
    if (acc == 12345.678f) ring[0] = acc;
}

int main() {
    const int steps = 400, grid = 256, groups = 64;
    float* ring; unsigned *ps, *cs, *xcc, *err; unsigned long long* mm;
    hipMalloc(&ring, (size_t)groups * RING * BLOCK_BYTES);
    hipMalloc(&ps, groups * 32 * 4); hipMalloc(&cs, groups * 4 * 32 * 4); hipMalloc(&xcc, grid * 4); hipMalloc(&err, 4); hipMalloc(&mm, 8);
    hipMemset(ring, 0, (size_t)groups * RING * BLOCK_BYTES);
    for (int mode = 0; mode < 2; ++mode) {
        auto k = mode ? handoff<1> : handoff<0>;
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(ps, 0, groups * 32 * 4); hipMemset(cs, 0, groups * 4 * 32 * 4); hipMemset(err, 0, 4); hipMemset(mm, 0, 8);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 100 * 1024, 0, ring, ps, cs, xcc, err, mm, steps);
            hipEventRecord(e1);
            if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\n"); return 2; }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        unsigned herr = 0; unsigned long long hmm = 0; std::vector<unsigned> hx(grid);
        hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost); hipMemcpy(&hmm, mm, 8, hipMemcpyDeviceToHost); hipMemcpy(hx.data(), xcc, grid * 4, hipMemcpyDeviceToHost);
        int same = 0;
        for (int w = 0; w + 8 < grid; ++w) same += hx[w] == hx[w + 8];
        printf("%-44s %.3f us / step (%d steps); timeouts %u, wrong words %llu; workgroup pairs (w, w + 8) on one XCD: %d of %d\n",
               mode ? "hand-off per step (1 producer -> 3 consumers)" : "baseline: every workgroup reads a constant block", best * 1e3 / steps, steps, herr, hmm, same, grid - 8);
    }
    return 0;
}

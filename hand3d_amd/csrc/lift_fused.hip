// lift_fused.hip -- the whole 3-D lifting stage as ONE launch for small batches (round 3).
//
// PosePrior (_inference_pose3d_can, nets/ColorHandPose3DNetwork.py:249-272) and ViewpointNet (_rotation_estimation, :285-309): two
// independent towers of 3 x {3x3 s1, 3x3 s2} convolutions on the 32 x 32 x 21 score map, then three fully connected layers each.  At
// batch 1 that is 0.2 GFLOP in 18 dependent layers: as 24 launches of the general kernels it costs 0.33 ms of a 1.7 ms pipeline
// (every launch ~13 us whatever it computes; profiles/r03_tuning_notes.md).  Here both towers walk their layers inside one
// persistent launch, phase by phase, separated by grid barriers (monotonic counter, agent-scope release / acquire per workgroup:
// cdna_hip_programming.md Guideline 16 and the MI355X_MICROARCH.md price list, "barrier-counter"):
//   phases 0..5  conv i of both towers: work item = (tower, image, 2x2 output pixels, 64 couts); the workgroup's sixteen waves split
//                the input channels, lane = cout, f32 FMAs against the input patch staged in LDS, fixed-order LDS reduction,
//                bias + leaky-ReLU, NHWC store.  Weights packed [cout block of 64][cin][tap][64]: one contiguous stream per wave;
//   phase  6     fc_rel0 / fc_vp0 over K slices of 256 rows -> raw partial sums [slice][image][cout];
//   phase  7     fc_rel1 / fc_vp1: the input is finished on the fly (sum of the slices in slice order + bias + leaky-ReLU);
//   phase  8     fc_xyz (or fc_bottleneck -> fc_xyz) / fc_vp_ux,uy,uz: one workgroup per (tower, image), final outputs.
// All sums have a fixed order: results are deterministic; they agree with the layer-by-layer path to float32 rounding (the
// accumulation order differs), well inside the path's 1e-4 bar (tests/test_gpu_parity.py::test_lift_fused_*).
// What the stage costs at batch 1 is DEPTH, not arithmetic: a dependent memory round trip is 2-5 us on the lightly loaded chip
// (measured per phase, profiles/r03_tuning_notes.md), so a work item gets a 1024-thread workgroup whose 16 waves split K and
// keep all their weight loads in flight at once (36 per lane and step), the weights are packed so that a wave streams one
// contiguous run (read in HWIO order the nine taps lie a power of two apart and the loads queue on one HBM channel: 3x slower),
// and patch / partial-sum loads are issued together before anything waits.  9 phases: 0.33 ms as launches -> 0.13 ms.
// The grid is small on purpose (one workgroup per two CUs, 114 registers, 42 KB of LDS): every workgroup is resident whatever
// else runs, which a hand-rolled grid barrier needs; the engine uses this kernel only for small batches (B <= 4), where the stage
// is latency-bound; larger batches keep the per-layer kernels.
#include "hp3d_common.h"
#include "lift_fused.h"
#include <cstdlib>
#include <cstring>

namespace {

constexpr int LF_THREADS = 1024;                 // 16 waves: up to 16 K slices per work item -- depth, not arithmetic, is the cost
constexpr int LF_WAVES = LF_THREADS / 64;
constexpr int LF_PATCH_FLOATS = 25 * 256;      // input patch of a 2x2 output block: up to 5x5 pixels x 256 channels
constexpr int LF_RED_FLOATS = LF_WAVES * 4 * 64;   // [K slice (wave)][pixel][cout]
constexpr int LF_SMEM_BYTES = (LF_PATCH_FLOATS + LF_RED_FLOATS) * 4;

#ifndef HP3D_EMU
// grid barrier: one monotonic counter; every workgroup drains its stores, ONE lane releases at agent scope, arrives, polls relaxed
// (agent-scope loads) until everybody of this round is there, acquires at agent scope; then the workgroup goes on with plain loads.
__device__ __forceinline__ void lf_grid_barrier(unsigned* counter, unsigned target, unsigned* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every wave: its stores of this phase have left
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 24)) {                  // a workgroup that never arrives must not hang the GPU: the outputs of this
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // launch are garbage, and the host is told so
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
#endif

struct ConvDesc { int H, W, Cin, Cout, stride; };       // input extent, channels (Cin as stored: multiple of 4), stride

// one conv work item: image b, output block (oy0, ox0) of 2x2 pixels, couts co0 .. co0+63
__device__ __forceinline__ void lf_conv_item(const float* in, int in_cs, const float* w, const float* bias, float* out, int out_cs,
                                              const ConvDesc d, int b, int oy0, int ox0, int co0, float* patch, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int Ho = (d.H + d.stride - 1) / d.stride, Wo = (d.W + d.stride - 1) / d.stride;
    // TF SAME padding: total = max((out-1)*s + 3 - in, 0), before = total / 2 (stride 2 on even sizes: 0 before, 1 after)
    const int pt = max((Ho - 1) * d.stride + 3 - d.H, 0) / 2, pl = max((Wo - 1) * d.stride + 3 - d.W, 0) / 2;
    const int PS = d.stride + 3;                      // patch side: 4 (stride 1) or 5 (stride 2)
    const int iy0 = oy0 * d.stride - pt, ix0 = ox0 * d.stride - pl;
    const int c4 = d.Cin >> 2;
    __syncthreads();                                  // the previous item is done with patch / red
    {   // all of this thread's patch loads first, then the LDS writes: one memory
        // round trip per item instead of one per element (the activations were written by other CUs a phase ago)
        f32x4 v[2];
        const int n = PS * PS * c4;          // <= 25 pixels x 64 quads = 1600 <= 2 per thread
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * LF_THREADS;
            const int q = e % c4, pp = e / c4, py = pp / PS, px = pp - py * PS;
            const int y = iy0 + py, x = ix0 + px;
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (e < n && (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W)
                v[i] = *(const f32x4*)(in + ((size_t)(b * d.H + y) * d.W + x) * in_cs + q * 4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * LF_THREADS;
            if (e < n) *(f32x4*)(patch + (e / c4) * d.Cin + (e % c4) * 4) = v[i];
        }
    }
    __syncthreads();
    const int co = co0 + lane;
    const bool cok = co < d.Cout;          // (the packed weights are zero-padded to whole blocks of 64 couts)
    // K slices: one per wave, whole channel quads, as many as the layer has (Cin = 32: 8 slices of 4 channels ... 256: 16 of 16)
    const int nk = min(LF_WAVES, d.Cin >> 2);
    const int cper = d.Cin / nk, c_lo = wv * cper;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (cok && wv < nk) {
        // The layers are latency-bound (weights come from HBM / the Infinity Cache once per launch, one dependent chain per wave):
        // all nine taps of four channels are loaded together -- 36 (with the unroll 72) independent loads in flight per lane.
        // (weights packed [cout block][cin][tap][64]: a wave's whole stream is ONE contiguous run of 256-byte rows -- read in
        //  HWIO order the nine taps lie a power of two apart and the 36 loads queue up on one HBM channel: measured 3x slower)
        const float* w0 = w + ((size_t)(co0 >> 6) * d.Cin + c_lo) * (9 * 64) + lane;
        const int row = d.stride * PS * d.Cin, col = d.stride * d.Cin;
        for (int c = 0; c < cper; c += 4) {
            float wg[9][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 9; ++t) wg[t][j] = w0[((c + j) * 9 + t) * 64];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float* p00 = patch + ((t / 3) * PS + (t % 3)) * d.Cin + c_lo + c;
                const f32x4 x00 = *(const f32x4*)p00, x01 = *(const f32x4*)(p00 + col), x10 = *(const f32x4*)(p00 + row), x11 = *(const f32x4*)(p00 + row + col);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0] = fmaf(x00[j], wg[t][j], acc[0]);
                    acc[1] = fmaf(x01[j], wg[t][j], acc[1]);
                    acc[2] = fmaf(x10[j], wg[t][j], acc[2]);
                    acc[3] = fmaf(x11[j], wg[t][j], acc[3]);
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) red[(wv * 4 + p) * 64 + lane] = acc[p];
    __syncthreads();
    if (wv < 4) {   // wave p finishes output pixel p: slices summed in wave order, bias, leaky-ReLU
        const int p = wv, oy = oy0 + (p >> 1), ox = ox0 + (p & 1);
        if (cok && oy < Ho && ox < Wo) {
            float s = 0.f;
            for (int k = 0; k < nk; ++k) s += red[(k * 4 + p) * 64 + lane];
            s += bias[co];
            s = fmaxf(s, HP3D_LEAKY_SLOPE * s);
            out[((size_t)(b * Ho + oy) * Wo + ox) * out_cs + co] = s;
        }
    }
}

// x[k] of one image for an FC item, staged in LDS: k in [k0, k0 + n): mode 0 = feature map (+ hand side behind it), mode 1 = the
// previous FC layer finished on the fly from its K-slice partial sums
__device__ __forceinline__ void lf_stage_x(float* xs, int k0, int n, int mode, const float* feat, int nfeat, const float* hs2,
                                           const float* part, int nslices, int pstride, const float* pbias) {
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += LF_THREADS) {       // n <= 512: at most two elements per thread
        const int k = k0 + i;
        float v;
        if (mode == 0) {
            v = k < nfeat ? feat[k] : hs2[k - nfeat];
        } else {
            float ps[17];                                      // all slices of this element in flight, summed in slice order
#pragma unroll
            for (int s = 0; s < 17; ++s) ps[s] = s < nslices ? part[(size_t)s * pstride + k] : 0.f;
            v = 0.f;
#pragma unroll
            for (int s = 0; s < 17; ++s) v += ps[s];
            v += pbias[k];
            v = fmaxf(v, HP3D_LEAKY_SLOPE * v);
        }
        xs[i] = v;
    }
    __syncthreads();
}

// out[co] (+)= sum_{k in slice} xs[k - k0] * w[k][co] for co0 .. co0+63; the waves split the slice, LDS reduce in wave order;
// returns the sum in every wave's lanes (valid where co < Cout)
__device__ __forceinline__ float lf_fc_dot(const float* xs, int n, const float* w, int k0, int Cout, int co0, float* red) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int co = co0 + lane;
    const int per = (n + LF_WAVES - 1) / LF_WAVES, a = min(wv * per, n), e = min(a + per, n);
    float acc = 0.f;
    if (co < Cout) {
        int i = a;
        for (; i + 16 <= e; i += 16) {                // 16 weight loads in flight (a 256-row slice is 16 rows per wave)
            float wg[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) wg[j] = w[(size_t)(k0 + i + j) * Cout + co];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = fmaf(xs[i + j], wg[j], acc);
        }
        for (; i < e; ++i) acc = fmaf(xs[i], w[(size_t)(k0 + i) * Cout + co], acc);
    }
    red[wv * 64 + lane] = acc;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LF_WAVES; ++k) s += red[k * 64 + lane];
    return s;
}

HP3D_KERNEL(LF_THREADS)
void lift_fused_kernel(const LiftFusedParams p) {
    HP3D_DYN_SMEM(smem);
    float* patch = smem;
    float* red = smem + LF_PATCH_FLOATS;
    const int nwg = gridDim.x;
    unsigned bar_round = 0;
    (void)bar_round;
    for (int phase = p.phase_lo; phase <= p.phase_hi; ++phase) {
        if (phase < 6) {
            // ---- conv layer `phase` of both towers -------------------------------------------------------------------
            int items_t[2], pgx[2], pgy[2], cbs[2];
            ConvDesc d[2];
            for (int t = 0; t < 2; ++t) {
                const int li = t * 6 + phase;
                const int hin = 32 >> (phase >> 1);                                  // 32,32,16,16,8,8
                d[t].H = d[t].W = hin; d[t].Cin = p.cin[li]; d[t].Cout = p.cout[li]; d[t].stride = (phase & 1) ? 2 : 1;
                const int ho = (hin + d[t].stride - 1) / d[t].stride;
                pgy[t] = pgx[t] = (ho + 1) / 2;
                cbs[t] = (d[t].Cout + 63) / 64;
                items_t[t] = (p.towers & (1 << t)) ? p.B * pgy[t] * pgx[t] * cbs[t] : 0;
            }
            for (int it = blockIdx.x; it < items_t[0] + items_t[1]; it += nwg) {
                const int t = it < items_t[0] ? 0 : 1;
                int r = it - (t ? items_t[0] : 0);
                const int cb = r % cbs[t]; r /= cbs[t];
                const int gx = r % pgx[t]; r /= pgx[t];
                const int gy = r % pgy[t]; const int b = r / pgy[t];
                const int li = t * 6 + phase;
                const float* in = phase == 0 ? p.sm : p.act[t][(phase - 1) & 1];
                const int in_cs = phase == 0 ? p.sm_cs : p.cout[li - 1];
                lf_conv_item(in, in_cs, p.w[li], p.b[li], p.act[t][phase & 1], d[t].Cout, d[t], b, gy * 2, gx * 2, cb * 64, patch, red);
            }
        } else if (phase == 6 || phase == 7) {
            // ---- fc layer 0 / 1 of both towers: K slices of 256 rows -> partial sums [slice][B][Cout] ----------------------
            const int f = phase - 6;
            int items_t[2], nsl[2], cbs[2], K[2], Co[2];
            for (int t = 0; t < 2; ++t) {
                K[t] = p.fc_in[t * 3 + f]; Co[t] = p.fc_out[t * 3 + f];
                nsl[t] = (K[t] + 255) / 256; cbs[t] = (Co[t] + 63) / 64;
                items_t[t] = (p.towers & (1 << t)) ? p.B * nsl[t] * cbs[t] : 0;
            }
            for (int it = blockIdx.x; it < items_t[0] + items_t[1]; it += nwg) {
                const int t = it < items_t[0] ? 0 : 1;
                int r = it - (t ? items_t[0] : 0);
                const int cb = r % cbs[t]; r /= cbs[t];
                const int sl = r % nsl[t]; const int b = r / nsl[t];
                const int k0 = sl * 256, n = min(256, K[t] - k0);
                const int fi = t * 3 + f;
                if (f == 0) {
                    const int nfeat = K[t] - 2;                                      // 4x4xC features, then the two hand-side flags
                    lf_stage_x(patch, k0, n, 0, p.act[t][1] + (size_t)b * nfeat, nfeat, p.hs + b * 2, nullptr, 0, 0, nullptr);
                } else {
                    const int pK = p.fc_out[fi - 1];
                    lf_stage_x(patch, k0, n, 1, nullptr, 0, nullptr, p.fcp[t][0] + (size_t)b * pK, (p.fc_in[fi - 1] + 255) / 256, p.B * pK, p.fb[fi - 1]);
                }
                const float s = lf_fc_dot(patch, n, p.fw[fi], k0, Co[t], cb * 64, red);
                const int co = cb * 64 + (threadIdx.x & 63);
                if (threadIdx.x < 64 && co < Co[t]) p.fcp[t][f][((size_t)sl * p.B + b) * Co[t] + co] = s;
            }
        } else {
            // ---- last fc layer(s): one workgroup per (tower, image); PosePrior optionally through the 30-wide bottleneck -------
            const int items = p.B * 2;
            for (int it = blockIdx.x; it < items; it += nwg) {
                const int t = it & 1, b = it >> 1;
                if (!(p.towers & (1 << t))) continue;
                const int fi = t * 3 + 2, pK = p.fc_out[fi - 1], K = p.fc_in[fi];
                lf_stage_x(patch, 0, K, 1, nullptr, 0, nullptr, p.fcp[t][1] + (size_t)b * pK, (p.fc_in[fi - 1] + 255) / 256, p.B * pK, p.fb[fi - 1]);
                if (t == 0 && p.bn_w) {                       // fc_bottleneck [512,30] (linear), then fc_xyz [30,63] (nets/PosePriorNetwork.py:115-116)
                    const float s = lf_fc_dot(patch, K, p.bn_w, 0, 30, 0, red);
                    __syncthreads();
                    if (threadIdx.x < 30) patch[512 + threadIdx.x] = s + p.bn_b[threadIdx.x];
                    __syncthreads();
                    const float s2 = lf_fc_dot(patch + 512, 30, p.fw[fi], 0, 63, 0, red);
                    if (threadIdx.x < 63) p.out[0][b * 63 + threadIdx.x] = s2 + p.fb[fi][threadIdx.x];
                } else {
                    const int Co = p.fc_out[fi];
                    const float s = lf_fc_dot(patch, K, p.fw[fi], 0, Co, 0, red);
                    if (threadIdx.x < Co) p.out[t][b * Co + threadIdx.x] = s + p.fb[fi][threadIdx.x];
                }
            }
        }
#ifndef HP3D_EMU
        if (phase < p.phase_hi && !(p.towers & 256)) lf_grid_barrier(p.bar, ++bar_round * (unsigned)nwg, p.err);
#endif
    }
}

}  // namespace

int lift_fused_launch(const LiftFusedParams& pin, hipStream_t s) {
    static bool attr_done[64] = {};
    if (hp3d_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute((const void*)lift_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LF_SMEM_BYTES);
    LiftFusedParams p = pin;
#ifdef HP3D_EMU
    // the interpreter runs workgroups one after another: a grid barrier cannot work there, the phases become launches
    for (int ph = 0; ph <= 8; ++ph) {
        p.phase_lo = p.phase_hi = ph;
        HP3D_LAUNCH(lift_fused_kernel, dim3(4), dim3(LF_THREADS), LF_SMEM_BYTES, s, p);
    }
#else
    if (hipMemsetAsync(p.bar, 0, sizeof(unsigned), s) != hipSuccess) return -1;
    // The hand-rolled grid barrier needs every workgroup RESIDENT at once.  One workgroup per two CUs by design; clamped to what the
    // occupancy query says fits (16 waves, 42 KB of LDS: one or two per CU) minus the margin MI355X_MICROARCH.md asks for near the
    // API's edge.  A GPU shared with other work can still delay a workgroup: the barrier's bounded spin then raises p.err.
    static int resident[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (!resident[dev]) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)lift_fused_kernel, LF_THREADS, LF_SMEM_BYTES) != hipSuccess || per_cu < 1) per_cu = 1;
        resident[dev] = per_cu * hp3d_num_cus();
    }
    int nwg = hp3d_num_cus() / 2 > 0 ? hp3d_num_cus() / 2 : 1;
    p.phase_lo = 0; p.phase_hi = 8;
    const int cap = resident[dev] > hp3d_num_cus() ? resident[dev] - hp3d_num_cus() / 8 : resident[dev] * 7 / 8;
    if (nwg > cap) nwg = cap > 0 ? cap : 1;
    HP3D_LAUNCH(lift_fused_kernel, dim3(nwg), dim3(LF_THREADS), LF_SMEM_BYTES, s, p);
#endif
    return 0;
}

// conv_pw2.hip -- two 1x1 convolutions back to back as ONE launch: the head pairs of both trunks (round 5).
//
// Call sites (float32 mode): HandSegNet conv6_1 (128 -> 512, leaky-ReLU) + conv6_2 (512 -> 2, linear), nets/ColorHandPose3DNetwork.py:160-161;
// PoseNet2D conv5_1 (128 -> 512) + conv5_2 (512 -> 21), :202-203; conv6_6 / conv7_6 (128 -> 128) + conv6_7 / conv7_7 (128 -> 21), :213-214
// (NetworkOps.conv_relu / conv, utils/general.py:36-59).  As two launches of the general kernel the wide intermediate -- 512 channels at
// 40 x 40 x 32 images = 105 MB -- was written to HBM by a launch running at 63 TFLOP/s and read back by one running at 3 (a 2-cout GEMM):
// 0.142 ms for HandSegNet's pair, 0.092 + 2 x 0.037 ms for PoseNet2D's three.  Here a workgroup takes 64 pixels through both layers and
// the intermediate only ever exists as a 64 x 128 slab in LDS:
//
//   for each 128-channel slab of the hidden layer:   S = leaky(x W1[:, slab] + b1[slab])     64 x 128 x 128: 128 v_mfma_f32_32x32x2_f32 per wave
//                                                    Y += S W2[slab, :]                      64 x 32 x 128:   32 per wave
//   out = Y + b2 (+ leaky-ReLU if the second layer has one)
//
//   * x tile [64 px][128 ch] and S [64 px][128] in LDS, row pitch 132 floats: a `ds_read_b128` of row-indexed lanes touches 16 different bank
//     quads per lane group (33 quads per row == 1 mod 16), as conv_mfma.hip's patch pitch does;
//   * both weight matrices are read in conv_mfma.hip's packed fragment order [Cin/8][Cout/32][h][n][j] straight global -> VGPR (1 KB per
//     wave and 8-channel step, all workgroups read the same 256 KB: L2 / L1 hits), the K index permuted so that A and B are one 16-byte
//     access per four MFMAs;
//   * stage 1: wave w owns hidden channels 32 w .. 32 w + 31 of the slab for both pixel halves (2 x 16 accumulators); stage 2: wave w owns
//     pixel half w & 1 and the slab's channel half w >> 1 (16 accumulators, kept across slabs; the two channel halves meet once at the end);
//   * 68 KB of LDS, `__launch_bounds__(256, 2)`: two workgroups per CU cover each other's load / store phases.
// Arithmetic: float32 MFMA chains in channel order (the general kernel's order inside a layer; the hidden activation is never rounded
// differently: it stays float32).
#include "hp3d_common.h"
#include <cstdlib>
#include <cstring>

namespace {

constexpr int PW_PX = 64;                  // pixels per workgroup
constexpr int PW_C = 128;                  // input channels (all four call sites) = channels per hidden slab
constexpr int PW_PITCH = PW_C + 4;         // LDS row pitch in floats
constexpr int PW_SMEM_BYTES = 2 * PW_PX * PW_PITCH * 4;       // x tile + hidden slab: 67584 B
constexpr int PW_RING = 4;                 // stage-1 weight fragments in flight (8-channel steps)

HP3D_KERNEL2(256, 2)
void conv_pw2_kernel(const Pw2Params p) {
    HP3D_DYN_SMEM(smem);
    float* const Xs = smem;                              // [64][132]
    float* const Hs = smem + PW_PX * PW_PITCH;           // [64][132]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = HP3D_READFIRSTLANE(tid >> 6);
    const int n = lane & 31, kh = lane >> 5;             // MFMA column / row, k half
    const long pix0 = (long)blockIdx.x * PW_PX;

    // ---- x tile -> LDS (pixels beyond the tensor read as zeros) ---------------------------------------------------------------------------
    const hp3d_rsrc_t irsrc = HP3D_MAKE_RSRC(p.in, (unsigned)p.npix * (unsigned)p.in_cs * 4u);
#pragma unroll
    for (int pass = 0; pass < PW_PX / 8; ++pass) {
        const int px = pass * 8 + (tid >> 5), q = tid & 31;
        const long gp = pix0 + px;
        const int off = gp < p.npix ? (int)(gp * p.in_cs + q * 4) * 4 : (int)0x80000000;
        *(f32x4*)(Xs + px * PW_PITCH + q * 4) = HP3D_BUFFER_LOAD16(irsrc, off, 0);
    }
    const int CO32 = p.H >> 5;                           // 32-channel blocks of the hidden layer
    const hp3d_rsrc_t w1rsrc = HP3D_MAKE_RSRC(p.w1, (unsigned)(PW_C * p.H) * 4u);
    const hp3d_rsrc_t w2rsrc = HP3D_MAKE_RSRC(p.w2, (unsigned)(p.H * 32) * 4u);
    const int bl = (kh * 32 + n) * 16;                   // this lane's 16 bytes inside a 1 KB fragment
    const int arow = (n * PW_PITCH + kh * 4) * 4;        // byte offset of this lane's A fragment in row block 0, 8-channel step 0
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    __syncthreads();

    const int nslab = p.H / PW_C;
    for (int slab = 0; slab < nslab; ++slab) {
        // ---- stage 1: S[64 px][32 hidden of this wave] = x W1 ------------------------------------------------------------------------------
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        const int co32 = slab * 4 + wave;
        // weight fragments through a ring of PW_RING 8-channel steps (left to itself the compiler keeps ONE step in flight: an L2 round trip
        // per 512 matrix-core cycles), A fragments one step ahead; the scheduling barriers pin that order
        f32x4 bw[PW_RING], a0[2], a1[2];
#pragma unroll
        for (int t = 0; t < PW_RING; ++t) bw[t] = HP3D_BUFFER_LOAD16(w1rsrc, bl, (t * CO32 + co32) * 1024);
        a0[0] = *(const f32x4*)((const char*)Xs + arow);
        a1[0] = *(const f32x4*)((const char*)Xs + arow + 32 * PW_PITCH * 4);
#pragma unroll
        for (int c8 = 0; c8 < PW_C / 8; ++c8) {
            HP3D_SCHED_BARRIER();
            if (c8 + 1 < PW_C / 8) {
                a0[(c8 + 1) & 1] = *(const f32x4*)((const char*)Xs + arow + (c8 + 1) * 32);
                a1[(c8 + 1) & 1] = *(const f32x4*)((const char*)Xs + arow + 32 * PW_PITCH * 4 + (c8 + 1) * 32);
            }
            const f32x4 b = bw[c8 % PW_RING];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = HP3D_MFMA_32x32x2(a0[c8 & 1][j], b[j], acc[0]);
                acc[1] = HP3D_MFMA_32x32x2(a1[c8 & 1][j], b[j], acc[1]);
            }
            HP3D_SCHED_BARRIER();
            if (c8 + PW_RING < PW_C / 8) bw[c8 % PW_RING] = HP3D_BUFFER_LOAD16(w1rsrc, bl, ((c8 + PW_RING) * CO32 + co32) * 1024);
        }
        // bias + leaky-ReLU, slab -> LDS: accumulator r of lane l is pixel (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the half, hidden channel 32 wave + (l & 31)
        const float b1 = p.b1[co32 * 32 + n];
        if (slab) __syncthreads();                       // stage 2 of the previous slab has read Hs
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[m][r] + b1;
                if (p.act1) v = fmaxf(v, HP3D_LEAKY_SLOPE * v);
                Hs[(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * PW_PITCH + wave * 32 + n] = v;
            }
        __syncthreads();
        // ---- stage 2: Y[32 px of half w & 1][32] += S[:, 64 channels of half w >> 1] W2 ------------------------------------------------------
        const int ph = wave & 1, ch = wave >> 1;
        f32x4 b2w[8];          // (the slab's eight W2 fragments of this wave: all requested up front, they do not depend on the slab in LDS)
#pragma unroll
        for (int c8l = 0; c8l < 8; ++c8l) b2w[c8l] = HP3D_BUFFER_LOAD16(w2rsrc, bl, (slab * 16 + ch * 8 + c8l) * 1024);
#pragma unroll
        for (int c8l = 0; c8l < 8; ++c8l) {
            const int c8 = ch * 8 + c8l;
            const f32x4 a = *(const f32x4*)((const char*)Hs + arow + ph * 32 * PW_PITCH * 4 + c8 * 32);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc2 = HP3D_MFMA_32x32x2(a[j], b2w[c8l][j], acc2);
        }
    }
    // ---- the two channel halves meet: waves 2, 3 hand their sums to waves 0, 1 through LDS (Xs is free), then bias + activation + store ------
    __syncthreads();
    if (wave >= 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Xs[((wave & 1) * 16 + r) * 64 + lane] = acc2[r];
    }
    __syncthreads();
    if (wave < 2) {
        const float b2 = p.b2[n];
        const hp3d_rsrc_t orsrc = HP3D_MAKE_RSRC(p.out, (unsigned)p.npix * (unsigned)p.out_cs * 4u);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = (acc2[r] + Xs[(wave * 16 + r) * 64 + lane]) + b2;
            if (p.act2) v = fmaxf(v, HP3D_LEAKY_SLOPE * v);
            const long gp = pix0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            HP3D_BUFFER_STORE4(orsrc, v, (gp < p.npix && n < p.cout_store) ? (int)(gp * p.out_cs + n) * 4 : (int)0x80000000, 0);
        }
    }
}

}  // namespace

// 1: the pair can run here (both 1x1 / stride 1, 128 input channels, hidden width a multiple of 128, at most 32 outputs)
int conv_pw2_eligible(int cin_pad, int hidden_pad, int cout2_pad, long npix, int in_cs, int out_cs) {
    if (cin_pad != PW_C || hidden_pad % PW_C || hidden_pad < PW_C || cout2_pad != 32 || out_cs < 1 || in_cs < PW_C) return 0;
    if (npix * in_cs * 4 >= (1L << 31) || npix * out_cs * 4 >= (1L << 31)) return 0;
    return 1;
}

int conv_pw2_launch(const Pw2Params& p, hipStream_t s) {
    if (!conv_pw2_eligible(PW_C, p.H, 32, p.npix, p.in_cs, p.out_cs) || (p.in_cs & 3) || ((uintptr_t)p.in & 15)) return -1;
    static bool attr_done[64] = {};
    if (hp3d_first_use_on_device(attr_done))
        (void)hipFuncSetAttribute((const void*)conv_pw2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PW_SMEM_BYTES);
    const unsigned tiles = (unsigned)((p.npix + PW_PX - 1) / PW_PX);
    HP3D_LAUNCH(conv_pw2_kernel, dim3(tiles), dim3(256), PW_SMEM_BYTES, s, p);
    return 0;
}

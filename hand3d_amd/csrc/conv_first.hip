// conv_first.hip -- conv1_1 of HandSegNet / PoseNet2D: 3x3, stride 1, SAME, Cin = 3 -> Cout = 64, bias + leaky-ReLU
// (nets/ColorHandPose3DNetwork.py:144,183 through NetworkOps.conv_relu, utils/general.py:36-59), float32.
//
// The layer is store-bound (0.2 GFLOP per 13 MB of output per image), and on the general kernel it was bound by
// something else again: 25 600 workgroups of ~8 us, each paying its own weight staging and window latency (0.43 ms at
// 320x320, B=32 = 1.9 TB/s).  Here:
//   * K = 27 (padded to 28): 14 k-steps of v_mfma_f32_32x32x2_f32 per 32 pixels x 32 couts;
//   * the 28 x 64 filter matrix lives in 28 registers per lane for the whole workgroup (both 32-cout halves);
//   * the im2col row of a pixel is never materialised: lane (pixel, k-half) gathers its 14 operands straight from the
//     image with buffer loads (zero padding and the K tail = out-of-range offsets); the next tile's 14 loads fly under
//     the current tile's MFMAs and stores;
//   * the bias rides in the spare K row (k = 27: weight row = bias, image operand = 1.0), so the epilogue is only the
//     leaky-ReLU and the stores;
//   * stores (the layer is bound by its 13 MB / image of output): D[pixel][cout], one 4-byte store per accumulator
//     register -- 32 lanes cover the 32 couts (128 contiguous bytes) of one pixel.  Measured alternatives
//     (profiles/r02_tuning_notes.md): 16-byte stores straight from a D[cout][pixel] register layout (32 bytes per pixel
//     and instruction) are 30 % SLOWER; the same block transposed through LDS into 16-byte full-row stores runs at exactly
//     the same speed as this form -- the store form is not the limiter.  A pure fill kernel reaches 5.1-5.8 TB/s on this
//     GPU (scripts/micro/write_bw.hip); this kernel writes at 3.6 TB/s next to its gathers and 28 MFMAs per tile;
//   * no LDS, ~100 registers: four waves per SIMD hide the rest; a workgroup walks a strip of tiles;
//   * F16 = true (half-precision trunks, BASELINE config 5): operands rounded to half and multiplied by v_mfma_f32_32x32x16_f16
//     (K = 27 + two bias rows in two K = 16 steps: 4 MFMAs of 32 cycles per 32 x 64 block instead of 28 of 64 -- the f32
//     form was matrix-bound at half-precision store rates), float32 accumulate, and the 32 x 64 block leaves as HALVES: 2-byte stores would be issue-bound, so the wave's
//     block is transposed through a private 4.5 KB LDS slab (ds_write_b64 from the D[cout][pixel] register layout, where
//     registers 4a..4a+3 are four consecutive couts of a pixel) into 16-byte stores in which 8 consecutive lanes write the
//     whole 128-byte row of a pixel (config 5, 480x640 B=128: 4.5 ms on the general kernel -> see profiles/r02_tuning_notes.md).
#include "hp3d_common.h"
#include <algorithm>

namespace {

constexpr int FT_TH = 8, FT_TW = 16;       // tile: 8 rows x 16 pixels = 4 MFMA row blocks (one per wave)
constexpr int FK = 14;                      // k-steps (K = 28 >= 27)

constexpr int FT_PITCH_H = 72;              // halves per pixel row of the F16 transpose slab (64 + 8: 144 B)
template <bool F16>
HP3D_KERNEL(256)
void conv_first_kernel(const ConvParams p, int tq, int trem) {
    HP3D_DYN_SMEM(slab_all);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = HP3D_READFIRSTLANE(tid >> 6);
    const int m = lane & 31, kh = lane >> 5;
    // filter operands: k = 2 kk + kh, engine channel e = k = (r*3+s)*3 + c of the packed K=32 x 64 matrix
    // wpk[c8][co32][h][n][j] (engine.hip:pack_conv, mode 1)
    // F16: v_mfma_f32_32x32x16_f16, K = 32 in two steps; operand slot q = 8 j + i of step j is k = 16 j + 8 kh + i.  The bias
    // rides in TWO spare rows as hi + lo halves (k = 27, 28; image operand 1.0): float32 bias to 2^-22
    constexpr int NK = F16 ? 16 : FK;
    auto kof = [&](int q) { return F16 ? 16 * (q >> 3) + 8 * kh + (q & 7) : 2 * q + kh; };
    auto wat = [&](int nb, int k) { return p.wpk[(((k >> 3) * 2 + nb) * 2 + ((k >> 2) & 1)) * 128 + m * 4 + (k & 3)]; };
    float bw[2][FK];
    f32x4 bwh[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        if (F16) {
            const float bias = p.bias[nb * 32 + m], bias_hi = (float)(hp3d_f16)bias;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f16x8 h;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = 16 * j + 8 * kh + i;
                    h[i] = (hp3d_f16)(k < 27 ? wat(nb, k) : k == 27 ? bias_hi : k == 28 ? bias - bias_hi : 0.f);
                }
                bwh[nb][j] = __builtin_bit_cast(f32x4, h);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < FK; ++kk) {
                const int k = 2 * kk + kh;
                bw[nb][kk] = k == 27 ? p.bias[nb * 32 + m]    // the spare K row carries the bias (its image operand is 1)
                                     : wat(nb, k);
            }
        }
    }

    // this workgroup's run of tiles in the launch's linear order (image, tile row, tile): the first `trem` workgroups walk tq + 1, the others tq
    const int wg = blockIdx.x;
    const int t0 = wg * tq + min(wg, trem), ntile = tq + (wg < trem ? 1 : 0);
    int tx = t0 % p.tiles_x, ty, b;
    {
        const int row = t0 / p.tiles_x;
        ty = row % p.tiles_y;
        b = row / p.tiles_y;
    }
    int ntx = tx, nty = ty, nb = b;             // the tile whose operands are gathered next
    auto advance = [&](int& x, int& yy, int& bb) {
        if (++x == p.tiles_x) { x = 0; if (++yy == p.tiles_y) { yy = 0; ++bb; } }
    };

    const hp3d_rsrc_t irsrc = HP3D_MAKE_RSRC(p.in, (unsigned)p.B * (unsigned)(p.H * p.W) * 12u);
    const hp3d_rsrc_t orsrc = HP3D_MAKE_RSRC(p.out, (unsigned)p.B * (unsigned)(p.H * p.W) * (unsigned)p.out_cs * (F16 ? 2u : 4u));
    constexpr int OOR = (int)0x80000000;

    // image operand of lane (pixel m of this wave's 2 x 16 row block, k-half kh), k-step kk: image[y+r-1][x+s-1][c]
    auto gather = [&](int b, int ty, int tx, float (&a)[NK]) {
        const int y = ty * FT_TH + 2 * wave + (m >> 4);
        const int x = tx * FT_TW + (m & 15);
#pragma unroll
        for (int q = 0; q < NK; ++q) {
            const int k = kof(q);
            const int r = k / 9, s = (k / 3) % 3, c = k % 3;
            const int yy = y + r - 1, xx = x + s - 1;
            const bool ok = k < 27 && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            a[q] = HP3D_BUFFER_LOAD4(irsrc, ok ? (((b * p.H + yy) * p.W + xx) * 3 + c) * 4 : OOR, 0);
        }
        if (F16) { if (kh) a[11] = a[12] = 1.0f; }       // k = 27, 28: multiply the two bias rows
        else if (kh) a[FK - 1] = 1.0f;                    // k = 27: multiplies the bias row
    };
    float a_cur[NK], a_nxt[NK];
    gather(nb, nty, ntx, a_cur);
    for (int it = 0; it < ntile; ++it) {
        if (it + 1 < ntile) { advance(ntx, nty, nb); gather(nb, nty, ntx, a_nxt); }
        f32x16 acc0, acc1;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (F16) {
            // D[cout][pixel]: this lane holds pixel m; accumulator register 4a + j of half nb is cout 32 nb + 8 a + 4 kh + j
            f32x4 ah[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f16x8 h;
#pragma unroll
                for (int i = 0; i < 8; ++i) h[i] = (hp3d_f16)a_cur[8 * j + i];
                ah[j] = __builtin_bit_cast(f32x4, h);
            }
            acc0 = HP3D_MFMA_32x32x16_F16(bwh[0][0], ah[0], zero);
            acc1 = HP3D_MFMA_32x32x16_F16(bwh[1][0], ah[0], zero);
            acc0 = HP3D_MFMA_32x32x16_F16(bwh[0][1], ah[1], acc0);
            acc1 = HP3D_MFMA_32x32x16_F16(bwh[1][1], ah[1], acc1);
            hp3d_f16* slab = (hp3d_f16*)slab_all + wave * (32 * FT_PITCH_H);
            HP3D_WAVE_LDS_SYNC();                 // the previous tile's slab reads are done
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f16x4 h0, h1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v0 = acc0[4 * a + j], v1 = acc1[4 * a + j];
                    if (p.act) { v0 = fmaxf(v0, HP3D_LEAKY_SLOPE * v0); v1 = fmaxf(v1, HP3D_LEAKY_SLOPE * v1); }
                    h0[j] = (hp3d_f16)v0; h1[j] = (hp3d_f16)v1;
                }
                *(f16x4*)(slab + m * FT_PITCH_H + 8 * a + 4 * kh) = h0;
                *(f16x4*)(slab + m * FT_PITCH_H + 32 + 8 * a + 4 * kh) = h1;
            }
            HP3D_WAVE_LDS_SYNC();
            // lane -> (pixel 8 j + (lane >> 3), 16-byte chunk lane & 7): 8 consecutive lanes = one pixel's 64 halves
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pm = 8 * j + (lane >> 3), ch = lane & 7;
                const int oy = ty * FT_TH + 2 * wave + (pm >> 4), ox = tx * FT_TW + (pm & 15);
                const f32x4 v = *(const f32x4*)(slab + pm * FT_PITCH_H + ch * 8);
                HP3D_BUFFER_STORE16(orsrc, v, (oy < p.H && ox < p.W) ? (((b * p.H + oy) * p.W + ox) * p.out_cs + ch * 8) * 2 : OOR, 0);
            }
        } else {
            acc0 = HP3D_MFMA_32x32x2(a_cur[0], bw[0][0], zero);
            acc1 = HP3D_MFMA_32x32x2(a_cur[0], bw[1][0], zero);
#pragma unroll
            for (int kk = 1; kk < FK; ++kk) {
                acc0 = HP3D_MFMA_32x32x2(a_cur[kk], bw[0][kk], acc0);
                acc1 = HP3D_MFMA_32x32x2(a_cur[kk], bw[1][kk], acc1);
            }
            // accumulator register r <-> pixel (r & 3) + 8 (r >> 2) + 4 kh of the row block, column = cout m of the half
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pm = (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int oy = ty * FT_TH + 2 * wave + (pm >> 4), ox = tx * FT_TW + (pm & 15);
                const int base = (oy < p.H && ox < p.W) ? (((b * p.H + oy) * p.W + ox) * p.out_cs + m) * 4 : OOR;
                float v0 = acc0[r], v1 = acc1[r];
                if (p.act) { v0 = fmaxf(v0, HP3D_LEAKY_SLOPE * v0); v1 = fmaxf(v1, HP3D_LEAKY_SLOPE * v1); }
                // non-temporal stores (round 5): the 64-channel activation is written once and read back one layer later, after 0.5-0.8 GB
                // have passed the 256 MB memory-side cache.  Measured at B = 32, 320x320: this layer 0.237 -> 0.251 ms (HandSegNet) and
                // 0.157 -> 0.126 ms (PoseNet2D: 4.46 TB/s), the conv1_2 behind it 0.706 -> 0.686 and 0.458 -> 0.441: -0.05 ms per step
                HP3D_BUFFER_STORE4_NT(orsrc, v0, base, 0);
                HP3D_BUFFER_STORE4_NT(orsrc, v1, base, 128);
            }
        }
#pragma unroll
        for (int q = 0; q < NK; ++q) a_cur[q] = a_nxt[q];
        advance(tx, ty, b);
    }
}

}  // namespace

// conv1_1 shape only: 3x3 / stride 1 / Cin 3 / Cout 64 / float32 in and out, tensors inside 32-bit byte offsets
int conv_first_eligible(int k, int stride, int Cin, int Cout, int B, int H, int W, int out_cs, int f16) {
    if (k != 3 || stride != 1 || Cin != 3 || Cout != 64 || out_cs < 64 || out_cs % 8 || B < 1) return 0;
    return (long)H * W * out_cs * (f16 ? 2 : 4) < (1L << 31);      // one image inside 32-bit offsets (the launcher chunks the batch)
}

// resident workgroups of the kernel over the chip (3 per CU at 125 + 32 registers; asked, not assumed)
static int conv_first_slots(bool f16) {
    static int per_cu[2] = {0, 0};
    if (!per_cu[f16]) {
        int n = 0;
#ifndef HP3D_EMU
        const void* k = f16 ? (const void*)conv_first_kernel<true> : (const void*)conv_first_kernel<false>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, f16 ? 4 * 32 * FT_PITCH_H * 2 : 0) != hipSuccess) { (void)hipGetLastError(); n = 0; }
#endif
        per_cu[f16] = n >= 1 ? n : 3;
    }
    return per_cu[f16] * hp3d_num_cus();
}

int conv_first_launch(const ConvParams& pin, hipStream_t s, int balanced) {
    ConvParams p = pin;
    p.tiles_x = (p.W + FT_TW - 1) / FT_TW;
    p.tiles_y = (p.H + FT_TH - 1) / FT_TH;
    // the kernel addresses input and output with 32-bit byte offsets from the tensor base: batches whose output passes 2^31
    // bytes (config 5: 128 x 480 x 640 x 64 halves = 5 GB) run as consecutive launches over image ranges
    const long per_img = (long)p.H * p.W * p.out_cs * (p.f16 ? 2 : 4);
    const int maxb = (int)std::max<long>(1, ((1L << 31) - 1) / per_img);
    for (int b0 = 0; b0 < pin.B; b0 += maxb) {
        p.B = std::min(maxb, pin.B - b0);
        p.in = pin.in + (size_t)b0 * p.H * p.W * 3;
        p.out = (float*)((char*)pin.out + (size_t)b0 * per_img);
        // A workgroup walks a run of consecutive tiles (tile rows continue into the next row / image), every run within one tile of the same
        // length, one workgroup per resident slot: the launch is ONE balanced round.  (Rounds 2-4 walked whole tile rows: 1280 workgroups of
        // 20 tiles on 768 slots at 320x320, B = 32 -- 1.67 rounds; round 5.)  Small launches: runs of at least 4 tiles (filter load + first gather).
        const long total = (long)p.B * p.tiles_y * p.tiles_x;
        long nwg = (long)conv_first_slots(p.f16 != 0);
        if (!balanced) nwg = (long)p.B * p.tiles_y;            // (option "first_walk" = "rows": the previous decomposition, for A/B timing)
        nwg = std::max<long>(1, std::min(nwg, total / 4));
        const int tq = (int)(total / nwg), trem = (int)(total % nwg);
        if (p.f16)      // half-precision output [B,H,W,out_cs halves]
            HP3D_LAUNCH(conv_first_kernel<true>, dim3((unsigned)nwg), dim3(256), 4 * 32 * FT_PITCH_H * 2, s, p, tq, trem);
        else
            HP3D_LAUNCH(conv_first_kernel<false>, dim3((unsigned)nwg), dim3(256), 0, s, p, tq, trem);
    }
    return 0;
}

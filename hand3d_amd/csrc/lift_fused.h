// lift_fused.h -- parameters of the one-launch lifting stage (lift_fused.hip); shared with the executor.
#pragma once
#include "hp3d_common.h"

struct LiftFusedParams {
    const float* sm;          // [B,32,32,sm_cs] score maps (21 real channels, the rest zero)
    int sm_cs;
    const float* hs;          // [B,2] hand side
    // conv layers: index t * 6 + i, t = 0 PosePrior / 1 ViewpointNet, i = 0..5 ({s1, s2} x 3); weights packed
    // [ceil(Cout/64)][cin (padded to a multiple of 16, zero rows)][tap 9][64 couts (zero padded)]
    const float* w[12];
    const float* b[12];
    int cin[12], cout[12];
    // fc layers: index t * 3 + f: fc_rel0, fc_rel1, fc_xyz | fc_vp0, fc_vp1, fc_vp_u ([in,out] row-major)
    const float* fw[6];
    const float* fb[6];
    int fc_in[6], fc_out[6];
    const float* bn_w;        // PosePriorNetwork variant "bottleneck": fc_bottleneck [512,30] + bias, fc_xyz then is [30,63]; else null
    const float* bn_b;
    float* act[2][2];         // per tower, ping-pong activations: B x 32 x 32 x 64 floats each
    float* fcp[2][2];         // per tower: partial sums of fc layer 0 / 1, [K slices][B][Cout]
    float* out[2];            // [B,63] canonical coordinates; [B,3] rotation vector
    unsigned* bar;            // grid-barrier counter (zeroed by the launcher)
    unsigned* err;            // error word in mapped host memory: set to 1 by a workgroup whose barrier wait ran out (the
                              // executor checks it at every synchronisation point and fails the call: hp3d_sync / finish_op)
    int B, towers;            // towers: bit 0 PosePrior, bit 1 ViewpointNet
    int phase_lo, phase_hi;   // set by the launcher
};
int lift_fused_launch(const LiftFusedParams& p, hipStream_t s);

/*
 * hp3d.h -- C-ABI of libhp3d.so: the MI355X (gfx950) engine behind the Python call
 * surface of lmb-freiburg/hand3d's ColorHandPose3DNetwork / PosePriorNetwork.
 *
 * The reference has no FFI of its own: its "operator API" is the method surface of two
 * Python classes whose arithmetic runs inside TensorFlow 1.3.  Each entry
 * point below names the reference interface it replaces (file:line relative to the
 * reference tree).  The reference-side binding (ctypes) is hand3d_amd/_lib.py and is
 * reproduced in INTEGRATION.md.
 *
 * Conventions
 *   - return 0 on success, a negative hp3d_status otherwise; hp3d_last_error() gives text;
 *   - no C++ exception crosses this boundary, no torch/TF types appear in it;
 *   - all tensors are float32, NHWC, contiguous (the reference's layout: utils/general.py:40-46);
 *   - "host" entry points take caller-owned host buffers, are synchronous and never retain
 *     a pointer; "_dev" entry points take device pointers (hipMalloc'd by anyone in this
 *     process, e.g. a torch tensor's data_ptr) and are stream-ordered on hp3d_stream(ctx):
 *     call hp3d_sync() before reading results;
 *   - any output pointer may be NULL (that output is then not copied out);
 *   - a context is thread-compatible: no concurrent calls on one context.
 */
#ifndef HP3D_H
#define HP3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hp3d_ctx hp3d_ctx;

typedef enum {
    HP3D_OK = 0,
    HP3D_ERR_ARG = -1,        /* bad argument / shape (the reference's bare asserts)            */
    HP3D_ERR_HIP = -2,        /* a HIP runtime call failed                                      */
    HP3D_ERR_WEIGHTS = -3,    /* missing / mis-shaped variable at finalize, or not finalized    */
    HP3D_ERR_UNSUPPORTED = -4,/* e.g. evaluation=False / train=True, unsupported geometry          */
    HP3D_ERR_NOMEM = -5
} hp3d_status;

/* PosePriorNetwork variants -- nets/PosePriorNetwork.py:59-95 */
enum { HP3D_VARIANT_DIRECT = 0, HP3D_VARIANT_BOTTLENECK = 1, HP3D_VARIANT_PROPOSED = 2,
       HP3D_VARIANT_LOCAL = 3 /* 'local' and 'local_w_xyz_loss': + bone_rel_trafo_inv, utils/relative_trafo.py:243-295 */ };
/* activation fused behind a conv / fc -- utils/general.py:55-59,132-136 */
enum { HP3D_ACT_NONE = 0, HP3D_ACT_LEAKY = 1 };

int hp3d_abi_version(void);
/* HIP devices visible to this process (hipGetDeviceCount); needs no context.  No reference counterpart (one tf.Session on whatever
 * device TensorFlow picks, run.py:44-50): a launcher uses it to refuse `--gpus N` on a box with fewer devices BEFORE any rank
 * starts, instead of leaving N - n ranks to fail one by one inside a rendezvous.  Returns 0 or HP3D_ERR_HIP (then *count = 0). */
int hp3d_device_count(int* count);
/* PCI address "dddd:bb:dd.f" of HIP device `device` (hipDeviceGetPCIBusId) into buf (cap >= 13); needs no context.  No reference
 * counterpart (same single-session script): a multi-rank launcher reads the device's NUMA node from
 * /sys/bus/pci/devices/<address>/numa_node and pins the rank's host threads next to its GPU (bench.py: pin_to_gpu_numa).
 * Returns 0, HP3D_ERR_HIP, or HP3D_ERR_UNSUPPORTED in the CPU interpreter build. */
int hp3d_device_pci_bus_id(int device, char* buf, int cap);

/* ---- context ----------------------------------------------------------------------------
 * replaces: tf.Session(config=...) + graph construction (run.py:44-50).                     */
int hp3d_create(int device, hp3d_ctx** out);
int hp3d_destroy(hp3d_ctx* ctx);
const char* hp3d_last_error(hp3d_ctx* ctx);          /* ctx may be NULL: last global error   */
void* hp3d_stream(hp3d_ctx* ctx);                    /* the hipStream_t all work is queued on */
int hp3d_sync(hp3d_ctx* ctx);
/* options: "empty_reduce" = "inf" | "fltmax" (oracle/general.py EMPTY_REDUCE);
 *          "conv_impl"    = "mfma" (default: direct MFMA kernel, and float32 Winograd F(2x2,3x3) for the stride-1 3x3
 *                            layers with Cout%64==0 and the 7x7 layers -- taken as nine 3x3 blocks -- whenever the grid
 *                            fills the chip, >= 256 work items) | "direct" (never Winograd: bit-identical to an fmaf
 *                            chain) | "winograd" (whenever the shape allows; per-op hp3d_conv2d then refuses other
 *                            shapes) | "naive" (debug cross-check kernel, never a fallback);
 *          "streams"      = "auto" (default) | "1" | "2": whole-path calls run the two halves of their batch concurrently
 *                            on two HIP streams (second arena, shared weights; fills the tail rounds of the persistent
 *                            kernels and the launch gaps).  auto = 2 when each half has at least 2.4 M input pixels (B = 32 at
 *                            480x640, two chunks of 32 at 240x320 / 320x320; 3 M until round 6) or, round 6, for 40 <= B < 64 in float32 mode (a full
 *                            chunk followed by a latency-bound remainder on one stream: B = 40 2487 -> 2616 images/s), else 1 (round 4: with "wino4_tail" the one-stream run no longer
 *                            loses a partial last round, and halves of 16 images fill the chip worse).  Results equal "1" to
 *                            rounding (images are independent; a half may take the small-batch kernel plan);
 *                            profiling / graph replay use one stream.  The halves overlap on the device-pointer entry points
 *                            (hp3d_infer_full_dev ...); with HOST output buffers the first half's pageable device->host
 *                            copies block the host before the second half is enqueued;
 *          "wino_splitk"  = "1" (default) | "0": Winograd layers whose work items under-fill the chip (small batches) split
 *                            their channel steps over up to 16 workgroups and sum float32 partials in a fixed order;
 *          "wino2"        = "auto" (default) | "0" | "1": which of the two float32 Winograd kernels a 3x3 / 7x7 layer takes when both
 *                            can run it.  conv_wino.hip gives one wave a whole SIMD (256 accumulators, 32-channel steps, 128-cout
 *                            items: long reductions); conv_wino2.hip runs two workgroups per CU on v_mfma_f32_16x16x4_f32 (128
 *                            accumulators, 16-channel steps, 64-cout items), so one item's epilogue / waits are the other's MFMA
 *                            time and a launch has four times as many, four times finer work items.  auto = a per-layer cost
 *                            model of both kernels' rounds of work items (engine.hip:wino2_auto): in practice conv_wino2 for
 *                            launches that under-fill or badly quantise conv_wino's grid (small batches), conv_wino for B = 32;
 *                            "1" = wherever the shape allows (tests); "0" = never.  Same arithmetic: results agree to
 *                            accumulation order;
 *          "wino4"        = "auto" (default) | "pose" | "0" | "1": Winograd F(4x4,3x3) (conv_wino4.hip: 36 products per 4x4 outputs, 2.25
 *                            multiply-adds per output instead of F(2x2,3x3)'s 4) for the 3x3 / 7x7 layers.  auto = every trunk layer a
 *                            per-layer cost model of the three Winograd kernels gives to it (filled launches: B >= 4 ... 8); "pose" =
 *                            PoseNet2D only (HandSegNet's score map feeds the mask threshold: with "auto" ~2 % more images differ from
 *                            the all-direct-kernel run in a mask pixel -- never in the crop box or, beyond 1e-4, in a keypoint over 512
 *                            images, scripts/mask_flip_rate.py); "1" = wherever the shape allows (tests); "0" = never.  Float32
 *                            throughout; end to end it moves heat-maps by 5e-6 and 3-D keypoints by 3e-6;
 *          "wino7"        = "auto" (default) | "0" | "1": PoseNet2D's ten 7x7 layers (ColorHandPose3DNetwork.py:206-215) as Winograd
 *                            F(4x4,4x4) over the filter's four 4x4-tap blocks (conv_wino7.hip, round 5: 169 instead of 289 plane products per
 *                            16 outputs, the transformed input shared by the four blocks, one work item = 16 tiles x 64 couts with no channel
 *                            split) when the launch fills the chip (>= 160 work items: B >= 20 on the 32 x 32 score maps) | never (the
 *                            nine-3x3-block form on conv_wino4.hip / conv_wino2.hip) | whenever the shape allows (tests).  Float32 throughout,
 *                            the same rounding error as the nine-block form (profiles/r05_wino7_numerics.md).  Launches below that fill
 *                            (small batches) run the same kernel with the 16-channel chunks split over workgroups -- raw 4x4 sums per
 *                            split, added in order by the reduce launch with bias and activation -- when "wino_splitk" = "1";
 *          "wino7_ksplit" = "auto" (default) | N: the number of channel splits of such a launch (auto: CUs / work items, at most one split
 *                            per chunk; N: tests and tuning, clamped to the number of chunks);
 *          "pw2"          = "1" (default) | "0" | "force": the 1x1 head pairs of both trunks (conv6_1 + conv6_2; conv5_1 + conv5_2, conv6_6 + conv6_7,
 *                            conv7_6 + conv7_7: ColorHandPose3DNetwork.py:160-161,202-203,213-214) as ONE launch each with the wide intermediate in
 *                            LDS (conv_pw2.hip, round 5) when the launch has a workgroup of 64 pixels per CU | two launches of the general kernel |
 *                            whenever the shapes allow (tests).  Float32 mode only;
 *          "wino4_tail"   = "1" (default) | "0": conv_wino4.hip deals its work items round-robin to one workgroup per CU; when the last
 *                            round is at most half full (HandSegNet's 40x40 layers at B = 32: 800 items on 256 CUs = 3.125 rounds) its
 *                            items run as channel slices -- one piece per CU, raw sums to a scratch of 2 pieces x CUs x 128 KB = 64 MiB per context on a 256-CU
 *                            MI355X (the second-stream child context grows its own), added in slice order by a small
 *                            reduce launch (deterministic; the summation order differs from the unsplit item: float32 rounding);
 *          "first_touch"  = "auto" (default) | "0" | "1": conv1_1's kernel is store-bound and gathers its operands one tile ahead; an input
 *                            image that is COLD in the memory system (the caller's device buffer, an upload -- not the crop or the uint8
 *                            front end's output, which the kernel in front has just written) costs it a third of its rate.  auto: such an
 *                            image of 8 ... 128 MB is streamed once through the memory-side cache first (13 us for 39 MB; HandSegNet's
 *                            conv1_1 at B = 32, 320 x 320: 0.253 -> 0.175 ms) | never | always.  Reads only: results unchanged;
 *          "first_touch_beside" = "1" (default) | "0": that read pass runs on a second stream beside conv1_1 | in front of it;
 *          "first_walk"   = "balanced" (default) | "rows": conv1_1's kernel (conv_first.hip) gives every resident workgroup one run of
 *                            consecutive 8 x 16 tiles, all runs within a tile of the same length | a whole tile row per workgroup (rounds 2-4;
 *                            kept for A/B timing).  Bit-identical results;
 *          "lift_overlap" = "1" (default) | "0": the unfused lifting stage (batches above 4) runs ViewpointNet on a second stream beside
 *                            PosePrior (the towers share only their input, ColorHandPose3DNetwork.py:231-235; 12 + 12 short dependent launches)
 *                            | one after the other.  Same kernels, same results bit for bit;
 *          "lift_fused"   = "auto" (default) | "0" | "1": PosePrior + ViewpointNet + the lifting epilogue
 *                            (ColorHandPose3DNetwork.py:221-334) as ONE persistent launch with grid barriers (lift_fused.hip)
 *                            instead of 24 launches.  auto = for at most 4 images per call (the stage is latency-bound there:
 *                            0.33 -> 0.14 ms at B = 1); "1" = always (tests; needs the whole grid resident, which the launcher
 *                            checks); "0" = never.  Results agree to accumulation order (2-5e-7);
 *          "micro_batch"  = "N" | "auto": whole-path calls (hp3d_infer_full*) run as consecutive chunks of at most N
 *                            images ("0" = never split; default "auto" = at most 32 in float32 mode -- fewer when H x W x 64 floats x N
 *                            would pass 2^31 bytes, e.g. 480x640: balanced chunks of <= 27 -- and no split with f16 trunks).
 *                            Bit-identical to making the calls chunk by chunk;
 *          "f16_impl"     = "h16" (default) | "mfma" | "h16_force": with half-precision trunks (hp3d_finalize_weights dtype 1),
 *                            the 3x3 / stride-1 layers with Cin >= 64 run on the half-precision trunk kernel (conv_h16.hip)
 *                            whenever their grid fills the chip | never (general kernel only) | whenever the shape allows
 *                            (tests).  Same MFMA and packed weights either way: results agree to accumulation order;
 *          "f16_k7k1"     = "1" (default) | "0": with "f16_impl" on conv_h16.hip, PoseNet2D's 7x7 score-map stages and the 1x1 layers with
 *                            >= 64 couts also run on it (single-buffer forms, patch of (16 + k - 1)^2 pixels) | on the general kernel.
 *                            Same MFMA and packed weights: results agree to accumulation order;
 *          "f16_fuse12"   = "1" (default) | "0" | "ring" | "resident": with half-precision trunks and the layer on conv_h16.hip, conv1_1 is computed
 *                            inside conv1_2's patch stage (one launch for conv1_1 + conv1_2 + max-pool; conv1_1's activation
 *                            never reaches HBM).  Bit-identical to the two-launch form.  Two forms of that launch (round 6): "ring" = two
 *                            workgroups per CU, conv1_2's filters through a register ring; "resident" = one workgroup per CU with the
 *                            filters resident in registers and the next tile's patch built between the MFMAs of the current one
 *                            (conv_h16_first_kernel); "1" takes "resident" from four tiles per CU on, else "ring";
 *          "wino4_split"  = "0" (default) | "auto" | "1" (round 6): the 3x3 / stride-1 trunk layers with Cin >= 128 whose launch fills the chip run
 *                            Winograd F(4x4,3x3) with the plane products on v_mfma_f32_16x16x32_bf16 over THREE bfloat16 pieces per operand (six
 *                            products, float32 accumulate: conv_wino4s.hip) | never | wherever the shape allows (tests).  Float32 in and out; per
 *                            layer as exact as "wino4" (0.7 ... 1.6x its error per shape, profiles/r06_split_numerics.md) and, when built, 1.03-1.09x its (since conv_wino4's late round-6 gains: 0.98x) speed:
 *                            an option, not the default;
 *          "fc_tail"      = "1" (default) | "0" (round 6): ViewpointNet's three FC layers (ColorHandPose3DNetwork.py:299-307) as two launches -- the K slices
 *                            of fc_vp0, then their fixed-order reduction + fc_vp1 + fc_vp_u in one -- | as three partial + reduce pairs.  Same sums up
 *                            to the order inside fc_vp1 (1e-7);
 *          "tiny_gemm"    = "1" (default) | "0" (round 6): ViewpointNet/conv_vp_2_2 (3x3 / stride 2 on the 8x8 map, 16 output pixels per image) as a split-K
 *                            GEMM over its output pixels with the HWIO filter as the matrix | on the general kernel's 8x8-pixel tiles;
 *          "kp_up_side"   = "1" (default) | "0" (round 6): with "lift_overlap", the whole-path calls' heat-map up-sampling runs behind ViewpointNet on the
 *                            second stream | behind PosePrior on the first.  Bit-identical;
 *          "graph"        = "0" | "1": the device-pointer entry points (hp3d_infer_full_dev, hp3d_posenet2d_dev) replay
 *                            their launch sequence as one hipGraph from the third identical call on (same shape and
 *                            pointers); meant for small batches.  Default "0".                               */
int hp3d_set_option(hp3d_ctx* ctx, const char* key, const char* value);

/* ---- weights ----------------------------------------------------------------------------
 * replaces: ColorHandPose3DNetwork.init / PosePriorNetwork.init, i.e. pickle dict ->
 * tf.contrib.framework.assign_from_values (nets/ColorHandPose3DNetwork.py:34-59,
 * nets/PosePriorNetwork.py:36-57).  `tf_var_name` is the pickle key, e.g.
 * "PoseNet2D/conv6_1/weights"; conv weights HWIO [k,k,Cin,Cout], FC weights [in,out],
 * biases [out] (utils/general.py:41-50,117-126).  Data is copied; unknown names are
 * rejected (HP3D_ERR_ARG), mis-shaped ones too.                                              */
int hp3d_set_weight(hp3d_ctx* ctx, const char* tf_var_name, const float* data,
                    const int64_t* shape, int rank);
/* Pack everything set so far for the device (repack HWIO -> MFMA fragment order, permute
 * the concat channels of conv6_1/conv7_1, upload).  Nets whose variables are all present
 * become runnable; a net with only some of its variables is an error.
 * dtype 0: float32 everywhere (exact f32 MFMA).  dtype 1 (BASELINE config 5): the HandSegNet / PoseNet2D
 * filters and activations are float16 on v_mfma_f32_32x32x16_f16 with float32 accumulation, biases and
 * score-map heads; the lifting nets, the mask stage and all outputs stay float32.                   */
int hp3d_finalize_weights(hp3d_ctx* ctx, int dtype);
/* The packed device blob (identical layout on every rank): size, export to / import from a
 * device buffer.  Used for the one-off RCCL broadcast of weights (bench.py, N>1).           */
int hp3d_weights_blob_bytes(hp3d_ctx* ctx, size_t* bytes);
int hp3d_weights_blob_export(hp3d_ctx* ctx, void* dev_dst);
int hp3d_weights_blob_import(hp3d_ctx* ctx, const void* dev_src, int nets_mask);
int hp3d_nets_mask(hp3d_ctx* ctx);   /* bit0 HandSegNet, bit1 PoseNet2D, bit2 PosePrior, bit3 ViewpointNet, bit4 bottleneck, bit5 f16 trunks */

/* ---- whole-path entry points ------------------------------------------------------------
 * hp3d_infer_full   replaces ColorHandPose3DNetwork.inference (nets/ColorHandPose3DNetwork.py:61-99)
 *   image [B,H,W,3] (x/255-0.5 done by the caller, run.py:59), hand_side [B,2] one-hot ->
 *   hand_scoremap [B,H,W,2], image_crop [B,256,256,3], scale_crop [B,1], center [B,2] (row,col),
 *   keypoints_scoremap [B,256,256,21], keypoint_coord3d [B,21,3].  Any H, W >= 16 (the VALID 2x2 max-pools floor odd
 *   extents and the logits are resized from floor(H/8) x floor(W/8) back to H x W, as in the reference).
 * hp3d_infer_2d     replaces .inference2d (:101-129): keypoints_scoremap, image_crop, scale_crop, center.
 * hp3d_handsegnet   replaces .inference_detection (:131-168): scoremap_large [B,H,W,2]
 *   (scoremap_small [B,H/8,W/8,2] is the pre-upsampling map, for staged parity tests).
 * hp3d_posenet2d    replaces .inference_pose2d (:170-219): the 3 scoremaps [B,h/8,w/8,21].
 * hp3d_poseprior    replaces PosePriorNetwork(variant).inference (nets/PosePriorNetwork.py:59-95):
 *   scoremap256 [B,256,256,21] -> coord_xyz_rel_normed [B,21,3], coord3d [B,21,3], R [B,3,3]
 *   (R untouched for direct/bottleneck/local, where the reference returns None).
 * hp3d_pose3d       replaces ._inference_pose3d (:221-247) on a [B,32,32,21] scoremap.
 * hand_mask (extra, may be NULL): the internal objectmap [B,H,W] of single_obj_scoremap
 *   (utils/general.py:233-268), exposed so tests can assert mask equality first.            */
int hp3d_infer_full(hp3d_ctx* ctx, int B, int H, int W, const float* image, const float* hand_side,
                    float* hand_scoremap, float* image_crop, float* scale_crop, float* center,
                    float* keypoints_scoremap, float* keypoint_coord3d, float* hand_mask);
int hp3d_infer_full_dev(hp3d_ctx* ctx, int B, int H, int W, const float* image, const float* hand_side,
                        float* hand_scoremap, float* image_crop, float* scale_crop, float* center,
                        float* keypoints_scoremap, float* keypoint_coord3d, float* hand_mask);
/* The same calls with the host post-processing of the scripts done on the device (run.py:72-73, eval2d.py:93-94,
 * eval2d_gt_cropped.py:76): keypoint_hw_crop [B,21,2] int32 = detect_keypoints(keypoints_scoremap[b])
 * (utils/general.py:331-344: per channel the first maximum, (row, col)) and keypoint_hw [B,21,2] float64 =
 * trafo_coords(keypoint_hw_crop, center, scale_crop, 256) (utils/general.py:347-357).  Computed from the 32x32 maps
 * by re-evaluating tf.image.resize_images' arithmetic, so keypoints_scoremap (5.5 MB / image) may be NULL: eval loops
 * need no heat-map copy.  Bit-exact with the reference functions applied to the up-sampled map, ties included.      */
int hp3d_infer_full_kp(hp3d_ctx* ctx, int B, int H, int W, const float* image, const float* hand_side,
                       float* hand_scoremap, float* image_crop, float* scale_crop, float* center,
                       float* keypoints_scoremap, float* keypoint_coord3d, float* hand_mask,
                       int32_t* keypoint_hw_crop, double* keypoint_hw);
int hp3d_infer_full_kp_dev(hp3d_ctx* ctx, int B, int H, int W, const float* image, const float* hand_side,
                           float* hand_scoremap, float* image_crop, float* scale_crop, float* center,
                           float* keypoints_scoremap, float* keypoint_coord3d, float* hand_mask,
                           int32_t* keypoint_hw_crop, double* keypoint_hw);
int hp3d_infer_full_kp_u8(hp3d_ctx* ctx, int B, int Hin, int Win, const uint8_t* image_u8, int H, int W,
                          const float* hand_side, float* hand_scoremap, float* image_crop, float* scale_crop,
                          float* center, float* keypoints_scoremap, float* keypoint_coord3d, float* hand_mask,
                          int32_t* keypoint_hw_crop, double* keypoint_hw);
/* SURVEY.md 8f N2 -- the step immediately before the hot path, on device: uint8 frames
 * [B,Hin,Win,3] -> x/255-0.5 (data/BinaryDbReader.py:182, run.py:59) -> tf.image.resize_images to H x W
 * (eval_full.py:50, eval2d.py:53; equal sizes = identity) -> hp3d_infer_full.  4x less H2D traffic.      */
int hp3d_infer_full_u8(hp3d_ctx* ctx, int B, int Hin, int Win, const uint8_t* image_u8, int H, int W,
                       const float* hand_side, float* hand_scoremap, float* image_crop, float* scale_crop,
                       float* center, float* keypoints_scoremap, float* keypoint_coord3d, float* hand_mask);
int hp3d_preprocess_u8(hp3d_ctx* ctx, const uint8_t* image_u8, int B, int Hin, int Win, int H, int W, float* out);
int hp3d_infer_2d(hp3d_ctx* ctx, int B, int H, int W, const float* image,
                  float* keypoints_scoremap, float* image_crop, float* scale_crop, float* center);
int hp3d_infer_2d_kp(hp3d_ctx* ctx, int B, int H, int W, const float* image, float* keypoints_scoremap,
                     float* image_crop, float* scale_crop, float* center,
                     int32_t* keypoint_hw_crop, double* keypoint_hw);   /* + detect_keypoints / trafo_coords, as above */
int hp3d_handsegnet(hp3d_ctx* ctx, int B, int H, int W, const float* image,
                    float* scoremap_large, float* scoremap_small);
int hp3d_posenet2d(hp3d_ctx* ctx, int B, int H, int W, const float* image_crop,
                   float* scoremap0, float* scoremap1, float* scoremap2);
int hp3d_posenet2d_dev(hp3d_ctx* ctx, int B, int H, int W, const float* image_crop,
                       float* scoremap0, float* scoremap1, float* scoremap2);
int hp3d_poseprior(hp3d_ctx* ctx, int B, int variant, const float* scoremap256, const float* hand_side,
                   float* coord_xyz_rel_normed, float* coord3d, float* rot_mat);
int hp3d_pose3d(hp3d_ctx* ctx, int B, const float* scoremap32, const float* hand_side,
                float* coord_xyz_rel_normed, float* coord_can, float* rot_mat);

/* ---- per-op entry points (unit/parity tests; same kernels the pipeline runs) --------------
 * hp3d_conv2d          NetworkOps.conv/conv_relu (+ max_pool when pool=1): utils/general.py:36-65
 *                      x [B,H,W,Cin], w HWIO, SAME padding incl. the asymmetric stride-2 case.
 * hp3d_maxpool2        NetworkOps.max_pool, 2x2/2 VALID                       utils/general.py:61-65
 * hp3d_avgpool8        tf.nn.avg_pool 8x8/8                                   nets/PosePriorNetwork.py:61
 * hp3d_resize_bilinear tf.image.resize_images (TF1.3 legacy bilinear)         nets/ColorHandPose3DNetwork.py:97,128,166
 * hp3d_crop_and_resize crop_image_from_xy -> tf.image.crop_and_resize         utils/general.py:163-196
 * hp3d_mask_from_scoremap single_obj_scoremap + calc_center_bb + scale        utils/general.py:233-328, CHP3D.py:82-85
 *                      SIZE LIMIT (the reference has none): the growth runs on bit-packed maps in ONE workgroup's LDS,
 *                      3 * H * (ceil(W / 32) + 1) + 2 words <= 159 KB, e.g. 640x640 or 480x864 (since round 5 each row carries a
 *                      zero guard word: 640x672, accepted before, is refused now); larger maps -> HP3D_ERR_ARG "too large".
 *                      The whole-path entry points (hp3d_infer_full*, hp3d_infer_2d*) have the same limit on the input image.
 *                      -> mask [B,H,W], center [B,2], crop_size [B,1] (before *1.25), scale [B,1], seed int32 [B,2]
 * hp3d_fc              NetworkOps.fully_connected(_relu)                      utils/general.py:112-136
 * hp3d_argmax2d        detect_keypoints (per-channel first arg-max)           utils/general.py:331-344
 *                      x [B,H,W,C] -> int32 [B,C,2] (row, col)                                    */
int hp3d_conv2d(hp3d_ctx* ctx, const float* x, int B, int H, int W, int Cin,
                const float* w_hwio, const float* bias, int k, int stride, int Cout,
                int act, int pool, float* out);
int hp3d_maxpool2(hp3d_ctx* ctx, const float* x, int B, int H, int W, int C, float* out);
int hp3d_avgpool8(hp3d_ctx* ctx, const float* x, int B, int H, int W, int C, float* out);
int hp3d_resize_bilinear(hp3d_ctx* ctx, const float* x, int B, int H, int W, int C,
                         int out_h, int out_w, float* out);
int hp3d_crop_and_resize(hp3d_ctx* ctx, const float* image, int B, int H, int W, int C,
                         const float* center, const float* scale, int crop_size, float* out);
int hp3d_mask_from_scoremap(hp3d_ctx* ctx, const float* scoremap, int B, int H, int W,
                            float* mask, float* center, float* crop_size, float* scale, int32_t* seed);
int hp3d_fc(hp3d_ctx* ctx, const float* x, int B, int Cin, const float* w, const float* bias,
            int Cout, int act, float* out);
int hp3d_argmax2d(hp3d_ctx* ctx, const float* x, int B, int H, int W, int C, int32_t* out_rc);
/* detect_keypoints(tf.image.resize_images(scoremap, (out_h, out_w))[b]) for scoremap [B,h,w,C], h*w <= 4096, without
 * materialising the large map (utils/general.py:331-344 applied to nets/ColorHandPose3DNetwork.py:97): out_rc [B,C,2] */
int hp3d_detect_keypoints(hp3d_ctx* ctx, const float* scoremap, int B, int h, int w, int C, int out_h, int out_w,
                          int32_t* out_rc);

/* ---- measurement ------------------------------------------------------------------------
 * With profiling on (1: last whole-path call only, 2: accumulate over calls until switched off),
 * every launch is bracketed by hipEvents on the ctx stream.  hp3d_prof_get(i): layer name, kernel family, ms, algorithmic FLOPs and
 * algorithmic bytes (input once + weights once + output once).                */
int hp3d_set_profiling(hp3d_ctx* ctx, int on);
int hp3d_prof_count(hp3d_ctx* ctx);
int hp3d_prof_get(hp3d_ctx* ctx, int i, char* name, int name_cap, char* kernel, int kernel_cap,
                  float* ms, double* flops, double* bytes);
/* Per-stage GPU milliseconds of the profiled launches (SURVEY.md 8b `hp3d_get_timing`):
 * [0] HandSegNet, [1] mask / box / crop glue, [2] PoseNet2D (+ heat-map upsample), [3] PosePrior + ViewpointNet +
 * lifting epilogue, [4] everything.  Writes min(n, 5) values; needs hp3d_set_profiling(ctx, 1 | 2).  */
#define HP3D_TIMING_STAGES 5
int hp3d_get_timing(hp3d_ctx* ctx, float* ms_per_stage, int n);
/* Executor counters: "graph_captures" / "graph_replays" = hipGraphs instantiated / launched since hp3d_create (option
 * "graph" = "1"; a replay happens only with per-launch profiling off); "conv_h16_launches" = half-precision trunk
 * layers that ran on conv_h16.hip (option "f16_impl"); "conv_wino2_launches" = float32 layers that ran on conv_wino2.hip (option
 * "wino2"); "conv_wino4_launches" = float32 layers that ran on conv_wino4.hip (option "wino4"),
 * "conv_wino4_tail_launches" = those of them whose last round ran as channel slices (option "wino4_tail");
 * "conv_wino7_launches" = 7x7 layers that ran on conv_wino7.hip (option "wino7"), "conv_wino7_split_launches" = those of them in the channel-split form; "conv_pw2_launches" = 1x1 layer pairs that ran as one launch (option "pw2");
 * "first_touch_launches" = read passes in front of conv1_1 (option "first_touch");
 * "conv_first_launches" = conv1_1-shaped layers (3x3, 3 -> 64) that ran on conv_first.hip;
 * "lift_overlap_calls" = lifting stages that ran their two towers on two streams (option "lift_overlap");
 * "lift_fused_launches" = lifting stages that ran as the one fused launch (option "lift_fused"); "comm_ranks" = ranks of the live RCCL communicator as RCCL itself
 * reports them (ncclCommCount), 0 without one -- bench.py prints it so that a multi-GPU line proves its own world size. */
int hp3d_get_counter(hp3d_ctx* ctx, const char* name, long long* value);

/* ---- multi-GPU: one process and one context per GPU, RCCL over xGMI ---------
 * The path has no data-path collective.  These are the two exchanges it needs, on the context's own stream:
 * the one-off broadcast of the packed weight blob (replaces every rank un-pickling + re-packing 140 MB) and the
 * all-gather of per-shard results (keypoints: 252 B per image).  librccl is loaded on first use (dlopen), not
 * linked.  The 128-byte id is produced on one rank and handed to the others by the launcher (torch.distributed
 * store, MPI, a file ...); `hp3d_bcast_weights` on a non-root rank replaces that rank's weights (any nets mask /
 * precision the root finalized).                                                                  */
#define HP3D_COMM_ID_BYTES 128
int hp3d_comm_unique_id(void* id128);
int hp3d_comm_init(hp3d_ctx* ctx, int rank, int nranks, const void* id128);
int hp3d_bcast_weights(hp3d_ctx* ctx, int root);
int hp3d_allgather(hp3d_ctx* ctx, const float* send_host, int count, float* recv_host /* [nranks * count] */);
int hp3d_allgather_dev(hp3d_ctx* ctx, const float* send_dev, int count, float* recv_host /* [nranks * count] */);
int hp3d_comm_destroy(hp3d_ctx* ctx);

/* ---- memory for callers that keep their batches in HBM (the `_dev` entry points; bench.py, hand3d_amd/dist.py) ---
 * Replaces what the reference left to TensorFlow's feed_dict / allocator (run.py:61-64): device buffers, pinned host
 * buffers, blocking copies ordered on the context's stream (kind 0 = host->device, 1 = device->host, 2 = device->
 * device), and an upload on a SECOND stream so that the host->device copy of batch n+1 runs under the kernels of
 * batch n (hp3d_wait_upload makes the compute stream wait for the last hp3d_upload_async).  No PyTorch needed.     */
int hp3d_dev_alloc(hp3d_ctx* ctx, size_t bytes, void** out);
int hp3d_dev_free(hp3d_ctx* ctx, void* p);
int hp3d_host_alloc(hp3d_ctx* ctx, size_t bytes, void** out);      /* page-locked */
int hp3d_host_free(hp3d_ctx* ctx, void* p);
int hp3d_memcpy(hp3d_ctx* ctx, void* dst, const void* src, size_t bytes, int kind);
int hp3d_upload_async(hp3d_ctx* ctx, void* dst_dev, const void* src_pinned_host, size_t bytes);
int hp3d_wait_upload(hp3d_ctx* ctx);

/* ---- host utility ----------------------------------------------------------------------------
 * CRC-32C (Castagnoli) of a host buffer: the checksum TensorFlow checkpoints carry (hand3d_amd/utils/tf_checkpoint.py
 * verifies 140 MB of tensors with it; the pure-Python loop manages ~5 MB/s).  No device involved.          */
uint32_t hp3d_crc32c(const void* data, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* HP3D_H */

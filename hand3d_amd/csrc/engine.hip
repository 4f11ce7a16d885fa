// engine.hip -- context, weight packing, stream-ordered executor and the C-ABI of libhp3d.so.
// Host code only (the kernels live in conv_mfma.hip / glue.hip).  See include/hp3d.h for the
// reference interfaces each entry point replaces.
#include "hp3d_common.h"
#include "lift_fused.h"
#include "../../include/hp3d.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <chrono>
#include <atomic>
#include <functional>
#include <string>
#include <vector>
#ifndef HP3D_EMU
#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only: the library is dlopen'ed on first use (573 MB), never linked
#endif

namespace {

std::string g_last_error;

#define HP3D_FAIL(ctx, code, ...)                              \
    do {                                                       \
        char _b[512];                                          \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                 \
        set_error((ctx), _b);                                  \
        return (code);                                         \
    } while (0)
#define HIPCHK(ctx, expr)                                                                          \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) HP3D_FAIL(ctx, HP3D_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(_e)); \
    } while (0)
#define CHK(expr)                     \
    do {                              \
        int _r = (expr);              \
        if (_r != 0) return _r;       \
    } while (0)

enum { NET_SEG = 1, NET_POSE = 2, NET_PRIOR = 4, NET_VP = 8, NET_BOTTLENECK = 16 };

// ---- layer tables (mirror hand3d_amd/arch.py; nets/ColorHandPose3DNetwork.py:144-161,183-214,255-267,291-307)
struct ConvL {
    std::string name;      // tf scope/layer, e.g. "HandSegNet/conv1_2"
    int k, cin, cout, stride, relu;
    int mode;              // 0 plain, 1 im2col first layer (Cin=3), 2 concat-permuted (conv6_1/conv7_1)
    // packed geometry
    int ek, cin_pad, cout_pad;
    size_t w_off, b_off;   // offsets into the blob (floats)
    size_t ww_off = 0;     // Winograd-transformed filters U[16][cin_pad][cout_pad] (3x3/s1, cout%64==0), 0 = none
    size_t ww2_off = 0;    // the same filters in conv_wino2.hip's fragment order (two workgroups per CU), 0 = none
    size_t ww4_off = 0;    // F(4x4,3x3) filters U[36][...] in conv_wino4.hip's fragment order (trunk nets), 0 = none
    size_t ww4s_off = 0;   // the F(4x4,3x3) filters split into three bfloat16 pieces in conv_wino4s.hip's fragment order (3x3 trunk layers with Cin >= 128), 0 = none
    size_t ww7_off = 0;    // 7x7 layers: F(4x4,4x4) filters of the four 4x4-tap blocks in conv_wino7.hip's order [chunk][169][Cout/16][q][n][e], 0 = none
    size_t raw_off = 0;    // lifting nets only: [Cout/64][cin4][tap][64] for lift_fused.hip (one contiguous weight stream per wave), 0 = none
    size_t hwio_off = 0;   // lifting nets, the last stride-2 layer of a tower (8x8 -> 4x4 map): the plain HWIO filter = the matrix of conv_s2_gemm_launch, 0 = none
    int cin4 = 0;
    int cin_pad16 = 0;     // f16 mode: input channels padded to 64 halves (one 128-B chunk)
    size_t w16_off = 0;    // offset into the f16 blob (halves); trunk nets only
    int net;
};
struct FcL {
    std::string name;
    int cin, cout, relu;
    size_t w_off, b_off;
    int net;
};

int pad32(int c) { return (c + 31) / 32 * 32; }

struct Tables {
    std::vector<ConvL> conv;
    std::vector<FcL> fc;
    std::map<std::string, int> conv_idx, fc_idx;
    size_t blob_floats = 0, blob16_halves = 0;

    void add_conv(const char* scope, const char* name, int k, int cin, int cout, int stride, int relu, int net) {
        ConvL l;
        l.name = std::string(scope) + "/" + name;
        l.k = k; l.cin = cin; l.cout = cout; l.stride = stride; l.relu = relu; l.net = net;
        l.mode = (cin == 3) ? 1 : (cin == 149) ? 2 : 0;
        l.ek = (l.mode == 1) ? 1 : k;
        l.cin_pad = (l.mode == 1) ? 32 : (l.mode == 2) ? 160 : pad32(cin);
        l.cout_pad = pad32(cout);
        l.w_off = blob_floats;
        blob_floats += (size_t)l.ek * l.ek * l.cin_pad * l.cout_pad;
        l.b_off = blob_floats;
        blob_floats += l.cout_pad;
        // Winograd F(2x2,3x3) copy (16 planes; a 7x7 filter as nine 3x3 blocks along the channel axis)
        if (stride == 1 && cout % 64 == 0 && ((k == 3 && l.mode == 0) || (k == 7 && (l.mode == 0 || l.mode == 2)))) {
            l.ww_off = blob_floats;
            blob_floats += wino_packed_floats(k, l.cin_pad, l.cout_pad);
            l.ww2_off = blob_floats;
            blob_floats += wino_packed_floats(k, l.cin_pad, l.cout_pad);
            if (net == NET_SEG || net == NET_POSE) {
                l.ww4_off = blob_floats;
                blob_floats += wino4_packed_floats(k, l.cin_pad, l.cout_pad);
                if (k == 7) {
                    l.ww7_off = blob_floats;
                    blob_floats += wino7_packed_floats(l.cin_pad, l.cout_pad);
                }
                if (k == 3 && l.mode == 0 && l.cin_pad >= 128 && l.cin_pad % 16 == 0) {
                    blob_floats = (blob_floats + 3) / 4 * 4;
                    l.ww4s_off = blob_floats;
                    blob_floats += (wino4s_packed_bytes(l.cin_pad, l.cout_pad) + 15) / 16 * 4;
                }
            }
        }
        if (net == NET_PRIOR || net == NET_VP) {
            l.cin4 = (cin + 15) / 16 * 16;          // four waves x whole channel quads
            l.raw_off = blob_floats;
            blob_floats += (size_t)9 * l.cin4 * ((cout + 63) / 64 * 64);
        }
        if (net == NET_VP && stride == 2 && k == 3 && cin >= 256) {          // (ViewpointNet/conv_vp_2_2: 47 -> 39 us; PosePrior's 128-channel twin measured slower this way)
            blob_floats = (blob_floats + 3) / 4 * 4;
            l.hwio_off = blob_floats;
            blob_floats += (size_t)9 * cin * cout;
        }
        if (net == NET_SEG || net == NET_POSE) {     // half-precision copy for hp3d_finalize_weights(dtype=1)
            l.cin_pad16 = (l.mode == 1) ? 64 : (l.mode == 2) ? 192 : (cin + 63) / 64 * 64;
            l.w16_off = blob16_halves;
            blob16_halves += (size_t)l.ek * l.ek * l.cin_pad16 * l.cout_pad;
        }
        conv_idx[l.name] = (int)conv.size();
        conv.push_back(l);
    }
    void add_fc(const char* scope, const char* name, int cin, int cout, int relu, int net) {
        FcL l;
        l.name = std::string(scope) + "/" + name;
        l.cin = cin; l.cout = cout; l.relu = relu; l.net = net;
        l.w_off = blob_floats;
        blob_floats += ((size_t)cin * cout + 3) / 4 * 4;
        l.b_off = blob_floats;
        blob_floats += (cout + 3) / 4 * 4;
        fc_idx[l.name] = (int)fc.size();
        fc.push_back(l);
    }
    Tables() {
        char nm[32];
        {   // HandSegNet
            const int nl[4] = {2, 2, 4, 4}, ch[4] = {64, 128, 256, 512};
            int cin = 3;
            for (int b = 0; b < 4; ++b)
                for (int i = 0; i < nl[b]; ++i) {
                    snprintf(nm, sizeof nm, "conv%d_%d", b + 1, i + 1);
                    add_conv("HandSegNet", nm, 3, cin, ch[b], 1, 1, NET_SEG);
                    cin = ch[b];
                }
            add_conv("HandSegNet", "conv5_1", 3, 512, 512, 1, 1, NET_SEG);
            add_conv("HandSegNet", "conv5_2", 3, 512, 128, 1, 1, NET_SEG);
            add_conv("HandSegNet", "conv6_1", 1, 128, 512, 1, 1, NET_SEG);
            add_conv("HandSegNet", "conv6_2", 1, 512, 2, 1, 0, NET_SEG);
        }
        {   // PoseNet2D
            const int nl[4] = {2, 2, 4, 2}, ch[4] = {64, 128, 256, 512};
            int cin = 3;
            for (int b = 0; b < 4; ++b)
                for (int i = 0; i < nl[b]; ++i) {
                    snprintf(nm, sizeof nm, "conv%d_%d", b + 1, i + 1);
                    add_conv("PoseNet2D", nm, 3, cin, ch[b], 1, 1, NET_POSE);
                    cin = ch[b];
                }
            add_conv("PoseNet2D", "conv4_3", 3, 512, 256, 1, 1, NET_POSE);
            add_conv("PoseNet2D", "conv4_4", 3, 256, 256, 1, 1, NET_POSE);
            add_conv("PoseNet2D", "conv4_5", 3, 256, 256, 1, 1, NET_POSE);
            add_conv("PoseNet2D", "conv4_6", 3, 256, 256, 1, 1, NET_POSE);
            add_conv("PoseNet2D", "conv4_7", 3, 256, 128, 1, 1, NET_POSE);
            add_conv("PoseNet2D", "conv5_1", 1, 128, 512, 1, 1, NET_POSE);
            add_conv("PoseNet2D", "conv5_2", 1, 512, 21, 1, 0, NET_POSE);
            for (int p = 6; p <= 7; ++p) {
                int c = 149;
                for (int r = 1; r <= 5; ++r) {
                    snprintf(nm, sizeof nm, "conv%d_%d", p, r);
                    add_conv("PoseNet2D", nm, 7, c, 128, 1, 1, NET_POSE);
                    c = 128;
                }
                snprintf(nm, sizeof nm, "conv%d_6", p);
                add_conv("PoseNet2D", nm, 1, 128, 128, 1, 1, NET_POSE);
                snprintf(nm, sizeof nm, "conv%d_7", p);
                add_conv("PoseNet2D", nm, 1, 128, 21, 1, 0, NET_POSE);
            }
        }
        {   // PosePrior
            const int ch[3] = {32, 64, 128};
            int cin = 21;
            for (int i = 0; i < 3; ++i) {
                snprintf(nm, sizeof nm, "conv_pose_%d_1", i);
                add_conv("PosePrior", nm, 3, cin, ch[i], 1, 1, NET_PRIOR);
                snprintf(nm, sizeof nm, "conv_pose_%d_2", i);
                add_conv("PosePrior", nm, 3, ch[i], ch[i], 2, 1, NET_PRIOR);
                cin = ch[i];
            }
            add_fc("PosePrior", "fc_rel0", 2050, 512, 1, NET_PRIOR);
            add_fc("PosePrior", "fc_rel1", 512, 512, 1, NET_PRIOR);
            add_fc("PosePrior", "fc_xyz", 512, 63, 0, NET_PRIOR);
            add_fc("PosePrior", "fc_bottleneck", 512, 30, 0, NET_BOTTLENECK);
            add_fc("PosePrior", "fc_xyz#bn", 30, 63, 0, NET_BOTTLENECK);   // fc_xyz of the bottleneck variant
        }
        {   // ViewpointNet
            const int ch[3] = {64, 128, 256};
            int cin = 21;
            for (int i = 0; i < 3; ++i) {
                snprintf(nm, sizeof nm, "conv_vp_%d_1", i);
                add_conv("ViewpointNet", nm, 3, cin, ch[i], 1, 1, NET_VP);
                snprintf(nm, sizeof nm, "conv_vp_%d_2", i);
                add_conv("ViewpointNet", nm, 3, ch[i], ch[i], 2, 1, NET_VP);
                cin = ch[i];
            }
            add_fc("ViewpointNet", "fc_vp0", 4098, 256, 1, NET_VP);
            add_fc("ViewpointNet", "fc_vp1", 256, 128, 1, NET_VP);
            // fc_vp_ux / uy / uz ([128,1] each) are packed as one [128,3] matrix
            add_fc("ViewpointNet", "fc_vp_u", 128, 3, 0, NET_VP);
        }
    }
};

struct HostVar {
    std::vector<float> data;
    std::vector<int64_t> shape;
};

struct ProfRec {
    std::string name, kernel;
    double flops, bytes;
    hipEvent_t e0, e1;
};

}  // namespace

struct hp3d_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    Tables T;
    std::map<std::string, HostVar> vars;
    float* blob = nullptr;     // device, T.blob_floats
    hp3d_f16* blob16 = nullptr; // device, T.blob16_halves (dtype 1 only)
    int prec = 0;              // 0: f32 everywhere; 1: HandSegNet / PoseNet2D trunks in f16 (f32 accumulate)
    hp3d_f16* d_concat16 = nullptr;
    size_t concat16_px = 0;
    int nets = 0;              // finalized nets mask
    int empty_fltmax = 0;
    int conv_naive = 0;
    int use_wino = 1;          // 3x3/s1 layers with Cout%64==0 run as Winograd F(2x2,3x3) (conv_impl=direct disables)
    // debug copies of the unpacked HWIO weights for conv_impl=naive
    std::map<std::string, float*> naive_w;

    // arena
    size_t act_floats = 0;     // capacity of bufA/bufB
    float *bufA = nullptr, *bufB = nullptr, *col = nullptr;
    size_t col_floats = 0;
    int capB = 0;
    int sideB = 0;             // batch the FC buffers hold when this context is only the side tower of an overlapped lifting stage (ensure_side_tower)
    float *d_image = nullptr, *d_hs = nullptr, *d_large = nullptr, *d_crop = nullptr, *d_center = nullptr,
          *d_scale = nullptr, *d_cropsize = nullptr, *d_kpmap = nullptr, *d_coord = nullptr, *d_mask = nullptr,
          *d_segsmall = nullptr, *d_concat = nullptr, *d_sm[3] = {nullptr, nullptr, nullptr}, *d_can = nullptr,
          *d_rot = nullptr, *d_u = nullptr, *d_fc1 = nullptr, *d_fc2 = nullptr, *d_fg = nullptr,
          *d_pooled = nullptr, *d_fcpart = nullptr;
    hipStream_t copy_stream = nullptr;   // hp3d_upload_async: H2D of the next batch under the current batch's kernels
    hipEvent_t upload_done = nullptr;
    double* d_kpimg = nullptr;   // [B,21,2] float64 (trafo_coords of the detected keypoints)
    int* d_kpcrop = nullptr;     // [B,21,2] int32 (detect_keypoints: row, col in the crop)
    void* comm = nullptr;        // ncclComm_t (hp3d_comm_init)
    float* d_gather = nullptr;   // reusable send + receive staging of hp3d_allgather[_dev] (no allocation per step)
    size_t gather_floats = 0;
    int comm_rank = 0, comm_size = 1;
    int* d_seed = nullptr;
    unsigned long long* d_keys = nullptr;
    unsigned char* d_det = nullptr;
    unsigned char* d_u8 = nullptr;
    size_t u8_bytes = 0;
    size_t image_floats = 0, large_floats = 0, det_bytes = 0, pose_px = 0;

    // profiling
    int profiling = 0;
#ifndef HP3D_EMU
    struct GraphEntry { hipGraphExec_t exec = nullptr; long epoch = -1; int calls = 0; };
    std::map<std::string, GraphEntry> graphs;    // hp3d_set_option("graph", "1"): replayed whole-call launch sequences
#endif
    long conv_first_launches = 0;
    int first_touch_beside = 1;     // option "first_touch_beside": the read pass on the child context's stream beside conv1_1 (1) | in front of it (0)
    int first_touch = -1;      // option "first_touch": stream conv1_1's input image through the memory-side cache right before the launch: -1 auto
                               // (a cold image of 8 ... 128 MB), 0 never, 1 always
    bool trunk_input_hot = false;   // set by run_trunk: the trunk's input was written by the kernel in front of it (crop, uint8 front end)
    long first_touch_launches = 0;
    int first_balanced = 1;    // option "first_walk": conv_first.hip's workgroups walk balanced runs of tiles ("balanced") | whole tile rows ("rows": rounds 2-4)
    int use_first = 1;         // conv1_1 on its own kernel (conv_first.hip); conv_impl=direct keeps it on the general one
    int nstreams = -1;         // whole-path calls: halves of the batch on two HIP streams (option "streams"; -1 auto)
    hp3d_ctx* kid = nullptr;   // the second stream's context: own stream + arena, SHARES this context's weight blob
    bool shared_weights = false;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int use_h16 = 1;           // half-precision 3x3 trunk layers on conv_h16.hip (option "f16_impl" = "h16" | "mfma")
    int h16_k7k1 = 1;          // ... and the 7x7 / 1x1 layers with >= 64 couts on its single-buffer forms (option "f16_k7k1" = 0 | 1)
    int wino_splitk = 1;       // Winograd layers that under-fill the chip split their channel steps (option "wino_splitk")
    int lift_overlap = 1;      // option "lift_overlap": the two towers of the unfused lifting stage on two streams (ViewpointNet on the child context's)
    long lift_overlap_calls = 0;
    int use_lift_fused = -1;   // the lifting stage as one launch (lift_fused.hip): -1 auto (B <= 4), 0 never, 1 always (option "lift_fused")
    unsigned* d_liftbar = nullptr;
    unsigned* h_lifterr = nullptr;     // mapped host word: lift_fused.hip's grid barrier timed out (results of that launch are invalid)
    long lift_fused_launches = 0;
    bool two_streams_live = false;   // set while a whole-path call runs its two halves on two streams (kernel choice: wino2_auto)
    int use_wino2 = -1;        // conv_wino2.hip (two workgroups per CU): -1 auto (short reductions, under-filled launches), 0 never, 1 wherever eligible (option "wino2")
    int use_graph = 0;
    long graph_captures = 0, graph_replays = 0;     // hp3d_get_counter: did the hipGraph path really run?
    int fuse12 = 1;            // half-precision trunks: conv1_1 computed inside conv1_2's patch stage (option "f16_fuse12": 0 | 1 = form by size |
                               // 2 = "ring": two workgroups per CU, filter ring | 3 = "resident": one per CU, conv1_2's filters in registers)
    long conv_h16_first_resident_launches = 0;
    long conv_h16_launches = 0;                     // hp3d_get_counter: layers that went to conv_h16.hip (the child context counts its own)
    int use_wino4 = -2;        // conv_wino4.hip (Winograd F(4x4,3x3)), option "wino4": -2 auto (both trunks by cost model), -1 "pose" (PoseNet2D only, by cost
                               // model), 0 never, 1 wherever eligible (tests)
    int use_wino4s = 0;        // conv_wino4s.hip (F(4x4,3x3) on the bf16 matrix pipe, split operands), option "wino4_split": 0 never, -1 "auto" (the filled
                               // 3x3 launches with Cin >= 128 that conv_wino4.hip would take), 1 wherever eligible (tests)
    long conv_wino4s_launches = 0;
    long conv_wino4s_tail_launches = 0;
    int kp_up_side = 1;        // the heat-map up-sampling behind ViewpointNet on the child stream (1) or behind PosePrior on this one (0): option "kp_up_side"
    int fc_tail = 1;           // the tail of a lifting tower (reduce of the first FC layer + the two small FC layers) as one launch (option "fc_tail")
    int tiny_gemm = 1;         // the towers' last stride-2 layer (8x8 -> 4x4) as split-K GEMM over its output pixels (option "tiny_gemm")
    long fc_tail_launches = 0, conv_s2_gemm_launches = 0;
    int w4_tail = 1;           // conv_wino4.hip: cut an under-filled last round of items into channel slices (option "wino4_tail")
    int use_pw2 = 1;           // conv_pw2.hip: the 1x1 head pairs (conv6_1 + conv6_2, conv5_1 + conv5_2, conv6_6 + conv6_7, conv7_6 + conv7_7) as one launch each
                               // (option "pw2": 0 never, 1 when the launch has a workgroup per CU, 2 = "force": whenever the shapes allow, tests)
    long conv_pw2_launches = 0;
    int use_wino7 = -1;        // conv_wino7.hip (the 7x7 layers as Winograd F(4x4,4x4)), option "wino7": -1 auto (launches that fill the chip), 0 never, 1 wherever eligible
    long conv_wino7_launches = 0;
    int wino7_ksplit = 0;      // option "wino7_ksplit": 0 = auto (conv_wino7_eligible's choice for under-filled launches), N = that many channel splits (tests, tuning)
    long conv_wino7_split_launches = 0;
    long conv_wino4_tail_launches = 0;
    long conv_wino4_launches = 0;                   // hp3d_get_counter: layers that went to conv_wino4.hip
    long conv_wino2_launches = 0;                   // hp3d_get_counter: layers that went to conv_wino2.hip
    long graph_epoch = 0;      // bumped by anything a captured sequence depends on (allocations, weights, options)
    int micro_batch = -1;      // whole-path calls run in chunks of at most this many images (0: never split; -1 auto:
                               // 32 in float32 mode, no split with half-precision trunks -- measured optima)
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> event_pool;
    size_t event_next = 0;
};

namespace {

void set_error(hp3d_ctx* ctx, const char* msg) {
    g_last_error = msg;
    if (ctx) ctx->err = msg;
}

template <typename T>
int dev_realloc(hp3d_ctx* ctx, T** p, size_t count) {
    ++ctx->graph_epoch;            // captured launch sequences hold the old address
    if (*p) hipFree(*p);
    *p = nullptr;
    HIPCHK(ctx, hipMalloc((void**)p, count * sizeof(T)));
    return 0;
}

hipEvent_t get_event(hp3d_ctx* ctx) {
    if (ctx->event_next == ctx->event_pool.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        ctx->event_pool.push_back(e);
    }
    return ctx->event_pool[ctx->event_next++];
}

struct ProfScope {
    hp3d_ctx* ctx;
    int idx = -1;
    ProfScope(hp3d_ctx* c, const std::string& name, const char* kernel, double flops, double bytes) : ctx(c) {
        if (!c->profiling) return;
        ProfRec r{name, kernel, flops, bytes, get_event(c), get_event(c)};
        hipEventRecord(r.e0, c->stream);
        idx = (int)c->prof.size();
        c->prof.push_back(r);
    }
    ~ProfScope() {
        if (idx >= 0) hipEventRecord(ctx->prof[idx].e1, ctx->stream);
    }
};

void prof_reset(hp3d_ctx* ctx) {
    ctx->prof.clear();
    ctx->event_next = 0;
}

// ---- weight packing ------------------------------------------------------------------------
// wpk[tap][c8][co32][h][n][j] = W_hwio[tap][ref_channel(c8*8+h*4+j)][co32*32+n]   (conv_mfma.hip)
void pack_conv(const ConvL& l, const float* w, const float* b, float* blob) {
    float* wp = blob + l.w_off;
    float* bp = blob + l.b_off;
    const int taps = l.ek * l.ek, C8 = l.cin_pad / 8, CO32 = l.cout_pad / 32;
    memset(wp, 0, sizeof(float) * (size_t)taps * l.cin_pad * l.cout_pad);
    memset(bp, 0, sizeof(float) * l.cout_pad);
    for (int co = 0; co < l.cout; ++co) bp[co] = b[co];
    for (int tap = 0; tap < taps; ++tap)
        for (int e = 0; e < l.cin_pad; ++e) {
            // engine channel e -> (reference tap, reference channel)
            int rtap = tap, rc = -1;
            if (l.mode == 0) {
                rc = (e < l.cin) ? e : -1;
            } else if (l.mode == 1) {            // im2col: e = (r*3+s)*3 + c
                if (e < 27) { rtap = e / 3; rc = e % 3; }
            } else {                             // concat buffer: [encoding 0..127 | scoremap 128..148 | 0]
                if (e < 128) rc = 21 + e;        // reference concat is [scoremap(21), encoding(128)]
                else if (e < 149) rc = e - 128;
            }
            if (rc < 0) continue;
            const int c8 = e >> 3, h = (e >> 2) & 1, j = e & 3;
            const float* src = w + ((size_t)rtap * l.cin + rc) * l.cout;
            for (int co = 0; co < l.cout; ++co) {
                const int co32 = co >> 5, n = co & 31;
                wp[((((size_t)tap * C8 + c8) * CO32 + co32) * 2 + h) * 128 + n * 4 + j] = src[co];
            }
        }
}

// half-precision packing: wpk16[tap][Cin/16][Cout/32][h][n][8] = W[tap][ref_channel(16*kb + 8*h + e)][32*co32 + n]
void pack_conv16(const ConvL& l, const float* w, hp3d_f16* blob16) {
    hp3d_f16* wp = blob16 + l.w16_off;
    const int taps = l.ek * l.ek, KB = l.cin_pad16 / 16, CO32 = l.cout_pad / 32;
    for (size_t i = 0; i < (size_t)taps * l.cin_pad16 * l.cout_pad; ++i) wp[i] = (hp3d_f16)0.f;
    for (int tap = 0; tap < taps; ++tap)
        for (int e16 = 0; e16 < l.cin_pad16; ++e16) {
            int rtap = tap, rc = -1;
            if (l.mode == 0) rc = (e16 < l.cin) ? e16 : -1;
            else if (l.mode == 1) { if (e16 < 27) { rtap = e16 / 3; rc = e16 % 3; } }
            else { if (e16 < 128) rc = 21 + e16; else if (e16 < 149) rc = e16 - 128; }
            if (rc < 0) continue;
            const int kb = e16 >> 4, h = (e16 >> 3) & 1, e = e16 & 7;
            const float* src = w + ((size_t)rtap * l.cin + rc) * l.cout;
            for (int co = 0; co < l.cout; ++co)
                wp[((((size_t)tap * KB + kb) * CO32 + (co >> 5)) * 2 + h) * 256 + (co & 31) * 8 + e] = (hp3d_f16)src[co];
        }
}

int find_var(hp3d_ctx* ctx, const std::string& name, const HostVar** out) {
    auto it = ctx->vars.find(name);
    if (it == ctx->vars.end()) return -1;
    *out = &it->second;
    return 0;
}

// ---- arena ---------------------------------------------------------------------------------
int ensure_arena(hp3d_ctx* ctx, int B, int H, int W) {
    const size_t px = (size_t)B * std::max((size_t)H * W, (size_t)256 * 256);
    const size_t act = px * 64;
    if (act > ctx->act_floats) {
        CHK(dev_realloc(ctx, &ctx->bufA, act));
        CHK(dev_realloc(ctx, &ctx->bufB, act));
        ctx->act_floats = act;
    }
    const size_t imgf = (size_t)B * H * W * 3;
    if (imgf > ctx->image_floats) {
        CHK(dev_realloc(ctx, &ctx->d_image, imgf));
        ctx->image_floats = imgf;
    }
    const size_t largef = (size_t)B * H * W * 2;
    if (largef > ctx->large_floats) {
        CHK(dev_realloc(ctx, &ctx->d_large, largef));
        CHK(dev_realloc(ctx, &ctx->d_mask, (size_t)B * H * W));
        CHK(dev_realloc(ctx, &ctx->d_fg, (size_t)B * H * W));
        CHK(dev_realloc(ctx, &ctx->d_segsmall, (size_t)B * (H / 8 + 1) * (W / 8 + 1) * 32));
        ctx->large_floats = largef;
    }
    if ((size_t)B * H * W > ctx->det_bytes) {
        CHK(dev_realloc(ctx, &ctx->d_det, (size_t)B * H * W));
        ctx->det_bytes = (size_t)B * H * W;
    }
    if (B > ctx->capB) {
        CHK(dev_realloc(ctx, &ctx->d_hs, (size_t)B * 2));
        CHK(dev_realloc(ctx, &ctx->d_crop, (size_t)B * 256 * 256 * 3));
        CHK(dev_realloc(ctx, &ctx->d_center, (size_t)B * 2));
        CHK(dev_realloc(ctx, &ctx->d_scale, (size_t)B));
        CHK(dev_realloc(ctx, &ctx->d_cropsize, (size_t)B));
        CHK(dev_realloc(ctx, &ctx->d_kpmap, (size_t)B * 256 * 256 * 21));
        CHK(dev_realloc(ctx, &ctx->d_pooled, (size_t)B * 32 * 32 * 32));
        CHK(dev_realloc(ctx, &ctx->d_coord, (size_t)B * 63));
        CHK(dev_realloc(ctx, &ctx->d_kpimg, (size_t)B * 42));
        CHK(dev_realloc(ctx, &ctx->d_kpcrop, (size_t)B * 42));
        CHK(dev_realloc(ctx, &ctx->d_can, (size_t)B * 63));
        CHK(dev_realloc(ctx, &ctx->d_rot, (size_t)B * 9));
        CHK(dev_realloc(ctx, &ctx->d_u, (size_t)B * 4));
        CHK(dev_realloc(ctx, &ctx->d_fc1, (size_t)B * 512));
        CHK(dev_realloc(ctx, &ctx->d_fc2, (size_t)B * 512));
        CHK(dev_realloc(ctx, &ctx->d_fcpart, std::max((size_t)B * 17 * 512 + (size_t)B * 33 * 256, (size_t)B * 16 * 18 * 256)));      // (FC slices; conv_s2_gemm: 18 slices x 16 B rows x 256)
        CHK(dev_realloc(ctx, &ctx->d_seed, (size_t)B * 2));
        CHK(dev_realloc(ctx, &ctx->d_keys, (size_t)B));
        ctx->capB = B;
        ctx->sideB = B;
    }
    return 0;
}

// concat / scoremap buffers depend on the PoseNet input size (h/8 x w/8)
int ensure_pose_bufs(hp3d_ctx* ctx, int B, int hs, int ws) {
    const size_t px = (size_t)B * hs * ws;
    if (px > ctx->pose_px || !ctx->d_concat) {
        CHK(dev_realloc(ctx, &ctx->d_concat, px * 160));
        for (int i = 0; i < 3; ++i) CHK(dev_realloc(ctx, &ctx->d_sm[i], px * 32));
        CHK(dev_realloc(ctx, &ctx->d_concat16, px * 192));
        HIPCHK(ctx, hipMemsetAsync(ctx->d_concat16, 0, px * 192 * sizeof(hp3d_f16), ctx->stream));
        ctx->pose_px = px;
    }
    return 0;
}

void same_pad(int in, int k, int stride, int* out, int* before) {
    *out = (in + stride - 1) / stride;
    int total = (*out - 1) * stride + k - in;
    if (total < 0) total = 0;
    *before = total / 2;
}

// Which Winograd kernel a layer takes when both can run it (option "wino2" = "auto"): a small cost model in units of one CU-time of
// conv_wino.hip's item-step (32 tiles x 128 couts x 32 channels), fitted to per-layer timings of both kernels at B = 1 ... 32
// (profiles/r03_tuning_notes.md).  conv_wino.hip walks ceil(items / CUs) rounds of whole items; conv_wino2.hip's items are a quarter
// of that work each, two run per CU, so its time is the balanced share plus one small item of tail -- it wins where the coarse rounds
// quantise badly (300 items on 256 CUs) and at small batches, and loses ~7 % on long filled launches (twice the window traffic and
// input-transform work per MFMA).  `old_nt` / `old_ks`: conv_wino_eligible's answer for the same layer (0: it would not run).
bool wino2_auto(int k, int cin_pad, int cout_pad, int Ho, int Wo, int B, int old_nt, int old_ks, int ks2, bool two_streams) {
    if ((long)B * Ho * Wo < 512) return false;              // 16 x 16 maps and below: one direct launch beats split + reduce
    if (!old_nt) return Ho * Wo >= 900 && cin_pad >= 64;      // trunk layers conv_wino.hip cannot fill; the lifting nets' 16 x 16 / 8 x 8 maps stay direct
    const double cus = hp3d_num_cus(), reduce_cost = 0.7, c2 = 1.07;
    const long tiles = (long)B * ((Ho + 1) / 2) * ((Wo + 1) / 2);
    const int nsub = k == 7 ? 9 : 1;
    const long items1 = (tiles + old_nt - 1) / old_nt * (cout_pad / (old_nt == 32 ? 128 : 64));
    const int S1 = nsub * cin_pad / (old_nt == 32 ? 32 : 16);
    const double w1 = old_nt == 32 ? 1.0 : 0.5;
    const double t_old = old_ks > 1 ? std::ceil((double)S1 / old_ks) * w1 * std::max(1.0, items1 * old_ks / cus) + reduce_cost
                                    : std::ceil(items1 / cus) * S1 * w1;
    const long items2 = (tiles + 31) / 32 * (cout_pad / 64);
    const int S2 = nsub * cin_pad / 16;
    const double t_new = ks2 > 1 ? c2 * (std::ceil((double)S2 / ks2) * 0.25 * std::max(1.0, items2 * ks2 / cus)) + reduce_cost
                                 : c2 * (items2 * S2 * 0.25 / cus + S2 * 0.125);
    // with a second stream on the GPU the other half's kernels fill conv_wino.hip's tail rounds anyway (measured: B = 16 as 8 + 8
    // loses 2 % when its filled layers switch): only under-filled launches are candidates then
    if (two_streams && old_ks <= 1) return false;
    return t_new < 0.97 * t_old;
}

// Which layers take the F(4x4,3x3) kernel (option "wino4" = "auto": both trunks; "pose": PoseNet2D only).  Numerics: end to end the larger
// Winograd tile moves PoseNet2D's heat-maps by 5e-6 (gate 1e-3) and the 3-D keypoints by 3e-6 (gate 1e-4).  HandSegNet's score map
// feeds a THRESHOLD (the hand mask), where any change of summation order flips pixels whose two logits are equal to rounding: against
// the all-direct-kernel run, 24 of 512 synthetic images differ in at least one mask pixel with F(2x2,3x3) and 35 with F(4x4,3x3); the
// crop box and the keypoints (1e-4) agree on all 512 either way (scripts/mask_flip_rate.py, profiles/r03_tuning_notes.md section 7).
// Speed: the cost model below, fitted to per-layer timings of the three Winograd kernels at B = 1 ... 32.
// what option "wino4" = "auto" means: both trunks by the cost model below (-2).  HP3D_WINO4_AUTO=pose in the environment keeps
// HandSegNet on F(2x2,3x3) (-1; the same as option "wino4" = "pose") -- for A/B runs of whole test suites.
static int wino4_default() {
    const char* e = getenv("HP3D_WINO4_AUTO");
    return (e && !strcmp(e, "pose")) ? -1 : -2;
}
bool wino4_auto(int mode, bool posenet, int k, int cin_pad, int cout_pad, int Ho, int Wo, int B, int ks4, int old_nt, int old_ks, int ks2,
                bool two_streams) {
    if (mode == -1 && !posenet) return false;              // -1 = option "pose": PoseNet2D only; -2 = "auto": both trunks
    if ((long)B * Ho * Wo < 1024) return false;
    const double cus = hp3d_num_cus(), reduce_cost = 0.7;
    const int nsub = k == 7 ? 9 : 1;
    // units: one conv_wino.hip item-step (32 F(2x2) tiles x 128 couts x 32 channels).  A conv_wino4 item-step (32 F(4x4) tiles x 64
    // couts x 16 channels) covers the same amount of convolution and was measured at 0.72-0.78 of its time on filled launches
    // (7x7 layers 0.85: their split partial sums go through a reduce launch).  With the second stream live the other half's kernels
    // fill tail rounds, so rounds count fractionally.
    auto rounds = [&](double items) { return two_streams ? std::max(1.0, items / cus) : std::ceil(items / cus); };
    double t_best = 1e30;
    if (old_nt) {
        const long tiles = (long)B * ((Ho + 1) / 2) * ((Wo + 1) / 2);
        const long items1 = (tiles + old_nt - 1) / old_nt * (cout_pad / (old_nt == 32 ? 128 : 64));
        const int S1 = nsub * cin_pad / (old_nt == 32 ? 32 : 16);
        const double w1 = old_nt == 32 ? 1.0 : 0.5;
        t_best = old_ks > 1 ? std::ceil((double)S1 / old_ks) * w1 * std::max(1.0, items1 * old_ks / cus) + reduce_cost : rounds((double)items1) * S1 * w1;
    }
    {
        const long tiles = (long)B * ((Ho + 1) / 2) * ((Wo + 1) / 2);
        const long items2 = (tiles + 31) / 32 * (cout_pad / 64);
        const int S2 = nsub * cin_pad / 16;
        const double t2 = ks2 > 1 ? 1.07 * (std::ceil((double)S2 / ks2) * 0.25 * std::max(1.0, items2 * ks2 / cus)) + reduce_cost
                                  : 1.07 * (items2 * S2 * 0.25 / cus + S2 * 0.125);
        t_best = std::min(t_best, t2);
    }
    const long tiles4 = (long)B * ((Ho + 3) / 4) * ((Wo + 3) / 4);
    const long items4 = (tiles4 + 31) / 32 * (cout_pad / 64);
    const int S4 = nsub * cin_pad / 16;
    if (ks4 > 1 && (S4 + ks4 - 1) / ks4 < 8) return false;      // short split items are all prologue / epilogue (measured: B <= 2 loses)
    const double c4 = k == 7 ? 0.85 : 0.76;
    double t4 = ks4 > 1 ? c4 * std::ceil((double)S4 / ks4) * std::max(1.0, items4 * ks4 / cus) + reduce_cost : c4 * rounds((double)items4) * S4;
    // A 3x3 launch's last, under-filled round runs as tail pieces (conv_wino4_tail_plan): q steps instead of a whole round of S4, + the
    // pieces' epilogues and the tail reduce launch (~2 units).  Round 5: the model above charged a whole round and sent B = 12 / 24 layers with
    // 1.2-1.5 rounds of items to the F(2x2,3x3) kernels (PoseNet2D conv4_2 at B = 24: 0.416 ms there against 0.352 for THIRTY-TWO images here).
    // (Rounds 4-5 applied this to PoseNet2D only: HandSegNet's logits feed the mask threshold, and the reference-code fixtures demanded bit-equal
    // masks of noise maps whose knife-edge pixels move with ANY change of summation order -- a test fixture was choosing kernels.  Round 6: the
    // fixtures carry the reference run's knife-edge pixels and the tests tell such a flip from a defect (tests/test_gpu_reference_fixtures.py:
    // _mask_gate), so both trunks take the plan the model prices: +5-12 % at B = 8 ... 24, throughput monotonic in B again.)
    if (k == 3 && ks4 <= 1 && !two_streams) {
        int tail_items = 0;
        const int q = conv_wino4_tail_plan(cin_pad, cout_pad, Ho, Wo, B, &tail_items);
        if (q > 0 && tail_items > 0) t4 = std::min(t4, c4 * (std::floor(items4 / cus) * S4 + q) + 2.0);
    }
    return t4 < 0.97 * t_best;
}

static int wino2_ks_probe(hp3d_ctx* ctx, const ConvL& l, int Ho, int Wo, int B, int in_cs, int out_cs, int pool) {
    int ks = 1;
    (void)conv_wino2_eligible(l.k, l.stride, l.cin_pad, l.cout_pad, Ho, Wo, B, in_cs, out_cs, pool, ctx->wino_splitk ? &ks : nullptr);
    return ks;
}

// ---- one convolution layer -------------------------------------------------------------------
// in: [B,H,W,in_cs] (engine channels start at `in`), out: [B,Ho',Wo',out_cs] channel 0 at `out`.
// f16 = 1 (trunk nets after hp3d_finalize_weights(dtype=1)): `in` / `out` hold halves (except the raw image of
// conv1_1 and out_f32 heads); in_cs / out_cs are then counted in ELEMENTS of the respective tensor.
int kid_sync_state(hp3d_ctx* ctx);

static int wino7_ks_override(const hp3d_ctx* ctx, int ks, int cin_pad, long out_floats) {        // option "wino7_ksplit"
    if (ctx->wino7_ksplit <= 0 || !ctx->wino_splitk) return ks;
    ks = std::min(ctx->wino7_ksplit, cin_pad / 16);
    return (ks >= 2 && ks * out_floats * 4 < (1L << 31)) ? ks : 1;
}

int run_conv(hp3d_ctx* ctx, const ConvL& l, const float* in, int in_cs, int B, int H, int W, float* out, int out_cs,
             int pool, int* Ho_out, int* Wo_out, int f16 = 0, int out_f32 = 0) {
    int Ho, Wo, pt, pl;
    const int k = l.ek;
    same_pad(H, k, l.stride, &Ho, &pt);
    same_pad(W, k, l.stride, &Wo, &pl);
    const double flops = 2.0 * l.k * l.k * l.cin * l.cout * (double)Ho * Wo * B;
    const double bytes = (f16 ? 2.0 : 4.0) * ((double)B * H * W * l.cin + (double)l.k * l.k * l.cin * l.cout + l.cout +
                                (double)B * (pool ? (Ho / 2) * (Wo / 2) : Ho * Wo) * l.cout);
    if (l.hwio_off && ctx->tiny_gemm && !f16 && !pool && !ctx->conv_naive && l.stride == 2 && l.k == 3 && H == 8 && W == 8 && in_cs == l.cin && out_cs == l.cout) {
        // the last stride-2 layer of a lifting tower: 16 output pixels per image -- a split-K GEMM over them (glue.hip: conv_s2_gemm_launch)
        ProfScope ps(ctx, l.name, "conv_s2_gemm", flops, bytes);
        conv_s2_gemm_launch(in, B, 8, l.cin, ctx->blob + l.hwio_off, ctx->blob + l.b_off, l.cout, l.relu, out, ctx->d_fcpart, ctx->stream);
        ++ctx->conv_s2_gemm_launches;
        HIPCHK(ctx, hipGetLastError());
        if (Ho_out) *Ho_out = Ho;
        if (Wo_out) *Wo_out = Wo;
        return 0;
    }
    int wino_ks = 1, wino2_ks = 1;
    const int old_nt = (ctx->use_wino && !f16 && l.ww_off) ? conv_wino_eligible(ctx->use_wino, l.k, l.stride, l.cin_pad, l.cout_pad, Ho, Wo, B, in_cs, out_cs, pool,
                                                                                 ctx->wino_splitk ? &wino_ks : nullptr) : 0;
    // 7x7 layers (PoseNet2D's refinement units): Winograd F(4x4,4x4) over the filter's four 4x4-tap blocks when the launch fills the chip
    // (one work item = a 4x4 tile block x 64 couts, no channel split: 160 of 256 CUs busy already beats the split nine-block form + its reduce)
    // (under-filled launches -- small batches -- split the 16-channel chunks over workgroups and add the raw sums in a reduce launch, like the other
    //  Winograd kernels: option "wino_splitk")
    long items7 = 0;
    int ks7 = 1;
    const bool take7 = ctx->use_wino && ctx->use_wino7 && !f16 && l.ww7_off && !pool && !ctx->conv_naive &&
        conv_wino7_eligible(l.k, l.stride, l.cin_pad, l.cout_pad, Ho, Wo, B, in_cs, out_cs, &items7, ctx->wino_splitk ? &ks7 : nullptr) &&
        ((ks7 = wino7_ks_override(ctx, ks7, l.cin_pad, (long)B * Ho * Wo * l.cout_pad)), true) &&
        (ctx->use_wino7 == 1 || items7 >= (long)hp3d_num_cus() * 5 / 8 || (ks7 > 1 && ctx->use_wino7 == -1 && !ctx->two_streams_live));
    if (take7) {
        ConvParams p;
        p.in = in; p.wpk = ctx->blob + l.ww7_off; p.bias = ctx->blob + l.b_off; p.out = out;
        p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
        p.Cin = l.cin_pad; p.in_cs = in_cs; p.Cout = l.cout_pad; p.out_cs = out_cs;
        p.cout_store = std::min(l.cout_pad, out_cs);
        p.pad_t = pt; p.pad_l = pl; p.tiles_x = 0; p.tiles_y = 0;
        p.act = l.relu; p.im2col = 0; p.ksplit = ks7; p.partial = nullptr; p.f16 = 0; p.out_f32 = 0; p.nsub = 4;
        if (ks7 > 1) {
            const size_t need = (size_t)ks7 * B * Ho * Wo * l.cout_pad;
            if (need > ctx->col_floats) {
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                CHK(dev_realloc(ctx, &ctx->col, need));
                ctx->col_floats = need;
            }
            p.out = ctx->col; p.out_cs = l.cout_pad; p.cout_store = l.cout_pad;
        }
        {
            ProfScope ps(ctx, l.name, ks7 > 1 ? "conv_wino7_f4x4_4x4_as7x7_splitk" : "conv_wino7_f4x4_4x4_as7x7", flops, bytes);
            if (conv_wino7_launch(p, ctx->stream)) HP3D_FAIL(ctx, HP3D_ERR_ARG, "winograd conv (F(4x4,4x4)): launch refused");
        }
        if (ks7 > 1) {
            ProfScope ps(ctx, l.name, "conv_splitk_reduce", 0.0, 4.0 * (ks7 + 1) * B * Ho * Wo * l.cout_pad);
            conv_splitk_reduce_launch(ctx->col, ks7, (long)B * Ho * Wo, l.cout_pad, ctx->blob + l.b_off, l.relu, out, out_cs,
                                      std::min(l.cout_pad, out_cs), ctx->stream);
        }
        ++ctx->conv_wino7_launches;
        ctx->conv_wino7_split_launches += ks7 > 1;
        HIPCHK(ctx, hipGetLastError());
        if (Ho_out) *Ho_out = Ho;
        if (Wo_out) *Wo_out = Wo;
        return 0;
    }
    int wino4_ks = 1;
    const bool take4 = ctx->use_wino && ctx->use_wino4 && !f16 && l.ww4_off && !ctx->conv_naive &&
        conv_wino4_eligible(l.k, l.stride, l.cin_pad, l.cout_pad, Ho, Wo, B, in_cs, out_cs, pool, ctx->wino_splitk ? &wino4_ks : nullptr) &&
        (ctx->use_wino4 == 1 ||
         (ctx->use_wino4 < 0 && wino4_auto(ctx->use_wino4, l.net == NET_POSE, l.k, l.cin_pad, l.cout_pad, Ho, Wo, B, wino4_ks, old_nt, wino_ks,
                                           wino2_ks_probe(ctx, l, Ho, Wo, B, in_cs, out_cs, pool), ctx->two_streams_live)));
    // round 6: the filled 3x3 launches with Cin >= 128 on the bf16 matrix pipe with split operands (conv_wino4s.hip), option "wino4_split"
    int filled4s = 0;
    const bool take4s = ctx->use_wino && ctx->use_wino4s && !f16 && l.ww4s_off && !ctx->conv_naive && l.k == 3 &&
        conv_wino4s_eligible(l.k, l.stride, l.cin_pad, l.cout_pad, Ho, Wo, B, in_cs, out_cs, pool, &filled4s) &&
        (ctx->use_wino4s == 1 || (take4 && wino4_ks <= 1 && filled4s));
    if (take4s) {
        ConvParams p;
        p.in = in; p.wpk = ctx->blob + l.ww4s_off; p.bias = ctx->blob + l.b_off; p.out = out;
        p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
        p.Cin = l.cin_pad; p.in_cs = in_cs; p.Cout = l.cout_pad; p.out_cs = out_cs;
        p.cout_store = std::min(l.cout_pad, out_cs);
        p.pad_t = pt; p.pad_l = pl; p.tiles_x = 0; p.tiles_y = 0;
        p.act = l.relu; p.im2col = 0; p.ksplit = 1; p.partial = nullptr; p.f16 = 0; p.out_f32 = 0; p.nsub = 1;
        if (ctx->w4_tail && conv_wino4_tail_plan(l.cin_pad, l.cout_pad, Ho, Wo, B, nullptr) > 0) {
            const size_t need = conv_wino4s_tail_floats();
            if (need > ctx->col_floats) {
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                CHK(dev_realloc(ctx, &ctx->col, need));
                ctx->col_floats = need;
            }
            p.partial = ctx->col; p.partial_cap = ctx->col_floats;
        }
        {
            ProfScope ps(ctx, l.name, pool ? "conv_wino4s_f4x4_3x3_bf16x3_pool" : "conv_wino4s_f4x4_3x3_bf16x3", flops, bytes);
            const int lr = conv_wino4s_launch(p, pool, ctx->stream);
            if (lr < 0) HP3D_FAIL(ctx, HP3D_ERR_ARG, "winograd conv (F(4x4,3x3), split operands): launch refused");
            if (lr == 1) ++ctx->conv_wino4s_tail_launches;
        }
        ++ctx->conv_wino4s_launches;
        HIPCHK(ctx, hipGetLastError());
        if (Ho_out) *Ho_out = pool ? Ho / 2 : Ho;
        if (Wo_out) *Wo_out = pool ? Wo / 2 : Wo;
        return 0;
    }
    const bool take2 = !take4 && ctx->use_wino && ctx->use_wino2 && !f16 && l.ww2_off && !ctx->conv_naive &&
        conv_wino2_eligible(l.k, l.stride, l.cin_pad, l.cout_pad, Ho, Wo, B, in_cs, out_cs, pool, ctx->wino_splitk ? &wino2_ks : nullptr) &&
        (ctx->use_wino2 == 1 || wino2_auto(l.k, l.cin_pad, l.cout_pad, Ho, Wo, B, old_nt, wino_ks, wino2_ks, ctx->two_streams_live));
    if (take4 || take2) {
        if (take4) wino2_ks = wino4_ks;          // (the block below serves both kernels: same parameters, same split / reduce protocol)
        ConvParams p;
        p.in = in; p.wpk = ctx->blob + (take4 ? l.ww4_off : l.ww2_off); p.bias = ctx->blob + l.b_off; p.out = out;
        p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
        p.Cin = l.cin_pad; p.in_cs = in_cs; p.Cout = l.cout_pad; p.out_cs = out_cs;
        p.cout_store = std::min(l.cout_pad, out_cs);
        p.pad_t = pt; p.pad_l = pl; p.tiles_x = 0; p.tiles_y = 0;
        p.act = l.relu; p.im2col = 0; p.ksplit = wino2_ks; p.partial = nullptr; p.f16 = 0; p.out_f32 = 0;
        p.nsub = l.k == 7 ? 9 : 1;
        if (wino2_ks > 1) {
            const size_t need = (size_t)wino2_ks * B * Ho * Wo * l.cout_pad;
            if (need > ctx->col_floats) {
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                CHK(dev_realloc(ctx, &ctx->col, need));
                ctx->col_floats = need;
            }
            p.out = ctx->col; p.out_cs = l.cout_pad; p.cout_store = l.cout_pad;
        } else if (take4 && l.k == 3 && ctx->w4_tail &&
                   conv_wino4_tail_plan(l.cin_pad, l.cout_pad, Ho, Wo, B, nullptr) > 0) {
            // an under-filled last round of items runs as channel slices, one piece per CU (conv_wino4.hip, TAIL): scratch for the raw sums
            const size_t need = conv_wino4_tail_floats();
            if (need > ctx->col_floats) {
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                CHK(dev_realloc(ctx, &ctx->col, need));
                ctx->col_floats = need;
            }
            p.partial = ctx->col; p.partial_cap = ctx->col_floats;
        }
        {
            const char* kn = take4 ? (l.k == 7 ? (wino2_ks > 1 ? "conv_wino4_f4x4_3x3_as7x7_splitk" : "conv_wino4_f4x4_3x3_as7x7")
                                               : wino2_ks > 1 ? "conv_wino4_f4x4_3x3_splitk"
                                               : pool ? "conv_wino4_f4x4_3x3_pool" : "conv_wino4_f4x4_3x3")
                                    : (l.k == 7 ? (wino2_ks > 1 ? "conv_wino2_f2x2_3x3_as7x7_splitk" : "conv_wino2_f2x2_3x3_as7x7")
                                               : wino2_ks > 1 ? "conv_wino2_f2x2_3x3_splitk" : pool ? "conv_wino2_f2x2_3x3_pool" : "conv_wino2_f2x2_3x3");
            ProfScope ps(ctx, l.name, kn, flops, bytes);
            // (the F(4x4,3x3) launchers answer 1 when the last round really ran as tail pieces: the counter below is the tests' proof)
            const int lr = take4 ? conv_wino4_launch(p, wino2_ks > 1 ? 0 : pool, ctx->stream) : conv_wino2_launch(p, wino2_ks > 1 ? 0 : pool, ctx->stream);
            if (lr < 0) HP3D_FAIL(ctx, HP3D_ERR_ARG, "winograd conv (%s): launch refused", take4 ? "F(4x4,3x3)" : "2 workgroups per CU");
            if (take4 && lr == 1) ++ctx->conv_wino4_tail_launches;
        }
        ++(take4 ? ctx->conv_wino4_launches : ctx->conv_wino2_launches);
        if (wino2_ks > 1) {
            ProfScope ps(ctx, l.name, pool ? "conv_splitk_reduce_pool" : "conv_splitk_reduce", 0.0, 4.0 * (wino2_ks + 1) * B * Ho * Wo * l.cout_pad);
            if (pool)
                conv_splitk_reduce_pool_launch(ctx->col, wino2_ks, B, Ho, Wo, l.cout_pad, ctx->blob + l.b_off, l.relu, out, out_cs,
                                               std::min(l.cout_pad, out_cs), ctx->stream);
            else
                conv_splitk_reduce_launch(ctx->col, wino2_ks, (long)B * Ho * Wo, l.cout_pad, ctx->blob + l.b_off, l.relu, out, out_cs,
                                          std::min(l.cout_pad, out_cs), ctx->stream);
        }
    } else if (old_nt) {
        ConvParams p;
        p.in = in; p.wpk = ctx->blob + l.ww_off; p.bias = ctx->blob + l.b_off; p.out = out;
        p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
        p.Cin = l.cin_pad; p.in_cs = in_cs; p.Cout = l.cout_pad; p.out_cs = out_cs;
        p.cout_store = std::min(l.cout_pad, out_cs);
        p.pad_t = pt; p.pad_l = pl;
        p.tiles_x = (Wo + 15) / 16; p.tiles_y = (Ho + 7) / 8;
        p.act = l.relu; p.im2col = 0; p.ksplit = wino_ks; p.partial = nullptr; p.f16 = 0; p.out_f32 = 0;
        p.nsub = l.k == 7 ? 9 : 1;
        if (wino_ks > 1) {       // under-filled chip: channel steps split over workgroups, raw partial sums, then one reduce
            const size_t need = (size_t)wino_ks * B * Ho * Wo * l.cout_pad;
            if (need > ctx->col_floats) {
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                CHK(dev_realloc(ctx, &ctx->col, need));
                ctx->col_floats = need;
            }
            p.out = ctx->col; p.out_cs = l.cout_pad; p.cout_store = l.cout_pad;
        }
        {
            ProfScope ps(ctx, l.name, l.k == 7 ? (wino_ks > 1 ? "conv_wino_f2x2_3x3_as7x7_splitk" : "conv_wino_f2x2_3x3_as7x7")
                                      : wino_ks > 1 ? "conv_wino_f2x2_3x3_splitk" : pool ? "conv_wino_f2x2_3x3_pool" : "conv_wino_f2x2_3x3", flops, bytes);
            // split channel steps: raw partial sums at conv resolution, the pool (if any) happens in the reduce
            if (conv_wino_launch(p, wino_ks > 1 ? 0 : pool, ctx->stream)) HP3D_FAIL(ctx, HP3D_ERR_ARG, "winograd conv: tensor exceeds 32-bit offsets");
        }
        if (wino_ks > 1) {
            ProfScope ps(ctx, l.name, pool ? "conv_splitk_reduce_pool" : "conv_splitk_reduce", 0.0, 4.0 * (wino_ks + 1) * B * Ho * Wo * l.cout_pad);
            if (pool)
                conv_splitk_reduce_pool_launch(ctx->col, wino_ks, B, Ho, Wo, l.cout_pad, ctx->blob + l.b_off, l.relu, out, out_cs,
                                               std::min(l.cout_pad, out_cs), ctx->stream);
            else
                conv_splitk_reduce_launch(ctx->col, wino_ks, (long)B * Ho * Wo, l.cout_pad, ctx->blob + l.b_off, l.relu, out, out_cs,
                                          std::min(l.cout_pad, out_cs), ctx->stream);
        }
    } else if (l.mode == 1 && !pool && !out_f32 && ctx->use_first && !ctx->conv_naive &&
               conv_first_eligible(l.k, l.stride, l.cin, l.cout, B, H, W, out_cs, f16)) {
        ConvParams p;
        p.in = in; p.wpk = ctx->blob + l.w_off; p.bias = ctx->blob + l.b_off; p.out = out;
        p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
        p.Cin = 3; p.in_cs = 3; p.Cout = 64; p.out_cs = out_cs; p.cout_store = 64;
        p.pad_t = pt; p.pad_l = pl; p.tiles_x = 0; p.tiles_y = 0;
        p.act = l.relu; p.im2col = 1; p.ksplit = 1; p.partial = nullptr; p.f16 = f16; p.out_f32 = 0; p.nsub = 1;
        // A COLD input image (the caller's device buffer, an upload: not written by the kernel in front of this one) costs this
        // store-bound kernel a third of its rate -- its gathers run one tile ahead, an HBM miss takes longer than a tile (round 5: 3.5
        // TB/s in the pipeline against 4.8 standalone and for PoseNet2D's freshly written crop).  Streaming the image once through the
        // memory-side cache first (13 us for 39 MB) takes B = 32 at 320 x 320 from 0.253 to 0.175 ms (5.0 TB/s = 0.63 of the HBM spec).
        const size_t img_bytes = (size_t)B * H * W * 12;
        // The pass runs on the child context's stream BESIDE the convolution (it is ahead of the gathers after the first few tiles and
        // needs 16 registers per wave next to the convolution's 3 x 160): its 13 us disappear; in this stream when there is no second one.
        bool touch_beside = false;
        if (ctx->d_keys && (ctx->first_touch == 1 || (ctx->first_touch < 0 && !ctx->trunk_input_hot && img_bytes >= (8u << 20) && img_bytes <= (128u << 20)))) {
#ifndef HP3D_EMU
            if (ctx->first_touch_beside && !ctx->use_graph && !ctx->shared_weights && !ctx->two_streams_live && kid_sync_state(ctx) == 0) {
                hp3d_ctx* k = ctx->kid;
                HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));          // (whatever produced the image on this stream comes first)
                HIPCHK(ctx, hipStreamWaitEvent(k->stream, ctx->ev_fork, 0));
                touch_launch(in, (size_t)B * H * W * 3, (float*)ctx->d_keys, k->stream);
                HIPCHK(ctx, hipEventRecord(ctx->ev_join, k->stream));
                touch_beside = true;
            }
#endif
            if (!touch_beside) {
                ProfScope pt(ctx, l.name, "conv_first_touch", 0.0, 0.0);
                touch_launch(in, (size_t)B * H * W * 3, (float*)ctx->d_keys, ctx->stream);
            }
            ++ctx->first_touch_launches;
        }
        {
            ProfScope ps(ctx, l.name, f16 ? "conv_first_3x3_c3_f16" : "conv_first_3x3_c3", flops, bytes);
            conv_first_launch(p, ctx->stream, ctx->first_balanced);
        }
#ifndef HP3D_EMU
        if (touch_beside) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));      // the call's completion covers the read pass
#endif
        ++ctx->conv_first_launches;
    } else if (f16 && ctx->use_h16 && !ctx->conv_naive &&
               ((l.mode == 0 && l.k == 3) || (ctx->h16_k7k1 && ((l.k == 7 && (l.mode == 0 || l.mode == 2)) || (l.k == 1 && l.mode == 0)))) &&
               conv_h16_eligible(ctx->use_h16, l.k, l.stride, l.cin_pad16 / 2, l.cout_pad, Ho, Wo, B, out_f32, out_cs) &&
               ((uintptr_t)out & 15) == 0 && !(pool && ((Ho | Wo) & 1))) {
        ConvParams p;
        p.in = in; p.wpk = (const float*)(ctx->blob16 + l.w16_off); p.bias = ctx->blob + l.b_off; p.out = out;
        p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
        p.Cin = l.cin_pad16 / 2; p.in_cs = in_cs / 2; p.Cout = l.cout_pad; p.out_cs = out_cs;
        p.cout_store = std::min(l.cout_pad, out_cs);
        p.pad_t = pt; p.pad_l = pl; p.tiles_x = 0; p.tiles_y = 0;
        p.act = l.relu; p.im2col = 0; p.ksplit = 1; p.partial = nullptr; p.f16 = 1; p.out_f32 = 0; p.nsub = 1;
        ProfScope ps(ctx, l.name, l.k == 7 ? "conv_h16_7x7" : l.k == 1 ? "conv_h16_1x1" : pool ? "conv_h16_3x3_pool" : "conv_h16_3x3", flops, bytes);
        ++ctx->conv_h16_launches;
        if (conv_h16_launch(p, pool, ctx->stream)) HP3D_FAIL(ctx, HP3D_ERR_UNSUPPORTED, "conv_h16 launch failed for %s", l.name.c_str());
    } else if (ctx->conv_naive && l.mode == 0 && !pool && !f16) {
        ProfScope ps(ctx, l.name, "conv_naive", flops, bytes);
        conv_naive_launch(in, B, H, W, l.cin, in_cs, ctx->naive_w[l.name], ctx->blob + l.b_off, l.k, l.stride, l.cout,
                          l.relu, out, out_cs, Ho, Wo, pt, pl, ctx->stream);
    } else {
        ConvPlan plan;
        const int cin_units = f16 ? l.cin_pad16 / 2 : l.cin_pad;       // 4-byte channel units the kernel iterates
        if (conv_mfma_plan(k, l.stride, Ho, Wo, cin_units, l.cout_pad, pool, B, &plan) != 0)
            HP3D_FAIL(ctx, HP3D_ERR_UNSUPPORTED, "no conv_mfma variant for %s (k=%d s=%d)", l.name.c_str(), k, l.stride);
        if (l.mode == 1 || (f16 && !out_f32)) plan.ksplit = 1;          // split-K partials are float32
        if (plan.ksplit > 1) {
            const size_t need = (size_t)plan.ksplit * B * Ho * Wo * l.cout_pad;
            if (need > ctx->col_floats) {
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                CHK(dev_realloc(ctx, &ctx->col, need));
                ctx->col_floats = need;
            }
        }
        ConvParams p;
        p.in = in; p.wpk = f16 ? (const float*)(ctx->blob16 + l.w16_off) : ctx->blob + l.w_off;
        p.bias = ctx->blob + l.b_off; p.out = out;
        p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
        p.Cin = cin_units; p.in_cs = (f16 && l.mode != 1) ? in_cs / 2 : in_cs; p.Cout = l.cout_pad; p.out_cs = out_cs;
        p.f16 = f16; p.out_f32 = out_f32;
        p.cout_store = std::min(l.cout_pad, out_cs);
        p.pad_t = pt; p.pad_l = pl;
        p.tiles_x = (Wo + plan.tw - 1) / plan.tw; p.tiles_y = (Ho + plan.th - 1) / plan.th;
        p.act = l.relu; p.im2col = (l.mode == 1);
        p.ksplit = plan.ksplit; p.partial = ctx->col;
        std::string kname = conv_mfma_variant_name(k, l.stride, pool, plan);
        if (f16) kname += "_f16";
        ProfScope ps(ctx, l.name, kname.c_str(), flops, bytes);
        if (conv_mfma_launch(p, k, l.stride, pool, plan, ctx->stream) != 0)
            HP3D_FAIL(ctx, HP3D_ERR_UNSUPPORTED, "conv_mfma launch failed for %s", l.name.c_str());
        if (plan.ksplit > 1)
            conv_splitk_reduce_launch(ctx->col, plan.ksplit, (long)B * Ho * Wo, l.cout_pad, p.bias, l.relu, out, out_cs,
                                      p.cout_store, ctx->stream);
    }
    HIPCHK(ctx, hipGetLastError());
    if (Ho_out) *Ho_out = pool ? Ho / 2 : Ho;
    if (Wo_out) *Wo_out = pool ? Wo / 2 : Wo;
    return 0;
}

const ConvL& CL(hp3d_ctx* ctx, const char* name) { return ctx->T.conv[ctx->T.conv_idx.at(name)]; }
const FcL& FL(hp3d_ctx* ctx, const char* name) { return ctx->T.fc[ctx->T.fc_idx.at(name)]; }

// Two 1x1 layers as one launch (conv_pw2.hip): l1 = 128 -> H (a multiple of 128), l2 = H -> at most 32 couts, float32 mode, enough pixels to fill
// the chip (64 pixels per workgroup, two workgroups per CU).  Returns 1 if it ran, 0 if the shapes / options do not allow it (the caller
// then runs the two layers one by one).
int run_pw2(hp3d_ctx* ctx, const ConvL& l1, const ConvL& l2, const float* in, int in_cs, int B, int h, int w, float* out, int out_cs) {
    const long npix = (long)B * h * w;
    if (!ctx->use_pw2 || ctx->prec || ctx->conv_naive || l1.k != 1 || l2.k != 1 || l1.stride != 1 || l2.stride != 1 || l1.mode != 0 || l2.mode != 0 ||
        l2.cin_pad != l1.cout_pad || (ctx->use_pw2 == 1 && npix < 64L * hp3d_num_cus()) ||
        !conv_pw2_eligible(l1.cin_pad, l1.cout_pad, l2.cout_pad, npix, in_cs, out_cs) || (in_cs & 3) || ((uintptr_t)in & 15))
        return 0;
    Pw2Params p;
    p.in = in; p.in_cs = in_cs; p.npix = npix;
    p.w1 = ctx->blob + l1.w_off; p.b1 = ctx->blob + l1.b_off; p.H = l1.cout_pad; p.act1 = l1.relu;
    p.w2 = ctx->blob + l2.w_off; p.b2 = ctx->blob + l2.b_off; p.act2 = l2.relu;
    p.out = out; p.out_cs = out_cs; p.cout_store = std::min(l2.cout_pad, out_cs);
    const double flops = 2.0 * npix * ((double)l1.cin * l1.cout + (double)l2.cin * l2.cout);
    const double bytes = 4.0 * (npix * ((double)l1.cin + l2.cout) + (double)l1.cin * l1.cout + (double)l2.cin * l2.cout);
    ProfScope ps(ctx, l1.name + "+" + l2.name.substr(l2.name.find('/') + 1), "conv_pw2_1x1_1x1", flops, bytes);
    if (conv_pw2_launch(p, ctx->stream)) HP3D_FAIL(ctx, HP3D_ERR_ARG, "conv_pw2: launch refused for %s", l1.name.c_str());
    ++ctx->conv_pw2_launches;
    HIPCHK(ctx, hipGetLastError());
    return 1;
}

// Half-precision trunks: conv1_1 (3 -> 64) + conv1_2 (64 -> 64) + 2x2 max-pool as ONE launch of conv_h16.hip's fused form
// (conv1_1's activation never reaches HBM).  Returns 1 if it ran, 0 if the shape / options do not allow it.
int run_fused12(hp3d_ctx* ctx, const ConvL& l1, const ConvL& l2, const float* image, int B, int H, int W, float* out, int* oh, int* ow) {
    if (!ctx->prec || !ctx->use_h16 || !ctx->fuse12 || ctx->conv_naive || l1.mode != 1 || l2.mode != 0 || l1.cout != 64 || l2.cout != 64 ||
        !l1.relu || !l2.relu || ((H | W) & 1) || ((uintptr_t)out & 15) ||
        !conv_h16_eligible(ctx->use_h16, l2.k, l2.stride, l2.cin_pad16 / 2, l2.cout_pad, H, W, B, 0, 64))
        return 0;
    ConvParams p;
    p.in = image; p.wpk = (const float*)(ctx->blob16 + l2.w16_off); p.bias = ctx->blob + l2.b_off; p.out = out;
    p.wpk1 = ctx->blob + l1.w_off; p.bias1 = ctx->blob + l1.b_off;
    p.B = B; p.H = H; p.W = W; p.Ho = H; p.Wo = W;
    p.Cin = l2.cin_pad16 / 2; p.in_cs = 3; p.Cout = 64; p.out_cs = 64; p.cout_store = 64;
    p.pad_t = 1; p.pad_l = 1; p.tiles_x = 0; p.tiles_y = 0;
    p.act = 1; p.im2col = 1; p.ksplit = 1; p.partial = nullptr; p.f16 = 1; p.out_f32 = 0; p.nsub = 1;
    const double px = (double)B * H * W;
    const double flops = 2.0 * 9 * (3.0 * 64 + 64.0 * 64) * px;
    const double bytes = px * 12 + 2.0 * (9 * 3 * 64 + 9 * 64 * 64) + 2.0 * px / 4 * 64;
    ProfScope ps(ctx, l2.name, "conv_h16_fused_c1_1+c1_2_pool", flops, bytes);
    ++ctx->conv_h16_launches;
    p.first_form = ctx->fuse12 == 2 ? 1 : ctx->fuse12 == 3 ? 2 : 0;
    const int form = conv_h16_fused12_launch(p, ctx->stream);
    if (form < 0) HP3D_FAIL(ctx, HP3D_ERR_UNSUPPORTED, "conv_h16 fused launch failed for %s", l2.name.c_str());
    if (form == 2) ++ctx->conv_h16_first_resident_launches;
    *oh = H / 2; *ow = W / 2;
    return 1;
}

// VGG-style trunk shared by HandSegNet and PoseNet2D: conv1_1 (im2col) ... through block 4's first
// `n4` layers.  Returns the activation pointer / size after the trunk.
int run_trunk(hp3d_ctx* ctx, const char* scope, const float* image, int B, int H, int W, int n4, float** act, int* h,
              int* w, bool input_hot) {
    ctx->trunk_input_hot = input_hot;
    char nm[64];
    float* a = ctx->bufA;
    float* b = ctx->bufB;
    const int f16 = ctx->prec;        // activations below are halves when set (same buffers, half the bytes)
    int ch = 64, ih = H, iw = W;
    snprintf(nm, sizeof nm, "%s/conv1_1", scope);
    char nm2[64];
    snprintf(nm2, sizeof nm2, "%s/conv1_2", scope);
    int first = 1, fh = 0, fw = 0;
    const int fused = f16 ? run_fused12(ctx, CL(ctx, nm), CL(ctx, nm2), image, B, ih, iw, a, &fh, &fw) : 0;
    if (fused < 0) return fused;
    if (fused) { first = 2; ih = fh; iw = fw; }
    else CHK(run_conv(ctx, CL(ctx, nm), image, 3, B, ih, iw, a, 64, 0, nullptr, nullptr, f16));   // im2col fused in the loader
    const int nl[4] = {2, 2, 4, n4}, chs[4] = {64, 128, 256, 512};
    for (int blk = 0; blk < 4; ++blk) {
        for (int i = (blk == 0 ? first : 0); i < nl[blk]; ++i) {
            snprintf(nm, sizeof nm, "%s/conv%d_%d", scope, blk + 1, i + 1);
            const int pool = (blk < 3 && i == nl[blk] - 1) ? 1 : 0;
            int oh, ow;
            CHK(run_conv(ctx, CL(ctx, nm), a, ch, B, ih, iw, b, chs[blk], pool, &oh, &ow, f16));
            std::swap(a, b);
            ch = chs[blk]; ih = oh; iw = ow;
        }
    }
    *act = a; *h = ih; *w = iw;
    return 0;
}

// HandSegNet (nets/ColorHandPose3DNetwork.py:131-168) -> d_segsmall [B,H/8,W/8,32] (channels 0,1 real)
int run_handsegnet(hp3d_ctx* ctx, const float* image, int B, int H, int W, bool image_hot = false) {
    float* a; int h, w;
    CHK(run_trunk(ctx, "HandSegNet", image, B, H, W, 4, &a, &h, &w, image_hot));
    float* b = (a == ctx->bufA) ? ctx->bufB : ctx->bufA;
    const int f16 = ctx->prec;
    CHK(run_conv(ctx, CL(ctx, "HandSegNet/conv5_1"), a, 512, B, h, w, b, 512, 0, nullptr, nullptr, f16));
    CHK(run_conv(ctx, CL(ctx, "HandSegNet/conv5_2"), b, 512, B, h, w, a, 128, 0, nullptr, nullptr, f16));
    // conv6_1 (1x1, 128 -> 512) + conv6_2 (1x1, 512 -> 2): one launch where conv_pw2.hip takes the pair (the 512-channel map stays in LDS)
    const int fused = run_pw2(ctx, CL(ctx, "HandSegNet/conv6_1"), CL(ctx, "HandSegNet/conv6_2"), a, 128, B, h, w, ctx->d_segsmall, 32);
    if (fused < 0) return fused;
    if (fused) return 0;
    CHK(run_conv(ctx, CL(ctx, "HandSegNet/conv6_1"), a, 128, B, h, w, b, 512, 0, nullptr, nullptr, f16));
    // the 2-class head always leaves float32 logits for the softmax / mask stage
    CHK(run_conv(ctx, CL(ctx, "HandSegNet/conv6_2"), b, 512, B, h, w, ctx->d_segsmall, 32, 0, nullptr, nullptr, f16, 1));
    return 0;
}

// PoseNet2D (nets/ColorHandPose3DNetwork.py:170-219) -> d_sm[0..2] [B,h/8,w/8,32] (21 real channels)
int run_posenet(hp3d_ctx* ctx, const float* crop, int B, int H, int W, bool crop_hot = false) {
    float* a; int h, w;
    CHK(run_trunk(ctx, "PoseNet2D", crop, B, H, W, 2, &a, &h, &w, crop_hot));
    CHK(ensure_pose_bufs(ctx, B, h, w));
    const int f16 = ctx->prec;
    // concat([scoremap, encoding]) buffer: f32 [B,h,w,160] or f16 [B,h,w,192] (64-half chunks); channel 0..127 =
    // encoding, 128..148 = score map, rest zero (the f16 pad 160..191 is zeroed when the buffer is allocated)
    const int ccs = f16 ? 192 : 160;
    float* cat = f16 ? (float*)ctx->d_concat16 : ctx->d_concat;
    float* b = (a == ctx->bufA) ? ctx->bufB : ctx->bufA;
    CHK(run_conv(ctx, CL(ctx, "PoseNet2D/conv4_3"), a, 512, B, h, w, b, 256, 0, nullptr, nullptr, f16));
    CHK(run_conv(ctx, CL(ctx, "PoseNet2D/conv4_4"), b, 256, B, h, w, a, 256, 0, nullptr, nullptr, f16));
    CHK(run_conv(ctx, CL(ctx, "PoseNet2D/conv4_5"), a, 256, B, h, w, b, 256, 0, nullptr, nullptr, f16));
    CHK(run_conv(ctx, CL(ctx, "PoseNet2D/conv4_6"), b, 256, B, h, w, a, 256, 0, nullptr, nullptr, f16));
    CHK(run_conv(ctx, CL(ctx, "PoseNet2D/conv4_7"), a, 256, B, h, w, cat, ccs, 0, nullptr, nullptr, f16));
    // score-map heads always store float32 [.,32] (they feed the lifting nets, the up-sampler and the caller)
    {
        const int fused = f16 ? 0 : run_pw2(ctx, CL(ctx, "PoseNet2D/conv5_1"), CL(ctx, "PoseNet2D/conv5_2"), cat, ccs, B, h, w, ctx->d_sm[0], 32);
        if (fused < 0) return fused;
        if (!fused) {
            CHK(run_conv(ctx, CL(ctx, "PoseNet2D/conv5_1"), cat, ccs, B, h, w, a, 512, 0, nullptr, nullptr, f16));
            CHK(run_conv(ctx, CL(ctx, "PoseNet2D/conv5_2"), a, 512, B, h, w, ctx->d_sm[0], 32, 0, nullptr, nullptr, f16, 1));
        }
    }
    char nm[64];
    const int npix = B * h * w;
    for (int p = 0; p < 2; ++p) {
        // x = concat([scoremap, encoding]) : scoremap (21 real + 11 zero) -> channels 128..159
        if (f16) cvt_channels_f16_launch(ctx->d_sm[p], npix, 32, 32, ctx->d_concat16 + 128, 192, ctx->stream);
        else copy_channels_launch(ctx->d_sm[p], npix, 32, 32, ctx->d_concat + 128, 160, ctx->stream);
        const float* x = cat;
        int xcs = ccs;
        float* o = a;
        for (int r = 1; r <= 5; ++r) {
            snprintf(nm, sizeof nm, "PoseNet2D/conv%d_%d", p + 6, r);
            CHK(run_conv(ctx, CL(ctx, nm), x, xcs, B, h, w, o, 128, 0, nullptr, nullptr, f16));
            x = o; xcs = 128;
            o = (o == a) ? b : a;
        }
        char nm7[64];
        snprintf(nm, sizeof nm, "PoseNet2D/conv%d_6", p + 6);
        snprintf(nm7, sizeof nm7, "PoseNet2D/conv%d_7", p + 6);
        const int fused = f16 ? 0 : run_pw2(ctx, CL(ctx, nm), CL(ctx, nm7), x, 128, B, h, w, ctx->d_sm[p + 1], 32);
        if (fused < 0) return fused;
        if (!fused) {
            CHK(run_conv(ctx, CL(ctx, nm), x, 128, B, h, w, o, 128, 0, nullptr, nullptr, f16));
            CHK(run_conv(ctx, CL(ctx, nm7), o, 128, B, h, w, ctx->d_sm[p + 1], 32, 0, nullptr, nullptr, f16, 1));
        }
    }
    return 0;
}

// x2 != nullptr: the layer's last two inputs are the hand-side one-hot (concat(flatten, hand_side), :262-263 / :297-298, read in place)
int run_fc(hp3d_ctx* ctx, const FcL& l, const float* x, int B, int x_stride, float* out, int out_stride, const float* x2 = nullptr) {
    ProfScope ps(ctx, l.name, "fc", 2.0 * l.cin * l.cout * B, 4.0 * ((double)l.cin * l.cout + (double)B * (l.cin + l.cout)));
    fc_launch(x, B, l.cin, x_stride, ctx->blob + l.w_off, ctx->blob + l.b_off, l.cout, l.relu, out, out_stride,
              ctx->d_fcpart, ctx->stream, x2, l.cin - 2);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

// the three FC layers of a tower as two launches (round 6): the K slices of the first, then its reduction + the two small layers (glue.hip: fc_tail_kernel)
int run_fc_tail(hp3d_ctx* ctx, const FcL& l0, const FcL& l1, const FcL& l2, const float* x, int B, int x_stride, const float* hs, float* out, int out_stride) {
    int ns;
    {
        ProfScope ps(ctx, l0.name, "fc_partial", 2.0 * l0.cin * l0.cout * B, 4.0 * ((double)l0.cin * l0.cout + (double)B * (l0.cin + l0.cout)));
        ns = fc_partial_launch(x, B, l0.cin, x_stride, ctx->blob + l0.w_off, l0.cout, ctx->d_fcpart, ctx->stream, hs, l0.cin - 2);
    }
    {
        ProfScope ps(ctx, l1.name + "+" + l2.name.substr(l2.name.find('/') + 1), "fc_tail", 2.0 * B * ((double)l1.cin * l1.cout + (double)l2.cin * l2.cout),
                     4.0 * ((double)l1.cin * l1.cout + (double)l2.cin * l2.cout + (double)ns * B * l0.cout));
        fc_tail_launch(ctx->d_fcpart, ns, B, l0.cout, ctx->blob + l0.b_off, l0.relu, ctx->blob + l1.w_off, ctx->blob + l1.b_off, l1.cout, l1.relu,
                       ctx->blob + l2.w_off, ctx->blob + l2.b_off, l2.cout, l2.relu, out, out_stride, ctx->stream);
    }
    ++ctx->fc_tail_launches;
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

// _inference_pose3d_can (nets/ColorHandPose3DNetwork.py:249-272; bottleneck: nets/PosePriorNetwork.py:115-116)
int run_poseprior_can(hp3d_ctx* ctx, const float* sm32 /*[B,32,32,32]*/, const float* hs, int B, int bottleneck,
                      float* can) {
    char nm[64];
    const float* x = sm32;
    int cs = 32, h = 32, w = 32;
    float* a = ctx->bufA;
    float* b = ctx->bufB;
    const int ch[3] = {32, 64, 128};
    for (int i = 0; i < 3; ++i) {
        snprintf(nm, sizeof nm, "PosePrior/conv_pose_%d_1", i);
        CHK(run_conv(ctx, CL(ctx, nm), x, cs, B, h, w, a, ch[i], 0, &h, &w));
        snprintf(nm, sizeof nm, "PosePrior/conv_pose_%d_2", i);
        CHK(run_conv(ctx, CL(ctx, nm), a, ch[i], B, h, w, b, ch[i], 0, &h, &w));
        x = b; cs = ch[i];   // next pair: b -> a -> b (no aliasing)
    }
    // x: [B,4,4,128] contiguous == NHWC flatten (h,w,c)
    CHK(run_fc(ctx, FL(ctx, "PosePrior/fc_rel0"), x, B, 2048, ctx->d_fc1, 512, hs));
    // (round 6: ViewpointNet's tail runs as one launch, fc_tail_kernel; here that form measured SLOWER -- the 512 x 512 layer is a 1 MB weight
    //  stream per workgroup (60 us), and with that layer kept on fc_partial the 512 -> 63 layer's K loop alone took 39 us: three plain FC layers stay)
    CHK(run_fc(ctx, FL(ctx, "PosePrior/fc_rel1"), ctx->d_fc1, B, 512, ctx->d_fc2, 512));
    if (bottleneck) {
        CHK(run_fc(ctx, FL(ctx, "PosePrior/fc_bottleneck"), ctx->d_fc2, B, 512, ctx->d_fc1, 32));
        CHK(run_fc(ctx, FL(ctx, "PosePrior/fc_xyz#bn"), ctx->d_fc1, B, 32, can, 63));
    } else {
        CHK(run_fc(ctx, FL(ctx, "PosePrior/fc_xyz"), ctx->d_fc2, B, 512, can, 63));
    }
    return 0;
}

// _rotation_estimation (nets/ColorHandPose3DNetwork.py:285-309) -> u [B,3]
int run_viewpoint(hp3d_ctx* ctx, const float* sm32, const float* hs, int B, float* u) {
    char nm[64];
    const float* x = sm32;
    int cs = 32, h = 32, w = 32;
    float* a = ctx->bufA;
    float* b = ctx->bufB;
    const int ch[3] = {64, 128, 256};
    for (int i = 0; i < 3; ++i) {
        snprintf(nm, sizeof nm, "ViewpointNet/conv_vp_%d_1", i);
        CHK(run_conv(ctx, CL(ctx, nm), x, cs, B, h, w, a, ch[i], 0, &h, &w));
        snprintf(nm, sizeof nm, "ViewpointNet/conv_vp_%d_2", i);
        CHK(run_conv(ctx, CL(ctx, nm), a, ch[i], B, h, w, b, ch[i], 0, &h, &w));
        x = b; cs = ch[i];   // next pair: b -> a -> b (no aliasing)
    }
    if (ctx->fc_tail && fc_tail_eligible(256, 128, 3)) {
        CHK(run_fc_tail(ctx, FL(ctx, "ViewpointNet/fc_vp0"), FL(ctx, "ViewpointNet/fc_vp1"), FL(ctx, "ViewpointNet/fc_vp_u"), x, B, 4096, hs, u, 3));
        return 0;
    }
    CHK(run_fc(ctx, FL(ctx, "ViewpointNet/fc_vp0"), x, B, 4096, ctx->d_fc1, 256, hs));
    CHK(run_fc(ctx, FL(ctx, "ViewpointNet/fc_vp1"), ctx->d_fc1, B, 256, ctx->d_fc2, 128));
    CHK(run_fc(ctx, FL(ctx, "ViewpointNet/fc_vp_u"), ctx->d_fc2, B, 128, u, 3));
    return 0;
}

// lift_fused.hip's grid barrier gives up after a bounded spin (a workgroup that is not resident would otherwise hang the GPU) and
// raises this word; every synchronisation point of the executor turns it into an error instead of handing out garbage.
int check_lift_error(hp3d_ctx* ctx) {
    if (ctx->h_lifterr && *(volatile unsigned*)ctx->h_lifterr) {
        *ctx->h_lifterr = 0;
        HP3D_FAIL(ctx, HP3D_ERR_HIP, "lift_fused: grid barrier timed out (a workgroup was not resident); the outputs of that call are invalid -- "
                                     "set option lift_fused=0 on a shared GPU");
    }
    if (ctx->kid) return check_lift_error(ctx->kid);
    return 0;
}

// _inference_pose3d (nets/ColorHandPose3DNetwork.py:221-247)
// The same two towers as ONE launch (lift_fused.hip): small batches, where 24 dependent launches of ~13 us are the cost
int run_lift_fused(hp3d_ctx* ctx, const float* sm32, const float* hs, int B, int bottleneck, int do_rot) {
    LiftFusedParams p;
    memset(&p, 0, sizeof p);
    p.sm = sm32; p.sm_cs = 32; p.hs = hs; p.B = B; p.towers = do_rot ? 3 : 1;
    char nm[64];
    for (int t = 0; t < 2; ++t)
        for (int i = 0; i < 6; ++i) {
            snprintf(nm, sizeof nm, t == 0 ? "PosePrior/conv_pose_%d_%d" : "ViewpointNet/conv_vp_%d_%d", i / 2, i % 2 + 1);
            const ConvL& l = CL(ctx, nm);
            p.w[t * 6 + i] = ctx->blob + l.raw_off; p.b[t * 6 + i] = ctx->blob + l.b_off;
            p.cin[t * 6 + i] = l.cin4; p.cout[t * 6 + i] = l.cout;
        }
    const char* fcn[6] = {"PosePrior/fc_rel0", "PosePrior/fc_rel1", bottleneck ? "PosePrior/fc_xyz#bn" : "PosePrior/fc_xyz",
                          "ViewpointNet/fc_vp0", "ViewpointNet/fc_vp1", "ViewpointNet/fc_vp_u"};
    for (int f = 0; f < 6; ++f) {
        const FcL& l = FL(ctx, fcn[f]);
        p.fw[f] = ctx->blob + l.w_off; p.fb[f] = ctx->blob + l.b_off; p.fc_in[f] = l.cin; p.fc_out[f] = l.cout;
    }
    if (bottleneck) {
        const FcL& l = FL(ctx, "PosePrior/fc_bottleneck");
        p.bn_w = ctx->blob + l.w_off; p.bn_b = ctx->blob + l.b_off;
        p.fc_in[2] = 512;             // the last stage reads fc_rel1's 512 outputs; the 30-wide layer sits inside it
    }
    const size_t act = (size_t)B * 32 * 32 * 64;
    for (int t = 0; t < 2; ++t)
        for (int q = 0; q < 2; ++q) p.act[t][q] = ctx->bufA + (size_t)(t * 2 + q) * act;
    float* fp = ctx->d_fcpart;
    p.fcp[0][0] = fp; fp += (size_t)9 * B * 512;
    p.fcp[0][1] = fp; fp += (size_t)2 * B * 512;
    p.fcp[1][0] = fp; fp += (size_t)17 * B * 256;
    p.fcp[1][1] = fp;
    p.out[0] = ctx->d_can; p.out[1] = ctx->d_u;
    if (!ctx->d_liftbar) CHK(dev_realloc(ctx, &ctx->d_liftbar, 4));
    p.bar = ctx->d_liftbar;
#ifndef HP3D_EMU
    if (!ctx->h_lifterr) {
        HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_lifterr, sizeof(unsigned), hipHostMallocMapped));
        *ctx->h_lifterr = 0;
    }
    CHK(check_lift_error(ctx));            // an earlier launch of this context already failed: do not pile work on garbage
    HIPCHK(ctx, hipHostGetDevicePointer((void**)&p.err, ctx->h_lifterr, 0));
#endif
    ProfScope ps(ctx, "PosePrior+ViewpointNet", "lift_fused", 2.0 * B * (45.0e6 / 2 + 157.0e6 / 2 * (do_rot ? 1 : 0) + 2.7e6 / 2), 4.0 * 15.5e6);
    if (lift_fused_launch(p, ctx->stream)) HP3D_FAIL(ctx, HP3D_ERR_HIP, "lift_fused launch failed");
    ++ctx->lift_fused_launches;
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

int kid_sync_state(hp3d_ctx* ctx);

// The child context's share of an overlapped lifting stage: ViewpointNet's activations ([B,32,32,64] is the largest), its FC buffers
// and split-K partial sums.  (A later two-stream call grows the same arena through ensure_arena.)
int ensure_side_tower(hp3d_ctx* k, int B) {
    const size_t act = (size_t)B * 32 * 32 * 64;
    if (act > k->act_floats) {
        CHK(dev_realloc(k, &k->bufA, act));
        CHK(dev_realloc(k, &k->bufB, act));
        k->act_floats = act;
    }
    if (B > k->sideB) {
        CHK(dev_realloc(k, &k->d_fc1, (size_t)B * 512));
        CHK(dev_realloc(k, &k->d_fc2, (size_t)B * 512));
        CHK(dev_realloc(k, &k->d_fcpart, std::max((size_t)B * 17 * 512 + (size_t)B * 33 * 256, (size_t)B * 16 * 18 * 256)));
        k->sideB = B;
    }
    return 0;
}

// `beside` (may be null): work of the caller that depends on neither tower (the whole path's heat-map up-sampling and keypoint detection);
// it runs on this context's stream behind PosePrior -- i.e. beside ViewpointNet when the towers run on two streams -- and before the epilogue.
// `beside_side` (round 6, may be null): the part of that work that goes BEHIND ViewpointNet on the child stream instead when the towers run on two
// streams -- since ViewpointNet's tower got shorter (fc_tail, conv_s2_gemm: 0.206 -> 0.179 ms) the parent stream (PosePrior 0.15 + up-sampling 0.037 +
// keypoint detection 0.055) was the longer one; with the up-sampling on the child both are ~0.21 ms.  One stream: it simply runs after `beside`.
int run_pose3d(hp3d_ctx* ctx, const float* sm32, const float* hs, int B, int variant, const std::function<int(hipStream_t)>* beside = nullptr,
               const std::function<int(hipStream_t)>* beside_side = nullptr) {
    const int do_rot = (variant == HP3D_VARIANT_PROPOSED);
    const bool fused = !ctx->conv_naive && (ctx->use_lift_fused == 1 || (ctx->use_lift_fused < 0 && B <= 4)) &&
                       (size_t)4 * B * 32 * 32 * 64 <= ctx->act_floats;
    if (fused) {
        CHK(run_lift_fused(ctx, sm32, hs, B, variant == HP3D_VARIANT_BOTTLENECK, do_rot));
    } else {
        // The two towers share only their input (nets/ColorHandPose3DNetwork.py:231-235): 12 + 12 dependent launches of 13-25 us each that
        // leave most of the chip idle.  ViewpointNet runs on the child context's stream (own activations and partial sums, shared weights)
        // beside PosePrior on this one; the epilogue below waits for both.  Not while profiling per launch, replaying a graph, or when
        // this context is (or is busy with) the second stream of a two-stream call.
        bool side = false;
#ifndef HP3D_EMU
        side = do_rot && ctx->lift_overlap && !ctx->profiling && !ctx->use_graph && !ctx->conv_naive && !ctx->shared_weights &&
               !ctx->two_streams_live && kid_sync_state(ctx) == 0 && ensure_side_tower(ctx->kid, B) == 0;
        if (side) {
            hp3d_ctx* k = ctx->kid;
            HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
            HIPCHK(ctx, hipStreamWaitEvent(k->stream, ctx->ev_fork, 0));
            struct Join {       // recorded and waited for on every exit path (as in infer_full_chunked)
                hp3d_ctx *p, *k;
                ~Join() {
                    if (hipEventRecord(p->ev_join, k->stream) == hipSuccess) (void)hipStreamWaitEvent(p->stream, p->ev_join, 0);
                    else (void)hipStreamSynchronize(k->stream);
                }
            } join{ctx, k};
            const int rc = run_viewpoint(k, sm32, hs, B, ctx->d_u);
            if (rc != 0) { set_error(ctx, k->err.c_str()); return rc; }
            if (beside_side && ctx->kp_up_side) { CHK((*beside_side)(k->stream)); beside_side = nullptr; }
            CHK(run_poseprior_can(ctx, sm32, hs, B, variant == HP3D_VARIANT_BOTTLENECK, ctx->d_can));
            if (beside) { CHK((*beside)(ctx->stream)); beside = nullptr; }
            ++ctx->lift_overlap_calls;
        }
#endif
        if (!side) {
            CHK(run_poseprior_can(ctx, sm32, hs, B, variant == HP3D_VARIANT_BOTTLENECK, ctx->d_can));
            if (do_rot) CHK(run_viewpoint(ctx, sm32, hs, B, ctx->d_u));
        }
    }
    if (beside) CHK((*beside)(ctx->stream));
    if (beside_side) CHK((*beside_side)(ctx->stream));
    if (variant == HP3D_VARIANT_LOCAL)     // bone_rel_trafo_inv (nets/PosePriorNetwork.py:70-75)
        bone_rel_inv_launch(ctx->d_can, B, ctx->d_coord, ctx->stream);
    else
        lift_epilogue_launch(ctx->d_u, ctx->d_can, hs, B, ctx->d_rot, ctx->d_coord, do_rot, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

int need_nets(hp3d_ctx* ctx, int mask) {
    if (!ctx->blob) HP3D_FAIL(ctx, HP3D_ERR_WEIGHTS, "weights not finalized (call hp3d_finalize_weights)");
    if ((ctx->nets & mask) != mask)
        HP3D_FAIL(ctx, HP3D_ERR_WEIGHTS, "required network weights not loaded (have mask %d, need %d)", ctx->nets, mask);
    return 0;
}

int check_img(hp3d_ctx* ctx, int B, int H, int W) {
    if (B < 1 || H < 16 || W < 16)
        HP3D_FAIL(ctx, HP3D_ERR_ARG, "bad shape B=%d H=%d W=%d (need B>=1, H,W >=16)", B, H, W);
    if (mask_grow_lds_bytes(H, W) > 160 * 1024 - 1024)
        HP3D_FAIL(ctx, HP3D_ERR_ARG, "image %dx%d too large for the in-LDS mask growth", H, W);
    return 0;
}

int copy_in(hp3d_ctx* ctx, float* dst, const float* src, size_t n, bool dev) {
    HIPCHK(ctx, hipMemcpyAsync(dst, src, n * sizeof(float), dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                               ctx->stream));
    return 0;
}
int copy_out(hp3d_ctx* ctx, float* dst, const float* src, size_t n, bool dev) {
    if (!dst) return 0;
    HIPCHK(ctx, hipMemcpyAsync(dst, src, n * sizeof(float), dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                               ctx->stream));
    return 0;
}

// stages 2-8 of the full path on device-resident image/hand_side
int run_detect_and_crop(hp3d_ctx* ctx, const float* d_image, int B, int H, int W, int want_mask, bool image_hot = false) {
    CHK(run_handsegnet(ctx, d_image, B, H, W, image_hot));
    MaskBuffers mb{ctx->d_keys, ctx->d_det, nullptr};
    {
        ProfScope ps(ctx, "seg_upsample_softmax", "seg_upsample_softmax", 0.0, 4.0 * B * H * W * 2 + (double)B * H * W);
        seg_upsample_softmax_launch(ctx->d_segsmall, B, H / 8, W / 8, 32, H, W, ctx->d_large, mb, ctx->stream);
    }
    {
        ProfScope ps(ctx, "mask_grow", "mask_grow", 0.0, (double)B * H * W);
        mask_grow_launch(mb, B, H, W, ctx->empty_fltmax, want_mask ? ctx->d_mask : nullptr, ctx->d_center,
                         ctx->d_cropsize, ctx->d_scale, ctx->d_seed, ctx->stream);
    }
    {
        ProfScope ps(ctx, "crop_and_resize", "crop_and_resize", 0.0, 4.0 * B * (H * W * 3 + 256 * 256 * 3));
        crop_and_resize_launch(d_image, B, H, W, 3, ctx->d_center, ctx->d_scale, 256, ctx->d_crop, ctx->stream);
    }
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

// detect_keypoints + trafo_coords (utils/general.py:331-357; run.py:72-73, eval2d.py:93-94) on the 32 x 32 maps of the
// last PoseNet2D run: no 256 x 256 x 21 heat-map has to exist, let alone travel to the host
int run_kp_detect(hp3d_ctx* ctx, int B, int32_t* kp_crop, double* kp_image, bool dev) {
    ProfScope ps(ctx, "kp_detect", "kp_detect", 0.0, 4.0 * B * 32 * 32 * 21);
    kp_detect_launch(ctx->d_sm[2], B, 32, 32, 21, 32, 256, 256, ctx->d_scale, ctx->d_center,
                     dev && kp_crop ? kp_crop : ctx->d_kpcrop, dev && kp_image ? kp_image : ctx->d_kpimg, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    if (!dev) {
        if (kp_crop) HIPCHK(ctx, hipMemcpyAsync(kp_crop, ctx->d_kpcrop, sizeof(int32_t) * (size_t)B * 42, hipMemcpyDeviceToHost, ctx->stream));
        if (kp_image) HIPCHK(ctx, hipMemcpyAsync(kp_image, ctx->d_kpimg, sizeof(double) * (size_t)B * 42, hipMemcpyDeviceToHost, ctx->stream));
    }
    return 0;
}

int infer_full_impl(hp3d_ctx* ctx, int B, int H, int W, const float* image, const float* hand_side,
                    float* hand_scoremap, float* image_crop, float* scale_crop, float* center, float* kp_scoremap,
                    float* coord3d, float* hand_mask, bool dev, const unsigned char* image_u8 = nullptr, int Hin = 0,
                    int Win = 0, int32_t* kp_crop = nullptr, double* kp_image = nullptr, bool no_sync = false) {
    if (!ctx) return HP3D_ERR_ARG;
    if ((!image && !image_u8) || !hand_side) HP3D_FAIL(ctx, HP3D_ERR_ARG, "image / hand_side is NULL");
    CHK(check_img(ctx, B, H, W));
    CHK(need_nets(ctx, NET_SEG | NET_POSE | NET_PRIOR | NET_VP));
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CHK(ensure_arena(ctx, B, H, W));
    if (ctx->profiling != 2) prof_reset(ctx);   // mode 2 accumulates across calls
    const float* d_img = image;
    const float* d_hs = hand_side;
    if (image_u8) {          // uint8 frame in, x/255-0.5 + resize to the net size on device
        if (Hin < 2 || Win < 2) HP3D_FAIL(ctx, HP3D_ERR_ARG, "bad uint8 frame size %dx%d", Hin, Win);
        const size_t nb = (size_t)B * Hin * Win * 3;
        if (nb > ctx->u8_bytes) { CHK(dev_realloc(ctx, &ctx->d_u8, nb)); ctx->u8_bytes = nb; }
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_u8, image_u8, nb, hipMemcpyHostToDevice, ctx->stream));
        preprocess_u8_launch(ctx->d_u8, B, Hin, Win, H, W, ctx->d_image, ctx->stream);
        CHK(copy_in(ctx, ctx->d_hs, hand_side, (size_t)B * 2, false));
        d_img = ctx->d_image; d_hs = ctx->d_hs;
    } else if (!dev) {
        CHK(copy_in(ctx, ctx->d_image, image, (size_t)B * H * W * 3, false));
        CHK(copy_in(ctx, ctx->d_hs, hand_side, (size_t)B * 2, false));
        d_img = ctx->d_image; d_hs = ctx->d_hs;
    }
    CHK(run_detect_and_crop(ctx, d_img, B, H, W, hand_mask != nullptr, image_u8 != nullptr));     // (the uint8 front end has just written the image)
    CHK(run_posenet(ctx, ctx->d_crop, B, 256, 256, true));
    // (the heat-map up-sampling and the keypoint detection read PoseNet2D's last score map like the lifting towers do and depend on neither:
    //  they run beside ViewpointNet when the towers take two streams, option "lift_overlap")
    const std::function<int(hipStream_t)> kp_up = [&](hipStream_t st) -> int {
        if (kp_scoremap) {
            ProfScope ps(ctx, "kp_upsample", "resize_bilinear", 0.0, 4.0 * B * (32 * 32 * 21 + 256 * 256 * 21));
            resize_bilinear_launch(ctx->d_sm[2], B, 32, 32, 21, 32, 256, 256, dev ? kp_scoremap : ctx->d_kpmap, st);
        }
        return 0;
    };
    const std::function<int(hipStream_t)> kp_work = [&](hipStream_t) -> int {
        if (kp_crop || kp_image) CHK(run_kp_detect(ctx, B, kp_crop, kp_image, dev));
        return 0;
    };
    CHK(run_pose3d(ctx, ctx->d_sm[2], d_hs, B, HP3D_VARIANT_PROPOSED, &kp_work, &kp_up));
    CHK(copy_out(ctx, hand_scoremap, ctx->d_large, (size_t)B * H * W * 2, dev));
    CHK(copy_out(ctx, image_crop, ctx->d_crop, (size_t)B * 256 * 256 * 3, dev));
    CHK(copy_out(ctx, scale_crop, ctx->d_scale, (size_t)B, dev));
    CHK(copy_out(ctx, center, ctx->d_center, (size_t)B * 2, dev));
    if (!dev) CHK(copy_out(ctx, kp_scoremap, ctx->d_kpmap, (size_t)B * 256 * 256 * 21, false));
    CHK(copy_out(ctx, coord3d, ctx->d_coord, (size_t)B * 63, dev));
    CHK(copy_out(ctx, hand_mask, ctx->d_mask, (size_t)B * H * W, dev));
    if (!dev && !no_sync) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// Large batches run as consecutive chunks of micro_batch images: per-layer activations of one chunk (<= 0.84 GB at
// 320x320) stay closer to the caches and every tensor stays inside the 32-bit offsets of the Winograd kernel.  Measured:
// f32 320x320 B=32 1452 img/s, B=128 unsplit 1355, split 4 x 32 1460; f16 480x640 B=128 unsplit 2316, split 2072 --
// hence auto = at most 32 for float32, no split for the half-precision mode.  The float32 limit also follows the
// resolution: the largest activation of the path (conv1_1 / conv1_2: H x W x 64 floats per image) must stay below 2^31
// bytes per chunk or those layers fall off the Winograd kernel (480x640: 27 images), and the chunks are balanced
// (B = 32 at 480x640 -> 16 + 16, not 27 + 5) so that no chunk under-fills the persistent grids.
int auto_micro_batch(const hp3d_ctx* ctx, int B, int H, int W) {
    if (ctx->micro_batch >= 0) return ctx->micro_batch;
    if (ctx->prec) return 0;
    const long per_img = (long)H * W * 64 * 4;
    int lim = (int)std::min<long>(32, ((1L << 31) - 1) / std::max<long>(per_img, 1));
    if (lim < 1) lim = 1;
    if (B <= lim) return lim;
    // 32 images fill every persistent grid of the path at the reference's sizes (exactly 256 conv_wino7 items, whole rounds of conv_wino4
    // items on most layers): chunks of 32 and a remainder beat balanced chunks there (round 5, 320x320: B = 48 as 24 + 24: 20.16 ms;
    // as 32 + 16: 11.99 + 6.70).  Where the 32-bit offsets allow fewer than 32 images per chunk the chunks stay balanced.
    // (ADVICE r5 asked whether a small remainder -- B = 33 ... 40 -- should rather be balanced: measured in round 6, B = 40 at 320x320 as 32 + 8:
    //  16.09 ms, as 20 + 20: 16.65 ms.  The full chunk keeps its rate; what helps such batches is the second stream, see infer_full_streams.)
    if (lim == 32) return 32;
    const int chunks = (B + lim - 1) / lim;
    return (B + chunks - 1) / chunks;
}

int infer_full_chunked1(hp3d_ctx* ctx, int B, int H, int W, const float* image, const float* hand_side,
                       float* hand_scoremap, float* image_crop, float* scale_crop, float* center, float* kp_scoremap,
                       float* coord3d, float* hand_mask, bool dev, const unsigned char* image_u8 = nullptr, int Hin = 0,
                       int Win = 0, int32_t* kp_crop = nullptr, double* kp_image = nullptr, bool no_sync = false) {
    if (!ctx) return HP3D_ERR_ARG;
    const int mb = auto_micro_batch(ctx, B, H, W);
    if (mb <= 0 || B <= mb)
        return infer_full_impl(ctx, B, H, W, image, hand_side, hand_scoremap, image_crop, scale_crop, center, kp_scoremap,
                               coord3d, hand_mask, dev, image_u8, Hin, Win, kp_crop, kp_image, no_sync);
    CHK(check_img(ctx, B, H, W));
    if (!hand_side) HP3D_FAIL(ctx, HP3D_ERR_ARG, "image / hand_side is NULL");
    const int saved_prof = ctx->profiling;
    int rc = 0;
    for (int b0 = 0; b0 < B && rc == 0; b0 += mb) {
        const int nb = std::min(mb, B - b0);
        auto off = [&](float* p, size_t per) { return p ? p + (size_t)b0 * per : nullptr; };
        if (b0 > 0 && saved_prof == 1) ctx->profiling = 2;          // one call = one profile: keep the earlier chunks
        rc = infer_full_impl(ctx, nb, H, W, image ? image + (size_t)b0 * H * W * 3 : nullptr, hand_side + (size_t)b0 * 2,
                             off(hand_scoremap, (size_t)H * W * 2), off(image_crop, 256 * 256 * 3), off(scale_crop, 1),
                             off(center, 2), off(kp_scoremap, 256 * 256 * 21), off(coord3d, 63), off(hand_mask, (size_t)H * W),
                             dev, image_u8 ? image_u8 + (size_t)b0 * Hin * Win * 3 : nullptr, Hin, Win,
                             kp_crop ? kp_crop + (size_t)b0 * 42 : nullptr, kp_image ? kp_image + (size_t)b0 * 42 : nullptr, no_sync);
    }
    ctx->profiling = saved_prof;
    return rc;
}

// Two HIP streams per GPU: the batch is cut in two halves that run the whole path concurrently, each on its own stream and
// arena (the second one owned by a child context that shares the weight blob).  Every image is independent, and the
// persistent Winograd kernels leave CUs idle in their last round of work items (1600 items on 256 CUs = 6.25 rounds) and
// between dependent launches: the other half's kernels fill exactly those holes.  Measured (profiles/r02_tuning_notes.md):
// float32 B=32 320x320 1558 -> 1615 img/s, 240x320 +1.8 %, 480x640 +6.9 %, B=16 +4.5 %, B=64 +6.4 %, f16 trunks +2.1 %; four
// streams of 8 images lose 8 % (too few items per launch); round 3's "auto" was 2 streams from 16 images per call; round 4: see `ns` below.
// Per-launch profiling and hipGraph replay keep one stream.  A half-batch may take another kernel plan than the whole
// batch would (small-batch Winograd split-K): results equal the one-stream run to rounding, discrete decisions included.
int kid_sync_state(hp3d_ctx* ctx) {
#ifdef HP3D_EMU
    return -1;
#else
    if (!ctx->kid) {
        hp3d_ctx* k = new hp3d_ctx();
        k->device = ctx->device;
        if (hipStreamCreateWithFlags(&k->stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            delete k;
            return -1;
        }
        k->shared_weights = true;
        ctx->kid = k;
    }
    hp3d_ctx* k = ctx->kid;
    k->blob = ctx->blob; k->blob16 = ctx->blob16; k->nets = ctx->nets; k->prec = ctx->prec;
    k->empty_fltmax = ctx->empty_fltmax; k->conv_naive = ctx->conv_naive; k->use_wino = ctx->use_wino;
    k->use_first = ctx->use_first; k->first_touch = ctx->first_touch; k->first_balanced = ctx->first_balanced; k->use_wino2 = ctx->use_wino2; k->use_wino4 = ctx->use_wino4; k->fc_tail = ctx->fc_tail; k->tiny_gemm = ctx->tiny_gemm; k->use_wino4s = ctx->use_wino4s; k->w4_tail = ctx->w4_tail; k->use_wino7 = ctx->use_wino7; k->wino7_ksplit = ctx->wino7_ksplit; k->use_pw2 = ctx->use_pw2; k->use_lift_fused = ctx->use_lift_fused; k->use_h16 = ctx->use_h16; k->h16_k7k1 = ctx->h16_k7k1; k->fuse12 = ctx->fuse12; k->wino_splitk = ctx->wino_splitk; k->micro_batch = ctx->micro_batch;
    k->nstreams = 1; k->profiling = 0; k->use_graph = 0;
    return 0;
#endif
}

int infer_full_chunked(hp3d_ctx* ctx, int B, int H, int W, const float* image, const float* hand_side,
                       float* hand_scoremap, float* image_crop, float* scale_crop, float* center, float* kp_scoremap,
                       float* coord3d, float* hand_mask, bool dev, const unsigned char* image_u8 = nullptr, int Hin = 0,
                       int Win = 0, int32_t* kp_crop = nullptr, double* kp_image = nullptr) {
    if (!ctx) return HP3D_ERR_ARG;
    // "auto": two streams when each half still fills the persistent grids -- from 2.4 M input pixels per half (3 M until round 6).  Round 4 (tail pieces took the
    // last-round quantisation out of the one-stream run; same box, float32, images/s one / two streams): B=32 320x320 2190 / 2115,
    // B=16 1975 / 1787, B=8 1666 / 1372, B=32 240x320 2454 / 2408 -- one stream; B=64 320x320 (two chunks of 32) 2195 / 2292, B=32 480x640
    // 989 / 1051, f16 B=128 480x640 3234 / 3321 -- two (profiles/r04_tuning_notes.md).
    // Round 6: also between one and two full chunks (32 < B < 64 in float32 mode), where one stream runs a full chunk and then a latency-bound
    // remainder: B = 40 at 320x320 2487 (32 + 8) -> 2616 images/s, B = 48 2591 -> 2622, 240x320 B = 48 2945 -> 3012 (profiles/r06_batch_sweep.txt).
    // (From 40 images: B = 36 as 18 + 18 on two streams 2151, as 32 beside 4 2466, on one stream ~2430 -- halves below 20 images are too small.
    //  A full chunk BESIDE the remainder instead of two halves loses from B = 40 on: 2502 / 2587 against 2609 / 2617 at B = 40 / 48.)
    const bool between = !ctx->prec && ctx->micro_batch < 0 && B >= 40 && B < 64 && (long)H * W <= 320L * 320L;
    const int ns = ctx->nstreams >= 0 ? ctx->nstreams : (((long)(B / 2) * H * W >= 2400000L || between) ? 2 : 1);        // (2.4 M: B = 64 at 240x320 3048 -> 3132)
    if (ns < 2 || B < 2 || ctx->profiling || ctx->use_graph || ctx->conv_naive || ctx->shared_weights || !hand_side ||
        (!image && !image_u8) || kid_sync_state(ctx) != 0)
        return infer_full_chunked1(ctx, B, H, W, image, hand_side, hand_scoremap, image_crop, scale_crop, center, kp_scoremap,
                                   coord3d, hand_mask, dev, image_u8, Hin, Win, kp_crop, kp_image);
#ifndef HP3D_EMU
    hp3d_ctx* k = ctx->kid;
    const int b0 = (B + 1) / 2, b1 = B - b0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // the child's stream starts after everything already queued on the parent's (inputs produced there) ...
    HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(k->stream, ctx->ev_fork, 0));
    auto off = [&](float* p, size_t per) { return p ? p + (size_t)b0 * per : nullptr; };
    // Whatever happens below, the parent's stream must not run ahead of (and hp3d_sync / hp3d_dev_free / hp3d_destroy must
    // cover) what was already enqueued on the child's stream: the join is recorded and waited for on EVERY exit path.
    ctx->two_streams_live = k->two_streams_live = true;
    struct Join {
        hp3d_ctx *p, *k;
        ~Join() {
            p->two_streams_live = k->two_streams_live = false;
            if (hipEventRecord(p->ev_join, k->stream) == hipSuccess) (void)hipStreamWaitEvent(p->stream, p->ev_join, 0);
            else (void)hipStreamSynchronize(k->stream);
        }
    } join{ctx, k};
    // (host buffers: the child's pageable device->host copies block the host, so the two halves overlap on the `_dev`
    //  entry points only -- the ones bench.py and dist.py use)
    int rc = infer_full_chunked1(k, b1, H, W, image ? image + (size_t)b0 * H * W * 3 : nullptr, hand_side + (size_t)b0 * 2,
                                 off(hand_scoremap, (size_t)H * W * 2), off(image_crop, 256 * 256 * 3), off(scale_crop, 1),
                                 off(center, 2), off(kp_scoremap, 256 * 256 * 21), off(coord3d, 63), off(hand_mask, (size_t)H * W),
                                 dev, image_u8 ? image_u8 + (size_t)b0 * Hin * Win * 3 : nullptr, Hin, Win,
                                 kp_crop ? kp_crop + (size_t)b0 * 42 : nullptr, kp_image ? kp_image + (size_t)b0 * 42 : nullptr, true);
    if (rc != 0) { set_error(ctx, k->err.c_str()); }
    const int rc0 = rc != 0 ? rc : infer_full_chunked1(ctx, b0, H, W, image, hand_side, hand_scoremap, image_crop, scale_crop, center,
                                                       kp_scoremap, coord3d, hand_mask, dev, image_u8, Hin, Win, kp_crop, kp_image, true);
    if (rc0 != 0) return rc0;
    // ... and the parent's stream continues only after the child's half is done (host buffers: wait for both here)
    if (!dev) {
        HIPCHK(ctx, hipStreamSynchronize(k->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return 0;
#else
    return HP3D_ERR_UNSUPPORTED;
#endif
}

int posenet_impl(hp3d_ctx* ctx, int B, int H, int W, const float* image_crop, float* s0, float* s1, float* s2, bool dev) {
    if (!ctx) return HP3D_ERR_ARG;
    if (!image_crop) HP3D_FAIL(ctx, HP3D_ERR_ARG, "image_crop is NULL");
    CHK(check_img(ctx, B, H, W));
    CHK(need_nets(ctx, NET_POSE));
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CHK(ensure_arena(ctx, B, H, W));
    if (ctx->profiling != 2) prof_reset(ctx);   // mode 2 accumulates across calls
    const float* d_img = image_crop;
    if (!dev) {
        CHK(copy_in(ctx, ctx->d_image, image_crop, (size_t)B * H * W * 3, false));
        d_img = ctx->d_image;
    }
    CHK(run_posenet(ctx, d_img, B, H, W));
    const int hs = H / 8, ws = W / 8, npix = B * hs * ws;
    float* outs[3] = {s0, s1, s2};
    for (int i = 0; i < 3; ++i) {
        if (!outs[i]) continue;
        float* dst = dev ? outs[i] : ctx->bufA;   // compact [npix,21] staging
        copy_channels_launch(ctx->d_sm[i], npix, 21, 32, dst, 21, ctx->stream);
        if (!dev) {
            CHK(copy_out(ctx, outs[i], ctx->bufA, (size_t)npix * 21, false));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    HIPCHK(ctx, hipGetLastError());
    if (!dev) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// scratch helpers for the per-op entry points (host in -> device -> host out)
struct Scratch {
    hp3d_ctx* ctx;
    std::vector<void*> ptrs;
    explicit Scratch(hp3d_ctx* c) : ctx(c) {}
    ~Scratch() {
        for (void* p : ptrs) hipFree(p);
    }
    template <typename T>
    T* alloc(size_t n) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return (T*)p;
    }
    template <typename T>
    T* upload(const T* h, size_t n) {
        T* d = alloc<T>(n);
        if (d && hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return nullptr;
        return d;
    }
};
#define NN(ctx, p)                                                         \
    do {                                                                   \
        if (!(p)) HP3D_FAIL(ctx, HP3D_ERR_NOMEM, "scratch allocation/upload failed: %s", #p); \
    } while (0)

int finish_op(hp3d_ctx* ctx) {
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return check_lift_error(ctx);
}

}  // namespace

// ================================================================================================
// hp3d_set_option("graph", "1"): the device-pointer entry points replay their whole launch sequence (~85 launches for
// the full path) as one hipGraph once the same call (shape + pointers) is seen for the third time: call 1 runs normally
// (allocations, function attributes), call 2 is captured while it is enqueued, later calls are a single
// hipGraphLaunch.  Anything the sequence depends on (arena re-allocation, weights, options, profiling) bumps
// graph_epoch and drops the captured graphs.  Meant for small batches, where launch gaps are ~15 % of the time.
template <class F>
int run_graphed(hp3d_ctx* ctx, const std::string& key, F&& enqueue) {
#ifdef HP3D_EMU
    return enqueue();
#else
    if (!ctx->use_graph || ctx->profiling) return enqueue();
    hp3d_ctx::GraphEntry& g = ctx->graphs[key];
    if (g.exec && g.epoch != ctx->graph_epoch) { hipGraphExecDestroy(g.exec); g.exec = nullptr; g.calls = 0; }
    if (g.exec) {
        HIPCHK(ctx, hipGraphLaunch(g.exec, ctx->stream));
        ++ctx->graph_replays;
        return 0;
    }
    if (g.calls < 0 || g.calls++ == 0) return enqueue();                  // warm-up (or capture failed earlier)
    const long epoch0 = ctx->graph_epoch;
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { g.calls = -1; return enqueue(); }
    const int rc = enqueue();
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
    if (rc != 0 || e != hipSuccess || !graph || epoch0 != ctx->graph_epoch) {          // not capturable: stay on plain launches
        if (getenv("HP3D_GRAPH_DEBUG")) fprintf(stderr, "hp3d graph: capture of %s failed (rc %d, %s, epoch %ld -> %ld)\n", key.c_str(), rc, hipGetErrorString(e), epoch0, ctx->graph_epoch);
        if (graph) hipGraphDestroy(graph);
        (void)hipGetLastError();
        g.calls = -1;
        return rc != 0 ? rc : enqueue();
    }
    const hipError_t ei = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (ei != hipSuccess) { g.exec = nullptr; g.calls = -1; (void)hipGetLastError(); return enqueue(); }
    g.epoch = ctx->graph_epoch;
    if (getenv("HP3D_GRAPH_DEBUG")) fprintf(stderr, "hp3d graph: captured %s\n", key.c_str());
    ++ctx->graph_captures;
    HIPCHK(ctx, hipGraphLaunch(g.exec, ctx->stream));
    ++ctx->graph_replays;
    return 0;
#endif
}
static std::string graph_key(const char* what, std::initializer_list<long> dims, std::initializer_list<const void*> ptrs) {
    char buf[64];
    std::string k(what);
    for (long d : dims) { snprintf(buf, sizeof buf, "|%ld", d); k += buf; }
    for (const void* p : ptrs) { snprintf(buf, sizeof buf, "|%p", p); k += buf; }
    return k;
}

static long long comm_ranks(hp3d_ctx* ctx);     // ranks of the live RCCL communicator (ncclCommCount), 0 = none

extern "C" {

int hp3d_abi_version(void) { return 1; }

int hp3d_device_count(int* count) {
    if (!count) return HP3D_ERR_ARG;
    *count = 0;
#ifdef HP3D_EMU
    *count = 1;             // the CPU interpreter plays one device
    return 0;
#else
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return HP3D_ERR_HIP; }
    *count = n;
    return 0;
#endif
}

int hp3d_device_pci_bus_id(int device, char* buf, int cap) {
    if (!buf || cap < 13) return HP3D_ERR_ARG;
    buf[0] = 0;
#ifdef HP3D_EMU
    (void)device;
    return HP3D_ERR_UNSUPPORTED;          // the CPU interpreter's one device sits on no bus
#else
    if (hipDeviceGetPCIBusId(buf, cap, device) != hipSuccess) { (void)hipGetLastError(); buf[0] = 0; return HP3D_ERR_HIP; }
    return 0;
#endif
}

int hp3d_create(int device, hp3d_ctx** out) {
    if (!out) return HP3D_ERR_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error(nullptr, "hp3d_create: no HIP device visible (the engine has no CPU fallback)");
        return HP3D_ERR_HIP;
    }
    if (device < 0 || device >= n) {
        set_error(nullptr, "hp3d_create: device index out of range");
        return HP3D_ERR_ARG;
    }
    hp3d_ctx* ctx = new hp3d_ctx();
    ctx->device = device;
    ctx->use_wino4 = wino4_default();
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&ctx->stream) != hipSuccess) {
        set_error(nullptr, "hp3d_create: hipSetDevice/hipStreamCreate failed");
        delete ctx;
        return HP3D_ERR_HIP;
    }
    *out = ctx;
    return 0;
}

int hp3d_destroy(hp3d_ctx* ctx) {
    if (!ctx) return HP3D_ERR_ARG;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
#ifndef HP3D_EMU
    if (ctx->copy_stream) hipStreamSynchronize(ctx->copy_stream);
#endif
    if (ctx->kid) { hp3d_destroy(ctx->kid); ctx->kid = nullptr; }
#ifndef HP3D_EMU
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
#endif
    if (ctx->shared_weights) { ctx->blob = nullptr; ctx->blob16 = nullptr; }     // owned by the parent context
    float** fp[] = {&ctx->blob, &ctx->bufA, &ctx->bufB, &ctx->col, &ctx->d_image, &ctx->d_hs, &ctx->d_large,
                    &ctx->d_crop, &ctx->d_center, &ctx->d_scale, &ctx->d_cropsize, &ctx->d_kpmap, &ctx->d_coord,
                    &ctx->d_mask, &ctx->d_segsmall, &ctx->d_concat, &ctx->d_sm[0], &ctx->d_sm[1], &ctx->d_sm[2],
                    &ctx->d_can, &ctx->d_rot, &ctx->d_u, &ctx->d_fc1, &ctx->d_fc2, &ctx->d_fg,
                    &ctx->d_pooled, &ctx->d_fcpart};
    for (float** p : fp)
        if (*p) hipFree(*p);
    if (ctx->d_kpimg) hipFree(ctx->d_kpimg);
    if (ctx->d_gather) hipFree(ctx->d_gather);
#ifndef HP3D_EMU
    if (ctx->upload_done) hipEventDestroy(ctx->upload_done);
    if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
#endif
    if (ctx->d_kpcrop) hipFree(ctx->d_kpcrop);
#ifndef HP3D_EMU
    for (auto& kv : ctx->graphs)
        if (kv.second.exec) hipGraphExecDestroy(kv.second.exec);
#endif
    if (ctx->comm) hp3d_comm_destroy(ctx);
    if (ctx->d_seed) hipFree(ctx->d_seed);
    if (ctx->d_liftbar) hipFree(ctx->d_liftbar);
#ifndef HP3D_EMU
    if (ctx->h_lifterr) hipHostFree(ctx->h_lifterr);
#endif
    if (ctx->d_keys) hipFree(ctx->d_keys);
    if (ctx->d_det) hipFree(ctx->d_det);
    if (ctx->d_u8) hipFree(ctx->d_u8);
    if (ctx->blob16) hipFree(ctx->blob16);
    if (ctx->d_concat16) hipFree(ctx->d_concat16);
    for (auto& kv : ctx->naive_w) hipFree(kv.second);
    for (hipEvent_t e : ctx->event_pool) hipEventDestroy(e);
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

const char* hp3d_last_error(hp3d_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }
void* hp3d_stream(hp3d_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
int hp3d_sync(hp3d_ctx* ctx) {
    if (!ctx) return HP3D_ERR_ARG;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return check_lift_error(ctx);
}

// ---- device / pinned-host memory for callers that hold their batches in HBM (bench.py, dist.py): the C ABI needs no
//      PyTorch for that.  Copies are ordered on the engine stream (hp3d_memcpy blocks until done); hp3d_upload_async
//      runs on a second stream so that the H2D of batch n+1 overlaps the kernels of batch n, hp3d_wait_upload makes
//      the engine stream wait for the last upload.
int hp3d_dev_alloc(hp3d_ctx* ctx, size_t bytes, void** out) {
    if (!ctx || !out) return HP3D_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { (void)hipGetLastError(); HP3D_FAIL(ctx, HP3D_ERR_NOMEM, "hipMalloc(%zu) failed", bytes); }
    *out = p;
    return 0;
}
int hp3d_dev_free(hp3d_ctx* ctx, void* p) {
    if (!ctx) return HP3D_ERR_ARG;
    if (!p) return 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->copy_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));    // a pending hp3d_upload_async may target it
    if (ctx->kid) HIPCHK(ctx, hipStreamSynchronize(ctx->kid->stream));
    HIPCHK(ctx, hipFree(p));
    return 0;
}
#ifndef HP3D_EMU
int hp3d_host_alloc(hp3d_ctx* ctx, size_t bytes, void** out) {
    if (!ctx || !out) return HP3D_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); HP3D_FAIL(ctx, HP3D_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes); }
    *out = p;
    return 0;
}
int hp3d_host_free(hp3d_ctx* ctx, void* p) {
    if (!ctx) return HP3D_ERR_ARG;
    if (!p) return 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->copy_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));    // an upload may still be reading the buffer
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipHostFree(p));
    return 0;
}
int hp3d_upload_async(hp3d_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    if (!ctx || !dst_dev || !src_host) return HP3D_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->copy_stream) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->upload_done, hipEventDisableTiming));
    }
    HIPCHK(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
    HIPCHK(ctx, hipEventRecord(ctx->upload_done, ctx->copy_stream));
    return 0;
}
int hp3d_wait_upload(hp3d_ctx* ctx) {
    if (!ctx) return HP3D_ERR_ARG;
    if (ctx->upload_done) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->upload_done, 0));
    return 0;
}
#else
int hp3d_host_alloc(hp3d_ctx* ctx, size_t bytes, void** out) { if (!ctx || !out) return HP3D_ERR_ARG; *out = malloc(bytes ? bytes : 1); return *out ? 0 : HP3D_ERR_NOMEM; }
int hp3d_host_free(hp3d_ctx* ctx, void* p) { if (!ctx) return HP3D_ERR_ARG; free(p); return 0; }
int hp3d_upload_async(hp3d_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    if (!ctx || !dst_dev || !src_host) return HP3D_ERR_ARG;
    memcpy(dst_dev, src_host, bytes);
    return 0;
}
int hp3d_wait_upload(hp3d_ctx* ctx) { return ctx ? 0 : HP3D_ERR_ARG; }
#endif
int hp3d_memcpy(hp3d_ctx* ctx, void* dst, const void* src, size_t bytes, int kind) {
    if (!ctx || !dst || !src || kind < 0 || kind > 2) return HP3D_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, k, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int hp3d_set_option(hp3d_ctx* ctx, const char* key, const char* value) {
    if (!ctx || !key || !value) return HP3D_ERR_ARG;
    const std::string k(key), v(value);
    ++ctx->graph_epoch;             // captured launch sequences may depend on any option
    if (k == "empty_reduce" && (v == "inf" || v == "fltmax")) { ctx->empty_fltmax = (v == "fltmax"); return 0; }
    if (k == "wino_splitk" && (v == "0" || v == "1")) { ctx->wino_splitk = v == "1"; return 0; }
    if (k == "wino4_tail" && (v == "0" || v == "1")) { ctx->w4_tail = v == "1"; ++ctx->graph_epoch; return 0; }
    if (k == "first_touch_beside" && (v == "0" || v == "1")) { ctx->first_touch_beside = v == "1"; return 0; }
    if (k == "first_touch" && (v == "0" || v == "1" || v == "auto")) { ctx->first_touch = v == "auto" ? -1 : v == "1"; return 0; }
    if (k == "first_walk" && (v == "balanced" || v == "rows")) { ctx->first_balanced = v == "balanced"; return 0; }
    if (k == "lift_overlap" && (v == "0" || v == "1")) { ctx->lift_overlap = v == "1"; return 0; }
    if (k == "lift_fused" && (v == "0" || v == "1" || v == "auto")) { ctx->use_lift_fused = v == "auto" ? -1 : v == "1" ? 1 : 0; return 0; }
    if (k == "wino4" && (v == "0" || v == "1" || v == "pose" || v == "auto" || v == "all")) {
        ctx->use_wino4 = v == "auto" ? wino4_default() : v == "all" ? -2 : v == "1" ? 1 : v == "pose" ? -1 : 0;
        return 0;
    }
    if (k == "wino4_split" && (v == "0" || v == "1" || v == "auto")) { ctx->use_wino4s = v == "auto" ? -1 : v == "1" ? 1 : 0; return 0; }
    if (k == "fc_tail" && (v == "0" || v == "1")) { ctx->fc_tail = v == "1"; return 0; }
    if (k == "kp_up_side" && (v == "0" || v == "1")) { ctx->kp_up_side = v == "1"; return 0; }
    if (k == "tiny_gemm" && (v == "0" || v == "1")) { ctx->tiny_gemm = v == "1"; return 0; }
    if (k == "pw2" && (v == "0" || v == "1" || v == "force")) { ctx->use_pw2 = v == "0" ? 0 : v == "1" ? 1 : 2; return 0; }
    if (k == "wino7_ksplit") { ctx->wino7_ksplit = v == "auto" ? 0 : std::max(0, atoi(v.c_str())); return 0; }
    if (k == "wino7" && (v == "0" || v == "1" || v == "auto")) { ctx->use_wino7 = v == "auto" ? -1 : v == "1" ? 1 : 0; return 0; }
    if (k == "wino2" && (v == "0" || v == "1" || v == "auto")) { ctx->use_wino2 = v == "auto" ? -1 : v == "1" ? 1 : 0; return 0; }
    if (k == "f16_fuse12" && (v == "0" || v == "1" || v == "ring" || v == "resident")) {
        ctx->fuse12 = v == "0" ? 0 : v == "1" ? 1 : v == "ring" ? 2 : 3; ++ctx->graph_epoch; return 0;
    }
    if (k == "f16_k7k1" && (v == "0" || v == "1")) { ctx->h16_k7k1 = v == "1"; return 0; }
    if (k == "f16_impl" && (v == "h16" || v == "mfma" || v == "h16_force")) { ctx->use_h16 = v == "mfma" ? 0 : v == "h16" ? 1 : 2; return 0; }
    if (k == "streams" && (v == "1" || v == "2" || v == "auto")) { ctx->nstreams = v == "auto" ? -1 : v == "2" ? 2 : 1; return 0; }
    if (k == "conv_impl" && (v == "mfma" || v == "naive" || v == "direct" || v == "winograd")) {
        ctx->conv_naive = (v == "naive");
        ctx->use_wino = (v == "direct" || v == "naive") ? 0 : (v == "winograd") ? 2 : 1;   // mfma = auto
        ctx->use_first = (v == "direct" || v == "naive") ? 0 : 1;
        return 0;
    }
    if (k == "graph" && (v == "0" || v == "1")) {
#ifdef HP3D_EMU
        if (v == "1") HP3D_FAIL(ctx, HP3D_ERR_UNSUPPORTED, "no hipGraph in the CPU interpreter build");
#endif
        ctx->use_graph = v == "1";
        return 0;
    }
    if (k == "micro_batch" && v == "auto") { ctx->micro_batch = -1; return 0; }
    if (k == "micro_batch") {
        char* end = nullptr;
        const long n = strtol(value, &end, 10);
        if (end == value || *end || n < 0 || n > (1 << 20)) HP3D_FAIL(ctx, HP3D_ERR_ARG, "micro_batch wants a non-negative integer, got %s", value);
        ctx->micro_batch = (int)n;
        return 0;
    }
    HP3D_FAIL(ctx, HP3D_ERR_ARG, "unknown option %s=%s", key, value);
}

int hp3d_set_weight(hp3d_ctx* ctx, const char* tf_var_name, const float* data, const int64_t* shape, int rank) {
    if (!ctx || !tf_var_name || !data || !shape || rank < 1 || rank > 4) return HP3D_ERR_ARG;
    std::string name(tf_var_name);
    const size_t slash = name.rfind('/');
    if (slash == std::string::npos) HP3D_FAIL(ctx, HP3D_ERR_ARG, "bad variable name %s", tf_var_name);
    const std::string layer = name.substr(0, slash), kind = name.substr(slash + 1);
    if (kind != "weights" && kind != "biases") HP3D_FAIL(ctx, HP3D_ERR_ARG, "bad variable name %s", tf_var_name);
    std::vector<int64_t> want;
    bool known = false;
    auto ci = ctx->T.conv_idx.find(layer);
    if (ci != ctx->T.conv_idx.end()) {
        const ConvL& l = ctx->T.conv[ci->second];
        want = (kind == "weights") ? std::vector<int64_t>{l.k, l.k, l.cin, l.cout} : std::vector<int64_t>{l.cout};
        known = true;
    } else if (layer == "ViewpointNet/fc_vp_ux" || layer == "ViewpointNet/fc_vp_uy" || layer == "ViewpointNet/fc_vp_uz") {
        want = (kind == "weights") ? std::vector<int64_t>{128, 1} : std::vector<int64_t>{1};
        known = true;
    } else if (layer == "PosePrior/fc_xyz") {
        // [512,63] (proposed/direct) or [30,63] (bottleneck variant, nets/PosePriorNetwork.py:115-116)
        if (kind == "weights" && rank == 2 && shape[0] == 30) { want = {30, 63}; name = "PosePrior/fc_xyz#bn/weights"; }
        else want = (kind == "weights") ? std::vector<int64_t>{512, 63} : std::vector<int64_t>{63};
        known = true;
    } else {
        auto fi = ctx->T.fc_idx.find(layer);
        if (fi != ctx->T.fc_idx.end()) {
            const FcL& l = ctx->T.fc[fi->second];
            want = (kind == "weights") ? std::vector<int64_t>{l.cin, l.cout} : std::vector<int64_t>{l.cout};
            known = true;
        }
    }
    if (!known) HP3D_FAIL(ctx, HP3D_ERR_ARG, "unknown variable %s", tf_var_name);
    if ((int)want.size() != rank) HP3D_FAIL(ctx, HP3D_ERR_ARG, "rank mismatch for %s", tf_var_name);
    size_t n = 1;
    for (int i = 0; i < rank; ++i) {
        if (shape[i] != want[i]) HP3D_FAIL(ctx, HP3D_ERR_ARG, "shape mismatch for %s (dim %d: got %lld want %lld)",
                                           tf_var_name, i, (long long)shape[i], (long long)want[i]);
        n *= (size_t)shape[i];
    }
    HostVar v;
    v.data.assign(data, data + n);
    v.shape.assign(shape, shape + rank);
    ctx->vars[name] = std::move(v);
    return 0;
}

int hp3d_finalize_weights(hp3d_ctx* ctx, int dtype) {
    if (ctx) ++ctx->graph_epoch;
    if (!ctx) return HP3D_ERR_ARG;
    if (dtype != 0 && dtype != 1) HP3D_FAIL(ctx, HP3D_ERR_UNSUPPORTED, "dtype must be 0 (f32) or 1 (f16 trunks)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const bool pack_timing = getenv("HP3D_PACK_TIMING") != nullptr;      // diagnostics: host seconds of the phases below on stderr
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = now();
    auto lap = [&](const char* what) {
        if (pack_timing) fprintf(stderr, "hp3d_finalize_weights: %-28s %.2f s\n", what, now() - t_mark);
        t_mark = now();
    };
    // staging: the packed blob is assembled on the host and uploaded once.  (The CPU interpreter's "device" memory IS host memory: it
    // packs straight into the blob -- copying 1.3 GB once more costs tens of seconds in the sandboxes the CPU suite runs in.)
#ifdef HP3D_EMU
    // (in place is safe against the one failure below, "incomplete variable set": variables can only be ADDED to a context, so every network
    //  that was complete before this call is complete now, and all packing is done before that check -- the blob then holds every complete
    //  network repacked in full, which is what ctx->nets still says)
    if (!ctx->blob) CHK(dev_realloc(ctx, &ctx->blob, ctx->T.blob_floats));
    memset(ctx->blob, 0, sizeof(float) * ctx->T.blob_floats);
    struct { float* p; float* data() { return p; } float& operator[](size_t i) { return p[i]; } } host{ctx->blob};
#else
    std::vector<float> host(ctx->T.blob_floats, 0.f);
#endif
    lap("blob allocation");
    int have = 0, partial = 0, bn_ok = 0;
    auto mark = [&](int net, bool present) {
        if (present) have |= net; else partial |= net;
    };
    // the convolution layers pack into disjoint regions of the blob: one task per layer, run on a few host threads (the 36-plane
    // F(4x4,3x3) copies made this 40 s on one thread)
    std::vector<std::function<void()>> pack_tasks;
    for (const ConvL& l : ctx->T.conv) {
        const HostVar *w = nullptr, *b = nullptr;
        const bool ok = find_var(ctx, l.name + "/weights", &w) == 0 && find_var(ctx, l.name + "/biases", &b) == 0;
        mark(l.net, ok);
        if (!ok) continue;
        const ConvL* lp = &l;
        float* hp = host.data();
        pack_tasks.push_back([lp, w, b, hp]() {
            const ConvL& l = *lp;
            pack_conv(l, w->data.data(), b->data.data(), hp);
            if (l.raw_off)          // lift_fused.hip: [cout block of 64][cin4][tap][64], zero rows / columns as padding
                for (int t = 0; t < 9; ++t)
                    for (int c = 0; c < l.cin; ++c)
                        for (int co = 0; co < l.cout; ++co)
                            hp[l.raw_off + (((size_t)(co >> 6) * l.cin4 + c) * 9 + t) * 64 + (co & 63)] = w->data[((size_t)t * l.cin + c) * l.cout + co];
            if (l.hwio_off) memcpy(hp + l.hwio_off, w->data.data(), sizeof(float) * (size_t)9 * l.cin * l.cout);
            if (l.ww_off) {      // U = G g G^T in the Winograd kernels' fragment orders
                std::vector<int> cmap(l.cin_pad, -1);
                for (int e = 0; e < l.cin_pad; ++e) {
                    if (l.mode == 0) cmap[e] = e < l.cin ? e : -1;
                    else if (e < 128) cmap[e] = 21 + e;            // concat buffer [encoding | scoremap | 0] vs reference [scoremap, encoding]
                    else if (e < 149) cmap[e] = e - 128;
                }
                wino_pack_weights(w->data.data(), l.k, l.cin, l.cout, l.cin_pad, l.cout_pad, cmap.data(), hp + l.ww_off);
                wino2_pack_weights(w->data.data(), l.k, l.cin, l.cout, l.cin_pad, l.cout_pad, cmap.data(), hp + l.ww2_off);
                if (l.ww4_off) wino4_pack_weights(w->data.data(), l.k, l.cin, l.cout, l.cin_pad, l.cout_pad, cmap.data(), hp + l.ww4_off);
                if (l.ww7_off) wino7_pack_weights(w->data.data(), l.cin, l.cout, l.cin_pad, l.cout_pad, cmap.data(), hp + l.ww7_off);
                if (l.ww4s_off) wino4s_pack_weights(w->data.data(), l.cin, l.cout, l.cin_pad, l.cout_pad, cmap.data(), hp + l.ww4s_off);
            }
        });
    }
    {
        std::atomic<size_t> next{0};
        auto worker = [&]() { for (size_t i; (i = next.fetch_add(1)) < pack_tasks.size();) pack_tasks[i](); };
        unsigned nthr = std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
        nthr = (unsigned)std::min<size_t>(nthr, pack_tasks.size());
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nthr; ++t) pool.emplace_back(worker);
        worker();
        for (auto& t : pool) t.join();
    }
    lap("conv layers (threads)");
    for (const FcL& l : ctx->T.fc) {
        if (l.name == "ViewpointNet/fc_vp_u") {
            const HostVar *w[3], *b[3];
            const char* ax[3] = {"ux", "uy", "uz"};
            bool ok = true;
            for (int a = 0; a < 3; ++a)
                ok = ok && find_var(ctx, std::string("ViewpointNet/fc_vp_") + ax[a] + "/weights", &w[a]) == 0 &&
                     find_var(ctx, std::string("ViewpointNet/fc_vp_") + ax[a] + "/biases", &b[a]) == 0;
            mark(l.net, ok);
            if (ok)
                for (int a = 0; a < 3; ++a) {
                    for (int i = 0; i < 128; ++i) host[l.w_off + (size_t)i * 3 + a] = w[a]->data[i];
                    host[l.b_off + a] = b[a]->data[0];
                }
            continue;
        }
        std::string bias_name = l.name + "/biases";
        if (l.name == "PosePrior/fc_xyz#bn") bias_name = "PosePrior/fc_xyz/biases";
        const HostVar *w = nullptr, *b = nullptr;
        const bool ok = find_var(ctx, l.name + "/weights", &w) == 0 && find_var(ctx, bias_name, &b) == 0;
        if (l.name == "PosePrior/fc_xyz") {
            // absent in the bottleneck variant: not an error there
            if (!ok && ctx->vars.count("PosePrior/fc_xyz#bn/weights")) continue;
        }
        if (l.net == NET_BOTTLENECK) { if (ok) ++bn_ok; }
        else mark(l.net, ok);
        if (ok) {
            memcpy(&host[l.w_off], w->data.data(), sizeof(float) * (size_t)l.cin * l.cout);
            memcpy(&host[l.b_off], b->data.data(), sizeof(float) * l.cout);
        }
    }
    const int bad = have & partial & (NET_SEG | NET_POSE | NET_PRIOR | NET_VP);
    if (bad) HP3D_FAIL(ctx, HP3D_ERR_WEIGHTS, "incomplete variable set for network mask %d", bad);
    lap("fc layers");
#ifndef HP3D_EMU
    if (!ctx->blob) CHK(dev_realloc(ctx, &ctx->blob, ctx->T.blob_floats));
    HIPCHK(ctx, hipMemcpy(ctx->blob, host.data(), sizeof(float) * ctx->T.blob_floats, hipMemcpyHostToDevice));
#endif
    lap("blob upload");
    ctx->nets = have & ~partial;
    if (bn_ok == 2) ctx->nets |= NET_BOTTLENECK;
    ctx->prec = dtype;
    if (dtype == 1) {    // half-precision copies of the HandSegNet / PoseNet2D filters (biases, heads' outputs and the lifting nets stay f32)
        std::vector<hp3d_f16> h16(ctx->T.blob16_halves, (hp3d_f16)0.f);
        for (const ConvL& l : ctx->T.conv) {
            if (!(l.net == NET_SEG || l.net == NET_POSE) || !(ctx->nets & l.net)) continue;
            const HostVar* w = nullptr;
            if (find_var(ctx, l.name + "/weights", &w) == 0) pack_conv16(l, w->data.data(), h16.data());
        }
        if (!ctx->blob16) CHK(dev_realloc(ctx, &ctx->blob16, ctx->T.blob16_halves));
        HIPCHK(ctx, hipMemcpy(ctx->blob16, h16.data(), sizeof(hp3d_f16) * ctx->T.blob16_halves, hipMemcpyHostToDevice));
    }
    // raw HWIO copies for the debug conv (conv_impl=naive)
    if (ctx->conv_naive)
        for (const ConvL& l : ctx->T.conv) {
            const HostVar* w = nullptr;
            if (find_var(ctx, l.name + "/weights", &w) != 0) continue;
            float* d = nullptr;
            HIPCHK(ctx, hipMalloc((void**)&d, w->data.size() * sizeof(float)));
            HIPCHK(ctx, hipMemcpy(d, w->data.data(), w->data.size() * sizeof(float), hipMemcpyHostToDevice));
            if (ctx->naive_w.count(l.name)) hipFree(ctx->naive_w[l.name]);
            ctx->naive_w[l.name] = d;
        }
    return 0;
}

// blob = [float32 section (all nets) | float16 section (trunk filters, always reserved so every rank agrees on the size)]
int hp3d_weights_blob_bytes(hp3d_ctx* ctx, size_t* bytes) {
    if (!ctx || !bytes) return HP3D_ERR_ARG;
    *bytes = ctx->T.blob_floats * sizeof(float) + ctx->T.blob16_halves * sizeof(hp3d_f16);
    return 0;
}
int hp3d_weights_blob_export(hp3d_ctx* ctx, void* dev_dst) {
    if (!ctx || !dev_dst) return HP3D_ERR_ARG;
    if (!ctx->blob) HP3D_FAIL(ctx, HP3D_ERR_WEIGHTS, "weights not finalized");
    const size_t n32 = ctx->T.blob_floats * sizeof(float), n16 = ctx->T.blob16_halves * sizeof(hp3d_f16);
    HIPCHK(ctx, hipMemcpyAsync(dev_dst, ctx->blob, n32, hipMemcpyDeviceToDevice, ctx->stream));
    if (ctx->blob16)
        HIPCHK(ctx, hipMemcpyAsync((char*)dev_dst + n32, ctx->blob16, n16, hipMemcpyDeviceToDevice, ctx->stream));
    else
        HIPCHK(ctx, hipMemsetAsync((char*)dev_dst + n32, 0, n16, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
int hp3d_weights_blob_import(hp3d_ctx* ctx, const void* dev_src, int nets_mask) {
    if (ctx) ++ctx->graph_epoch;
    if (!ctx || !dev_src) return HP3D_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n32 = ctx->T.blob_floats * sizeof(float), n16 = ctx->T.blob16_halves * sizeof(hp3d_f16);
    if (!ctx->blob) CHK(dev_realloc(ctx, &ctx->blob, ctx->T.blob_floats));
    HIPCHK(ctx, hipMemcpyAsync(ctx->blob, dev_src, n32, hipMemcpyDeviceToDevice, ctx->stream));
    const int f16 = (nets_mask & 32) ? 1 : 0;     // bit 5: the half-precision section is live (dtype 1)
    if (f16) {
        if (!ctx->blob16) CHK(dev_realloc(ctx, &ctx->blob16, ctx->T.blob16_halves));
        HIPCHK(ctx, hipMemcpyAsync(ctx->blob16, (const char*)dev_src + n32, n16, hipMemcpyDeviceToDevice, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->nets = nets_mask & 31;
    ctx->prec = f16;
    return 0;
}
int hp3d_nets_mask(hp3d_ctx* ctx) { return ctx ? (ctx->nets | (ctx->prec ? 32 : 0)) : 0; }

int hp3d_infer_full(hp3d_ctx* ctx, int B, int H, int W, const float* image, const float* hand_side,
                    float* hand_scoremap, float* image_crop, float* scale_crop, float* center,
                    float* keypoints_scoremap, float* keypoint_coord3d, float* hand_mask) {
    return infer_full_chunked(ctx, B, H, W, image, hand_side, hand_scoremap, image_crop, scale_crop, center,
                              keypoints_scoremap, keypoint_coord3d, hand_mask, false);
}
int hp3d_infer_full_dev(hp3d_ctx* ctx, int B, int H, int W, const float* image, const float* hand_side,
                        float* hand_scoremap, float* image_crop, float* scale_crop, float* center,
                        float* keypoints_scoremap, float* keypoint_coord3d, float* hand_mask) {
    if (!ctx) return HP3D_ERR_ARG;
    return run_graphed(ctx, graph_key("full", {B, H, W}, {image, hand_side, hand_scoremap, image_crop, scale_crop, center,
                                                        keypoints_scoremap, keypoint_coord3d, hand_mask}), [&]() {
        return infer_full_chunked(ctx, B, H, W, image, hand_side, hand_scoremap, image_crop, scale_crop, center,
                                  keypoints_scoremap, keypoint_coord3d, hand_mask, true);
    });
}

int hp3d_infer_full_kp(hp3d_ctx* ctx, int B, int H, int W, const float* image, const float* hand_side,
                       float* hand_scoremap, float* image_crop, float* scale_crop, float* center,
                       float* keypoints_scoremap, float* keypoint_coord3d, float* hand_mask,
                       int32_t* keypoint_hw_crop, double* keypoint_hw) {
    return infer_full_chunked(ctx, B, H, W, image, hand_side, hand_scoremap, image_crop, scale_crop, center,
                              keypoints_scoremap, keypoint_coord3d, hand_mask, false, nullptr, 0, 0, keypoint_hw_crop, keypoint_hw);
}
int hp3d_infer_full_kp_dev(hp3d_ctx* ctx, int B, int H, int W, const float* image, const float* hand_side,
                           float* hand_scoremap, float* image_crop, float* scale_crop, float* center,
                           float* keypoints_scoremap, float* keypoint_coord3d, float* hand_mask,
                           int32_t* keypoint_hw_crop, double* keypoint_hw) {
    if (!ctx) return HP3D_ERR_ARG;
    return run_graphed(ctx, graph_key("fullkp", {B, H, W}, {image, hand_side, hand_scoremap, image_crop, scale_crop, center,
                                                          keypoints_scoremap, keypoint_coord3d, hand_mask, keypoint_hw_crop, keypoint_hw}), [&]() {
        return infer_full_chunked(ctx, B, H, W, image, hand_side, hand_scoremap, image_crop, scale_crop, center,
                                  keypoints_scoremap, keypoint_coord3d, hand_mask, true, nullptr, 0, 0, keypoint_hw_crop, keypoint_hw);
    });
}
int hp3d_infer_full_kp_u8(hp3d_ctx* ctx, int B, int Hin, int Win, const uint8_t* image_u8, int H, int W,
                          const float* hand_side, float* hand_scoremap, float* image_crop, float* scale_crop,
                          float* center, float* keypoints_scoremap, float* keypoint_coord3d, float* hand_mask,
                          int32_t* keypoint_hw_crop, double* keypoint_hw) {
    return infer_full_chunked(ctx, B, H, W, nullptr, hand_side, hand_scoremap, image_crop, scale_crop, center,
                              keypoints_scoremap, keypoint_coord3d, hand_mask, false, image_u8, Hin, Win, keypoint_hw_crop, keypoint_hw);
}

int hp3d_infer_full_u8(hp3d_ctx* ctx, int B, int Hin, int Win, const uint8_t* image_u8, int H, int W,
                       const float* hand_side, float* hand_scoremap, float* image_crop, float* scale_crop,
                       float* center, float* keypoints_scoremap, float* keypoint_coord3d, float* hand_mask) {
    return infer_full_chunked(ctx, B, H, W, nullptr, hand_side, hand_scoremap, image_crop, scale_crop, center,
                              keypoints_scoremap, keypoint_coord3d, hand_mask, false, image_u8, Hin, Win);
}

int hp3d_preprocess_u8(hp3d_ctx* ctx, const uint8_t* image_u8, int B, int Hin, int Win, int H, int W, float* out) {
    if (!ctx) return HP3D_ERR_ARG;
    if (!image_u8 || !out || B < 1 || Hin < 2 || Win < 2 || H < 1 || W < 1) HP3D_FAIL(ctx, HP3D_ERR_ARG, "bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Scratch S(ctx);
    unsigned char* d_in = S.upload(image_u8, (size_t)B * Hin * Win * 3); NN(ctx, d_in);
    float* d_o = S.alloc<float>((size_t)B * H * W * 3); NN(ctx, d_o);
    preprocess_u8_launch(d_in, B, Hin, Win, H, W, d_o, ctx->stream);
    HIPCHK(ctx, hipMemcpyAsync(out, d_o, sizeof(float) * (size_t)B * H * W * 3, hipMemcpyDeviceToHost, ctx->stream));
    return finish_op(ctx);
}

int hp3d_infer_2d_kp(hp3d_ctx* ctx, int B, int H, int W, const float* image, float* keypoints_scoremap,
                     float* image_crop, float* scale_crop, float* center, int32_t* keypoint_hw_crop, double* keypoint_hw) {
    if (!ctx) return HP3D_ERR_ARG;
    if (!image) HP3D_FAIL(ctx, HP3D_ERR_ARG, "image is NULL");
    CHK(check_img(ctx, B, H, W));
    CHK(need_nets(ctx, NET_SEG | NET_POSE));
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // same chunking as the full path (the float32 Winograd kernels address a chunk with 32-bit offsets)
    const int mb0 = auto_micro_batch(ctx, B, H, W);
    const int mb = mb0 <= 0 ? B : std::min(mb0, B);
    CHK(ensure_arena(ctx, mb, H, W));
    const int saved_prof = ctx->profiling;
    struct ProfRestore { hp3d_ctx* c; int v; ~ProfRestore() { c->profiling = v; } } prof_restore{ctx, saved_prof};   // every exit path
    if (ctx->profiling != 2) prof_reset(ctx);   // mode 2 accumulates across calls
    for (int b0 = 0; b0 < B; b0 += mb) {
        const int nb = std::min(mb, B - b0);
        if (b0 > 0 && saved_prof == 1) ctx->profiling = 2;
        CHK(copy_in(ctx, ctx->d_image, image + (size_t)b0 * H * W * 3, (size_t)nb * H * W * 3, false));
        CHK(run_detect_and_crop(ctx, ctx->d_image, nb, H, W, 0));
        CHK(run_posenet(ctx, ctx->d_crop, nb, 256, 256, true));
        if (keypoints_scoremap) {
            resize_bilinear_launch(ctx->d_sm[2], nb, 32, 32, 21, 32, 256, 256, ctx->d_kpmap, ctx->stream);
            CHK(copy_out(ctx, keypoints_scoremap + (size_t)b0 * 256 * 256 * 21, ctx->d_kpmap, (size_t)nb * 256 * 256 * 21, false));
        }
        if (keypoint_hw_crop || keypoint_hw)
            CHK(run_kp_detect(ctx, nb, keypoint_hw_crop ? keypoint_hw_crop + (size_t)b0 * 42 : nullptr,
                              keypoint_hw ? keypoint_hw + (size_t)b0 * 42 : nullptr, false));
        if (image_crop) CHK(copy_out(ctx, image_crop + (size_t)b0 * 256 * 256 * 3, ctx->d_crop, (size_t)nb * 256 * 256 * 3, false));
        if (scale_crop) CHK(copy_out(ctx, scale_crop + b0, ctx->d_scale, (size_t)nb, false));
        if (center) CHK(copy_out(ctx, center + (size_t)b0 * 2, ctx->d_center, (size_t)nb * 2, false));
    }
    return finish_op(ctx);
}
int hp3d_infer_2d(hp3d_ctx* ctx, int B, int H, int W, const float* image, float* keypoints_scoremap,
                  float* image_crop, float* scale_crop, float* center) {
    return hp3d_infer_2d_kp(ctx, B, H, W, image, keypoints_scoremap, image_crop, scale_crop, center, nullptr, nullptr);
}

// hp3d_detect_keypoints: detect_keypoints(tf.image.resize_images(scoremap, (oh, ow))) per image and channel without the
// large map (utils/general.py:331-344 on the output of CHP3D.py:97); scoremap [B,h,w,C] with h*w <= 4096
int hp3d_detect_keypoints(hp3d_ctx* ctx, const float* scoremap, int B, int h, int w, int C, int oh, int ow, int32_t* out_rc) {
    if (!ctx) return HP3D_ERR_ARG;
    if (!scoremap || !out_rc || B < 1 || h < 1 || w < 1 || C < 1 || oh < 1 || ow < 1 || h * w > 4096)
        HP3D_FAIL(ctx, HP3D_ERR_ARG, "bad arguments (h*w must be <= 4096)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Scratch S(ctx);
    float* d_x = S.upload(scoremap, (size_t)B * h * w * C); NN(ctx, d_x);
    int* d_o = S.alloc<int>((size_t)B * C * 2); NN(ctx, d_o);
    kp_detect_launch(d_x, B, h, w, C, C, oh, ow, nullptr, nullptr, d_o, nullptr, ctx->stream);
    HIPCHK(ctx, hipMemcpyAsync(out_rc, d_o, sizeof(int) * (size_t)B * C * 2, hipMemcpyDeviceToHost, ctx->stream));
    return finish_op(ctx);
}

int hp3d_handsegnet(hp3d_ctx* ctx, int B, int H, int W, const float* image, float* scoremap_large,
                    float* scoremap_small) {
    if (!ctx) return HP3D_ERR_ARG;
    if (!image) HP3D_FAIL(ctx, HP3D_ERR_ARG, "image is NULL");
    CHK(check_img(ctx, B, H, W));
    CHK(need_nets(ctx, NET_SEG));
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CHK(ensure_arena(ctx, B, H, W));
    if (ctx->profiling != 2) prof_reset(ctx);   // mode 2 accumulates across calls
    CHK(copy_in(ctx, ctx->d_image, image, (size_t)B * H * W * 3, false));
    CHK(run_handsegnet(ctx, ctx->d_image, B, H, W));
    if (scoremap_large) {
        resize_bilinear_launch(ctx->d_segsmall, B, H / 8, W / 8, 2, 32, H, W, ctx->d_large, ctx->stream);
        CHK(copy_out(ctx, scoremap_large, ctx->d_large, (size_t)B * H * W * 2, false));
    }
    if (scoremap_small) {
        const int npix = B * (H / 8) * (W / 8);
        copy_channels_launch(ctx->d_segsmall, npix, 2, 32, ctx->bufA, 2, ctx->stream);
        CHK(copy_out(ctx, scoremap_small, ctx->bufA, (size_t)npix * 2, false));
    }
    return finish_op(ctx);
}

int hp3d_posenet2d(hp3d_ctx* ctx, int B, int H, int W, const float* image_crop, float* s0, float* s1, float* s2) {
    return posenet_impl(ctx, B, H, W, image_crop, s0, s1, s2, false);
}
int hp3d_posenet2d_dev(hp3d_ctx* ctx, int B, int H, int W, const float* image_crop, float* s0, float* s1, float* s2) {
    if (!ctx) return HP3D_ERR_ARG;
    return run_graphed(ctx, graph_key("posenet", {B, H, W}, {image_crop, s0, s1, s2}),
                       [&]() { return posenet_impl(ctx, B, H, W, image_crop, s0, s1, s2, true); });
}

static int lift_common(hp3d_ctx* ctx, int B, int variant, const float* d_sm32pad, const float* hand_side,
                       float* rel, float* can, float* rot) {
    CHK(copy_in(ctx, ctx->d_hs, hand_side, (size_t)B * 2, false));
    CHK(run_pose3d(ctx, d_sm32pad, ctx->d_hs, B, variant));
    CHK(copy_out(ctx, rel, ctx->d_coord, (size_t)B * 63, false));
    CHK(copy_out(ctx, can, ctx->d_can, (size_t)B * 63, false));
    if (variant == HP3D_VARIANT_PROPOSED) CHK(copy_out(ctx, rot, ctx->d_rot, (size_t)B * 9, false));
    return finish_op(ctx);
}

int hp3d_poseprior(hp3d_ctx* ctx, int B, int variant, const float* scoremap256, const float* hand_side,
                   float* coord_xyz_rel_normed, float* coord3d, float* rot_mat) {
    if (!ctx) return HP3D_ERR_ARG;
    if (!scoremap256 || !hand_side || B < 1) HP3D_FAIL(ctx, HP3D_ERR_ARG, "bad arguments");
    if (variant < 0 || variant > 3) HP3D_FAIL(ctx, HP3D_ERR_ARG, "Unknown variant.");
    CHK(need_nets(ctx, variant == HP3D_VARIANT_PROPOSED ? (NET_PRIOR | NET_VP)
                       : variant == HP3D_VARIANT_BOTTLENECK ? (NET_PRIOR | NET_BOTTLENECK) : NET_PRIOR));
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CHK(ensure_arena(ctx, B, 256, 256));
    if (ctx->profiling != 2) prof_reset(ctx);   // mode 2 accumulates across calls
    // stage the [B,256,256,21] GT scoremaps in bufA, pool 8x8 into the padded [B,32,32,32] buffer
    CHK(copy_in(ctx, ctx->bufB, scoremap256, (size_t)B * 256 * 256 * 21, false));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_pooled, 0, sizeof(float) * (size_t)B * 32 * 32 * 32, ctx->stream));
    avgpool8_launch(ctx->bufB, B, 256, 256, 21, ctx->d_pooled, 32, ctx->stream);
    // the lifting convs ping-pong through bufA/bufB; d_pooled is separate
    return lift_common(ctx, B, variant, ctx->d_pooled, hand_side, coord_xyz_rel_normed, coord3d, rot_mat);
}

int hp3d_pose3d(hp3d_ctx* ctx, int B, const float* scoremap32, const float* hand_side, float* coord_xyz_rel_normed,
                float* coord_can, float* rot_mat) {
    if (!ctx) return HP3D_ERR_ARG;
    if (!scoremap32 || !hand_side || B < 1) HP3D_FAIL(ctx, HP3D_ERR_ARG, "bad arguments");
    CHK(need_nets(ctx, NET_PRIOR | NET_VP));
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CHK(ensure_arena(ctx, B, 256, 256));
    if (ctx->profiling != 2) prof_reset(ctx);   // mode 2 accumulates across calls
    CHK(copy_in(ctx, ctx->bufB, scoremap32, (size_t)B * 32 * 32 * 21, false));
    pad_channels_launch(ctx->bufB, B * 32 * 32, 21, ctx->d_pooled, 32, ctx->stream);
    return lift_common(ctx, B, HP3D_VARIANT_PROPOSED, ctx->d_pooled, hand_side, coord_xyz_rel_normed, coord_can, rot_mat);
}

// ---- per-op entry points -----------------------------------------------------------------------
int hp3d_conv2d(hp3d_ctx* ctx, const float* x, int B, int H, int W, int Cin, const float* w_hwio, const float* bias,
                int k, int stride, int Cout, int act, int pool, float* out) {
    if (!ctx) return HP3D_ERR_ARG;
    if (!x || !w_hwio || !bias || !out || B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1)
        HP3D_FAIL(ctx, HP3D_ERR_ARG, "bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ConvL l;
    l.name = "op/conv2d";
    l.k = k; l.cin = Cin; l.cout = Cout; l.stride = stride; l.relu = act; l.net = 0;
    l.mode = 0; l.ek = k; l.cin_pad = pad32(Cin); l.cout_pad = pad32(Cout);
    if (ctx->use_wino == 2 && !ctx->conv_naive && (k == 3 || k == 7) && stride == 1) {       // conv_impl=winograd: no silent fallback
        l.cin_pad = (Cin + 63) / 64 * 64;
        if (Cout % 64) HP3D_FAIL(ctx, HP3D_ERR_UNSUPPORTED, "conv_impl=winograd needs Cout %% 64 == 0 (got %d)", Cout);
    }
    l.w_off = 0; l.b_off = (size_t)k * k * l.cin_pad * l.cout_pad;
    std::vector<float> packed(l.b_off + l.cout_pad);
    pack_conv(l, w_hwio, bias, packed.data());
    int Ho, Wo, pt, pl;
    same_pad(H, k, stride, &Ho, &pt);
    same_pad(W, k, stride, &Wo, &pl);
    const int Hs = pool ? Ho / 2 : Ho, Ws = pool ? Wo / 2 : Wo;
    Scratch S(ctx);
    float* d_x = S.upload(x, (size_t)B * H * W * Cin); NN(ctx, d_x);
    float* d_xp = d_x;
    if (l.cin_pad != Cin) {
        d_xp = S.alloc<float>((size_t)B * H * W * l.cin_pad); NN(ctx, d_xp);
        pad_channels_launch(d_x, B * H * W, Cin, d_xp, l.cin_pad, ctx->stream);
    }
    float* d_out = S.alloc<float>((size_t)B * Hs * Ws * Cout); NN(ctx, d_out);
    if (k == 3 && Cin == 3 && Cout == 64 && !pool && ctx->use_first && !ctx->conv_naive && ctx->use_wino != 2 &&
        conv_first_eligible(k, stride, Cin, Cout, B, H, W, Cout, 0)) {
        // the conv1_1 shape (nets/ColorHandPose3DNetwork.py:144,183) runs on the networks' own first-layer kernel (conv_first.hip)
        ConvL l1 = l;
        l1.mode = 1; l1.ek = 1; l1.cin_pad = 32; l1.cout_pad = 64;
        l1.w_off = 0; l1.b_off = (size_t)32 * 64;
        std::vector<float> pk1(l1.b_off + 64);
        pack_conv(l1, w_hwio, bias, pk1.data());
        float* d_pk = S.upload(pk1.data(), pk1.size()); NN(ctx, d_pk);
        ConvParams p;
        p.in = d_x; p.wpk = d_pk; p.bias = d_pk + l1.b_off; p.out = d_out;
        p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
        p.Cin = 3; p.in_cs = 3; p.Cout = 64; p.out_cs = 64; p.cout_store = 64;
        p.pad_t = pt; p.pad_l = pl; p.tiles_x = 0; p.tiles_y = 0;
        p.act = act; p.im2col = 1; p.ksplit = 1; p.partial = nullptr; p.f16 = 0; p.out_f32 = 0; p.nsub = 1;
        conv_first_launch(p, ctx->stream, ctx->first_balanced);
        ++ctx->conv_first_launches;
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(out, d_out, sizeof(float) * (size_t)B * Hs * Ws * Cout, hipMemcpyDeviceToHost, ctx->stream));
        return finish_op(ctx);
    }
    int op_ks7 = 1;
    if (ctx->use_wino && ctx->use_wino7 == 1 && !ctx->conv_naive && k == 7 && !pool && Cout % 64 == 0 &&
        conv_wino7_eligible(k, stride, l.cin_pad, l.cout_pad, Ho, Wo, B, l.cin_pad, Cout, nullptr, ctx->wino_splitk ? &op_ks7 : nullptr)) {
        op_ks7 = wino7_ks_override(ctx, op_ks7, l.cin_pad, (long)B * Ho * Wo * Cout);
        // option "wino7" = "1": the F(4x4,4x4) form of a 7x7 filter (conv_wino7.hip)
        const size_t wn = wino7_packed_floats(l.cin_pad, l.cout_pad);
        std::vector<float> pw(wn + l.cout_pad, 0.f);
        wino7_pack_weights(w_hwio, Cin, Cout, l.cin_pad, l.cout_pad, nullptr, pw.data());
        for (int co = 0; co < Cout; ++co) pw[wn + co] = bias[co];
        float* d_pk = S.upload(pw.data(), pw.size()); NN(ctx, d_pk);
        ConvParams p;
        p.in = d_xp; p.wpk = d_pk; p.bias = d_pk + wn; p.out = d_out;
        p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
        p.Cin = l.cin_pad; p.in_cs = l.cin_pad; p.Cout = l.cout_pad; p.out_cs = Cout; p.cout_store = Cout;
        p.pad_t = pt; p.pad_l = pl; p.tiles_x = 0; p.tiles_y = 0;
        p.act = act; p.im2col = 0; p.ksplit = op_ks7; p.partial = nullptr; p.f16 = 0; p.out_f32 = 0; p.nsub = 4;
        float* d_part7 = nullptr;
        if (op_ks7 > 1) {          // under-filled launch: raw partial sums per channel split, then the deterministic reduce
            d_part7 = S.alloc<float>((size_t)op_ks7 * B * Ho * Wo * Cout); NN(ctx, d_part7);
            p.out = d_part7;
        }
        if (conv_wino7_launch(p, ctx->stream)) HP3D_FAIL(ctx, HP3D_ERR_ARG, "winograd conv (F(4x4,4x4)): launch refused");
        if (op_ks7 > 1) conv_splitk_reduce_launch(d_part7, op_ks7, (long)B * Ho * Wo, Cout, d_pk + wn, act, d_out, Cout, Cout, ctx->stream);
        ++ctx->conv_wino7_launches;
        ctx->conv_wino7_split_launches += op_ks7 > 1;
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(out, d_out, sizeof(float) * (size_t)B * Hs * Ws * Cout, hipMemcpyDeviceToHost, ctx->stream));
        return finish_op(ctx);
    }
    if (ctx->use_wino && ctx->use_wino4s == 1 && !ctx->conv_naive && Cout % 64 == 0 &&
        conv_wino4s_eligible(k, stride, l.cin_pad, l.cout_pad, Ho, Wo, B, l.cin_pad, Cout, pool, nullptr)) {
        // option "wino4_split" = "1": the F(4x4,3x3) kernel with split bfloat16 operands (conv_wino4s.hip)
        const size_t wn = (wino4s_packed_bytes(l.cin_pad, l.cout_pad) + 15) / 16 * 4;
        std::vector<float> pw(wn + l.cout_pad, 0.f);
        wino4s_pack_weights(w_hwio, Cin, Cout, l.cin_pad, l.cout_pad, nullptr, pw.data());
        for (int co = 0; co < Cout; ++co) pw[wn + co] = bias[co];
        float* d_pk = S.upload(pw.data(), pw.size()); NN(ctx, d_pk);
        ConvParams p;
        p.in = d_xp; p.wpk = d_pk; p.bias = d_pk + wn; p.out = d_out;
        p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
        p.Cin = l.cin_pad; p.in_cs = l.cin_pad; p.Cout = l.cout_pad; p.out_cs = Cout; p.cout_store = Cout;
        p.pad_t = pt; p.pad_l = pl; p.tiles_x = 0; p.tiles_y = 0;
        p.act = act; p.im2col = 0; p.ksplit = 1; p.partial = nullptr; p.f16 = 0; p.out_f32 = 0; p.nsub = 1;
        if (ctx->w4_tail && conv_wino4_tail_plan(l.cin_pad, l.cout_pad, Ho, Wo, B, nullptr) > 0) {
            p.partial = S.alloc<float>(conv_wino4s_tail_floats()); NN(ctx, p.partial);
            p.partial_cap = conv_wino4s_tail_floats();
        }
        const int lr = conv_wino4s_launch(p, pool, ctx->stream);
        if (lr < 0) HP3D_FAIL(ctx, HP3D_ERR_ARG, "winograd conv (F(4x4,3x3), split operands): launch refused");
        if (lr == 1) ++ctx->conv_wino4s_tail_launches;
        ++ctx->conv_wino4s_launches;
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(out, d_out, sizeof(float) * (size_t)B * Hs * Ws * Cout, hipMemcpyDeviceToHost, ctx->stream));
        return finish_op(ctx);
    }
    int op_ks = 1, op_ks2 = 1;
    const bool op4 = ctx->use_wino && ctx->use_wino4 == 1 && !ctx->conv_naive && Cout % 64 == 0 &&
        conv_wino4_eligible(k, stride, l.cin_pad, l.cout_pad, Ho, Wo, B, l.cin_pad, Cout, pool, ctx->wino_splitk ? &op_ks2 : nullptr);
    if (op4 || (ctx->use_wino && ctx->use_wino2 == 1 && !ctx->conv_naive && Cout % 64 == 0 &&
        conv_wino2_eligible(k, stride, l.cin_pad, l.cout_pad, Ho, Wo, B, l.cin_pad, Cout, pool, ctx->wino_splitk ? &op_ks2 : nullptr))) {
        // option "wino4" = "1": the F(4x4,3x3) kernel (conv_wino4.hip); option "wino2" = "1": the two-workgroups-per-CU F(2x2,3x3) kernel
        const size_t wn = op4 ? wino4_packed_floats(k, l.cin_pad, l.cout_pad) : wino_packed_floats(k, l.cin_pad, l.cout_pad);
        std::vector<float> pw(wn + l.cout_pad, 0.f);
        if (op4) wino4_pack_weights(w_hwio, k, Cin, Cout, l.cin_pad, l.cout_pad, nullptr, pw.data());
        else wino2_pack_weights(w_hwio, k, Cin, Cout, l.cin_pad, l.cout_pad, nullptr, pw.data());
        for (int co = 0; co < Cout; ++co) pw[wn + co] = bias[co];
        float* d_pk = S.upload(pw.data(), pw.size()); NN(ctx, d_pk);
        ConvParams p;
        p.in = d_xp; p.wpk = d_pk; p.bias = d_pk + wn; p.out = d_out;
        p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
        p.Cin = l.cin_pad; p.in_cs = l.cin_pad; p.Cout = l.cout_pad; p.out_cs = Cout; p.cout_store = Cout;
        p.pad_t = pt; p.pad_l = pl; p.tiles_x = 0; p.tiles_y = 0;
        p.act = act; p.im2col = 0; p.ksplit = op_ks2; p.partial = nullptr; p.f16 = 0; p.out_f32 = 0;
        p.nsub = k == 7 ? 9 : 1;
        float* d_part = nullptr;
        if (op_ks2 > 1) {
            d_part = S.alloc<float>((size_t)op_ks2 * B * Ho * Wo * Cout); NN(ctx, d_part);
            p.out = d_part;
        } else if (op4 && k == 3 && ctx->w4_tail &&
                   conv_wino4_tail_plan(l.cin_pad, l.cout_pad, Ho, Wo, B, nullptr) > 0) {
            p.partial = S.alloc<float>(conv_wino4_tail_floats()); NN(ctx, p.partial);         // tail pieces (conv_wino4.hip, TAIL)
            p.partial_cap = conv_wino4_tail_floats();
        }
        const int lr = op4 ? conv_wino4_launch(p, op_ks2 > 1 ? 0 : pool, ctx->stream) : conv_wino2_launch(p, op_ks2 > 1 ? 0 : pool, ctx->stream);
        if (lr < 0) HP3D_FAIL(ctx, HP3D_ERR_ARG, "winograd conv (%s): launch refused", op4 ? "F(4x4,3x3)" : "2 workgroups per CU");
        if (op4 && lr == 1) ++ctx->conv_wino4_tail_launches;
        ++(op4 ? ctx->conv_wino4_launches : ctx->conv_wino2_launches);
        if (op_ks2 > 1 && pool)
            conv_splitk_reduce_pool_launch(d_part, op_ks2, B, Ho, Wo, Cout, d_pk + wn, act, d_out, Cout, Cout, ctx->stream);
        else if (op_ks2 > 1)
            conv_splitk_reduce_launch(d_part, op_ks2, (long)B * Ho * Wo, Cout, d_pk + wn, act, d_out, Cout, Cout, ctx->stream);
    } else if (ctx->use_wino && !ctx->conv_naive && conv_wino_eligible(ctx->use_wino, k, stride, l.cin_pad, l.cout_pad, Ho, Wo, B, l.cin_pad, Cout, pool, ctx->wino_splitk ? &op_ks : nullptr) && Cout % 64 == 0) {
        const size_t wn = wino_packed_floats(k, l.cin_pad, l.cout_pad);
        std::vector<float> pw(wn + l.cout_pad, 0.f);
        wino_pack_weights(w_hwio, k, Cin, Cout, l.cin_pad, l.cout_pad, nullptr, pw.data());
        for (int co = 0; co < Cout; ++co) pw[wn + co] = bias[co];
        float* d_pk = S.upload(pw.data(), pw.size()); NN(ctx, d_pk);
        ConvParams p;
        p.in = d_xp; p.wpk = d_pk; p.bias = d_pk + wn; p.out = d_out;
        p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
        p.Cin = l.cin_pad; p.in_cs = l.cin_pad; p.Cout = l.cout_pad; p.out_cs = Cout; p.cout_store = Cout;
        p.pad_t = pt; p.pad_l = pl;
        p.tiles_x = (Wo + 15) / 16; p.tiles_y = (Ho + 7) / 8;
        p.act = act; p.im2col = 0; p.ksplit = op_ks; p.partial = nullptr; p.f16 = 0; p.out_f32 = 0;
        p.nsub = k == 7 ? 9 : 1;
        float* d_part = nullptr;
        if (op_ks > 1) {
            d_part = S.alloc<float>((size_t)op_ks * B * Ho * Wo * Cout); NN(ctx, d_part);
            p.out = d_part;
        }
        if (conv_wino_launch(p, op_ks > 1 ? 0 : pool, ctx->stream)) HP3D_FAIL(ctx, HP3D_ERR_ARG, "winograd conv: tensor exceeds 32-bit offsets");
        if (op_ks > 1 && pool)
            conv_splitk_reduce_pool_launch(d_part, op_ks, B, Ho, Wo, Cout, d_pk + wn, act, d_out, Cout, Cout, ctx->stream);
        else if (op_ks > 1)
            conv_splitk_reduce_launch(d_part, op_ks, (long)B * Ho * Wo, Cout, d_pk + wn, act, d_out, Cout, Cout, ctx->stream);
    } else if (ctx->conv_naive) {
        if (pool) HP3D_FAIL(ctx, HP3D_ERR_UNSUPPORTED, "naive conv has no fused pool");
        float* d_w = S.upload(w_hwio, (size_t)k * k * Cin * Cout); NN(ctx, d_w);
        float* d_b = S.upload(bias, (size_t)Cout); NN(ctx, d_b);
        conv_naive_launch(d_x, B, H, W, Cin, Cin, d_w, d_b, k, stride, Cout, act, d_out, Cout, Ho, Wo, pt, pl, ctx->stream);
    } else {
        float* d_pk = S.upload(packed.data(), packed.size()); NN(ctx, d_pk);
        ConvPlan plan;
        if (conv_mfma_plan(k, stride, Ho, Wo, l.cin_pad, l.cout_pad, pool, B, &plan) != 0)
            HP3D_FAIL(ctx, HP3D_ERR_UNSUPPORTED, "no conv_mfma variant for k=%d stride=%d pool=%d", k, stride, pool);
        float* d_part = nullptr;
        if (plan.ksplit > 1) { d_part = S.alloc<float>((size_t)plan.ksplit * B * Ho * Wo * l.cout_pad); NN(ctx, d_part); }
        ConvParams p;
        p.in = d_xp; p.wpk = d_pk; p.bias = d_pk + l.b_off; p.out = d_out;
        p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
        p.Cin = l.cin_pad; p.in_cs = l.cin_pad; p.Cout = l.cout_pad; p.out_cs = Cout; p.cout_store = Cout;
        p.pad_t = pt; p.pad_l = pl;
        p.tiles_x = (Wo + plan.tw - 1) / plan.tw; p.tiles_y = (Ho + plan.th - 1) / plan.th;
        p.act = act; p.im2col = 0; p.f16 = 0; p.out_f32 = 0;
        p.ksplit = plan.ksplit; p.partial = d_part;
        if (conv_mfma_launch(p, k, stride, pool, plan, ctx->stream) != 0)
            HP3D_FAIL(ctx, HP3D_ERR_UNSUPPORTED, "conv_mfma launch failed");
        if (plan.ksplit > 1)
            conv_splitk_reduce_launch(d_part, plan.ksplit, (long)B * Ho * Wo, l.cout_pad, p.bias, act, d_out, Cout, Cout,
                                      ctx->stream);
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, d_out, sizeof(float) * (size_t)B * Hs * Ws * Cout, hipMemcpyDeviceToHost, ctx->stream));
    return finish_op(ctx);
}

int hp3d_maxpool2(hp3d_ctx* ctx, const float* x, int B, int H, int W, int C, float* out) {
    if (!ctx) return HP3D_ERR_ARG;
    if (!x || !out || B < 1 || H < 2 || W < 2 || C < 1) HP3D_FAIL(ctx, HP3D_ERR_ARG, "bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Scratch S(ctx);
    const size_t no = (size_t)B * (H / 2) * (W / 2) * C;
    float* d_x = S.upload(x, (size_t)B * H * W * C); NN(ctx, d_x);
    float* d_o = S.alloc<float>(no); NN(ctx, d_o);
    maxpool2_launch(d_x, B, H, W, C, C, d_o, ctx->stream);
    HIPCHK(ctx, hipMemcpyAsync(out, d_o, no * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    return finish_op(ctx);
}

int hp3d_avgpool8(hp3d_ctx* ctx, const float* x, int B, int H, int W, int C, float* out) {
    if (!ctx) return HP3D_ERR_ARG;
    if (!x || !out || B < 1 || (H % 8) || (W % 8) || C < 1) HP3D_FAIL(ctx, HP3D_ERR_ARG, "bad arguments (H,W %% 8)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Scratch S(ctx);
    const size_t no = (size_t)B * (H / 8) * (W / 8) * C;
    float* d_x = S.upload(x, (size_t)B * H * W * C); NN(ctx, d_x);
    float* d_o = S.alloc<float>(no); NN(ctx, d_o);
    avgpool8_launch(d_x, B, H, W, C, d_o, C, ctx->stream);
    HIPCHK(ctx, hipMemcpyAsync(out, d_o, no * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    return finish_op(ctx);
}

int hp3d_resize_bilinear(hp3d_ctx* ctx, const float* x, int B, int H, int W, int C, int out_h, int out_w, float* out) {
    if (!ctx) return HP3D_ERR_ARG;
    if (!x || !out || B < 1 || H < 1 || W < 1 || C < 1 || out_h < 1 || out_w < 1) HP3D_FAIL(ctx, HP3D_ERR_ARG, "bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Scratch S(ctx);
    const size_t no = (size_t)B * out_h * out_w * C;
    float* d_x = S.upload(x, (size_t)B * H * W * C); NN(ctx, d_x);
    float* d_o = S.alloc<float>(no); NN(ctx, d_o);
    resize_bilinear_launch(d_x, B, H, W, C, C, out_h, out_w, d_o, ctx->stream);
    HIPCHK(ctx, hipMemcpyAsync(out, d_o, no * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    return finish_op(ctx);
}

int hp3d_crop_and_resize(hp3d_ctx* ctx, const float* image, int B, int H, int W, int C, const float* center,
                         const float* scale, int crop_size, float* out) {
    if (!ctx) return HP3D_ERR_ARG;
    if (!image || !center || !scale || !out || B < 1 || H < 2 || W < 2 || C < 1 || crop_size < 1)
        HP3D_FAIL(ctx, HP3D_ERR_ARG, "bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Scratch S(ctx);
    const size_t no = (size_t)B * crop_size * crop_size * C;
    float* d_x = S.upload(image, (size_t)B * H * W * C); NN(ctx, d_x);
    float* d_c = S.upload(center, (size_t)B * 2); NN(ctx, d_c);
    float* d_s = S.upload(scale, (size_t)B); NN(ctx, d_s);
    float* d_o = S.alloc<float>(no); NN(ctx, d_o);
    crop_and_resize_launch(d_x, B, H, W, C, d_c, d_s, crop_size, d_o, ctx->stream);
    HIPCHK(ctx, hipMemcpyAsync(out, d_o, no * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    return finish_op(ctx);
}

int hp3d_mask_from_scoremap(hp3d_ctx* ctx, const float* scoremap, int B, int H, int W, float* mask, float* center,
                            float* crop_size, float* scale, int32_t* seed) {
    if (!ctx) return HP3D_ERR_ARG;
    if (!scoremap || B < 1 || H < 1 || W < 1) HP3D_FAIL(ctx, HP3D_ERR_ARG, "bad arguments");
    if (!(B < H && B < W)) HP3D_FAIL(ctx, HP3D_ERR_ARG, "Scoremap must be [Batch, Width, Height]");  // general.py:210
    if (mask_grow_lds_bytes(H, W) > 160 * 1024 - 1024) HP3D_FAIL(ctx, HP3D_ERR_ARG, "map too large");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Scratch S(ctx);
    const size_t npx = (size_t)B * H * W;
    float* d_sm = S.upload(scoremap, npx * 2); NN(ctx, d_sm);
    MaskBuffers mb;
    mb.argmax_key = S.alloc<unsigned long long>(B); NN(ctx, mb.argmax_key);
    mb.det = S.alloc<unsigned char>(npx); NN(ctx, mb.det);
    mb.fg = nullptr;
    float* d_mask = S.alloc<float>(npx); NN(ctx, d_mask);
    float* d_c = S.alloc<float>((size_t)B * 2); NN(ctx, d_c);
    float* d_cs = S.alloc<float>(B); NN(ctx, d_cs);
    float* d_sc = S.alloc<float>(B); NN(ctx, d_sc);
    int* d_seed = S.alloc<int>((size_t)B * 2); NN(ctx, d_seed);
    seg_softmax_launch(d_sm, B, H, W, mb, ctx->stream);
    mask_grow_launch(mb, B, H, W, ctx->empty_fltmax, d_mask, d_c, d_cs, d_sc, d_seed, ctx->stream);
    if (mask) HIPCHK(ctx, hipMemcpyAsync(mask, d_mask, npx * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (center) HIPCHK(ctx, hipMemcpyAsync(center, d_c, sizeof(float) * B * 2, hipMemcpyDeviceToHost, ctx->stream));
    if (crop_size) HIPCHK(ctx, hipMemcpyAsync(crop_size, d_cs, sizeof(float) * B, hipMemcpyDeviceToHost, ctx->stream));
    if (scale) HIPCHK(ctx, hipMemcpyAsync(scale, d_sc, sizeof(float) * B, hipMemcpyDeviceToHost, ctx->stream));
    if (seed) HIPCHK(ctx, hipMemcpyAsync(seed, d_seed, sizeof(int) * B * 2, hipMemcpyDeviceToHost, ctx->stream));
    return finish_op(ctx);
}

int hp3d_fc(hp3d_ctx* ctx, const float* x, int B, int Cin, const float* w, const float* bias, int Cout, int act,
            float* out) {
    if (!ctx) return HP3D_ERR_ARG;
    if (!x || !w || !bias || !out || B < 1 || Cin < 1 || Cout < 1) HP3D_FAIL(ctx, HP3D_ERR_ARG, "bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Scratch S(ctx);
    float* d_x = S.upload(x, (size_t)B * Cin); NN(ctx, d_x);
    float* d_w = S.upload(w, (size_t)Cin * Cout); NN(ctx, d_w);
    float* d_b = S.upload(bias, (size_t)Cout); NN(ctx, d_b);
    float* d_o = S.alloc<float>((size_t)B * Cout); NN(ctx, d_o);
    float* d_p = S.alloc<float>(fc_scratch_floats(B, Cin, Cout)); NN(ctx, d_p);
    fc_launch(d_x, B, Cin, Cin, d_w, d_b, Cout, act, d_o, Cout, d_p, ctx->stream);
    HIPCHK(ctx, hipMemcpyAsync(out, d_o, sizeof(float) * (size_t)B * Cout, hipMemcpyDeviceToHost, ctx->stream));
    return finish_op(ctx);
}

int hp3d_argmax2d(hp3d_ctx* ctx, const float* x, int B, int H, int W, int C, int32_t* out_rc) {
    if (!ctx) return HP3D_ERR_ARG;
    if (!x || !out_rc || B < 1 || H < 1 || W < 1 || C < 1) HP3D_FAIL(ctx, HP3D_ERR_ARG, "bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Scratch S(ctx);
    float* d_x = S.upload(x, (size_t)B * H * W * C); NN(ctx, d_x);
    int* d_o = S.alloc<int>((size_t)B * C * 2); NN(ctx, d_o);
    argmax2d_launch(d_x, B, H, W, C, C, d_o, ctx->stream);
    HIPCHK(ctx, hipMemcpyAsync(out_rc, d_o, sizeof(int) * (size_t)B * C * 2, hipMemcpyDeviceToHost, ctx->stream));
    return finish_op(ctx);
}

// ---- measurement ---------------------------------------------------------------------------------
int hp3d_set_profiling(hp3d_ctx* ctx, int on) {
    if (!ctx) return HP3D_ERR_ARG;
    if (on != ctx->profiling) prof_reset(ctx);
    ctx->profiling = (on == 2) ? 2 : (on ? 1 : 0);
    return 0;
}
int hp3d_prof_count(hp3d_ctx* ctx) { return ctx ? (int)ctx->prof.size() : 0; }
int hp3d_prof_get(hp3d_ctx* ctx, int i, char* name, int name_cap, char* kernel, int kernel_cap, float* ms,
                  double* flops, double* bytes) {
    if (!ctx || i < 0 || i >= (int)ctx->prof.size()) return HP3D_ERR_ARG;
    const ProfRec& r = ctx->prof[i];
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", r.name.c_str());
    if (kernel && kernel_cap > 0) snprintf(kernel, kernel_cap, "%s", r.kernel.c_str());
    if (ms) {
        HIPCHK(ctx, hipEventSynchronize(r.e1));
        HIPCHK(ctx, hipEventElapsedTime(ms, r.e0, r.e1));
    }
    if (flops) *flops = r.flops;
    if (bytes) *bytes = r.bytes;
    return 0;
}

int hp3d_get_counter(hp3d_ctx* ctx, const char* name, long long* value) {
    if (!ctx || !name || !value) return HP3D_ERR_ARG;
    const std::string k(name);
    if (k == "graph_captures") { *value = ctx->graph_captures; return 0; }
    if (k == "graph_replays") { *value = ctx->graph_replays; return 0; }
    if (k == "conv_h16_launches") { *value = ctx->conv_h16_launches; return 0; }
    if (k == "conv_h16_first_resident_launches") { *value = ctx->conv_h16_first_resident_launches; return 0; }
    if (k == "first_touch_launches") { *value = ctx->first_touch_launches + (ctx->kid ? ctx->kid->first_touch_launches : 0); return 0; }
    if (k == "conv_first_launches") { *value = ctx->conv_first_launches + (ctx->kid ? ctx->kid->conv_first_launches : 0); return 0; }
    if (k == "lift_overlap_calls") { *value = ctx->lift_overlap_calls; return 0; }
    if (k == "lift_fused_launches") { *value = ctx->lift_fused_launches + (ctx->kid ? ctx->kid->lift_fused_launches : 0); return 0; }
    if (k == "conv_wino4_tail_launches") { *value = ctx->conv_wino4_tail_launches + (ctx->kid ? ctx->kid->conv_wino4_tail_launches : 0); return 0; }
    if (k == "conv_pw2_launches") { *value = ctx->conv_pw2_launches + (ctx->kid ? ctx->kid->conv_pw2_launches : 0); return 0; }
    if (k == "conv_wino7_split_launches") { *value = ctx->conv_wino7_split_launches + (ctx->kid ? ctx->kid->conv_wino7_split_launches : 0); return 0; }
    if (k == "conv_wino7_launches") { *value = ctx->conv_wino7_launches + (ctx->kid ? ctx->kid->conv_wino7_launches : 0); return 0; }
#ifdef HP3D_EMU
    if (k == "emu_soff_overreads") { *value = (long)hp3d_emu_soff_overreads; return 0; }     // interpreter only: 16-byte loads that left their buffer through the scalar offset
#endif
    if (k == "fc_tail_launches") { *value = ctx->fc_tail_launches + (ctx->kid ? ctx->kid->fc_tail_launches : 0); return 0; }
    if (k == "conv_s2_gemm_launches") { *value = ctx->conv_s2_gemm_launches + (ctx->kid ? ctx->kid->conv_s2_gemm_launches : 0); return 0; }
    if (k == "conv_wino4s_launches") { *value = ctx->conv_wino4s_launches + (ctx->kid ? ctx->kid->conv_wino4s_launches : 0); return 0; }
    if (k == "conv_wino4s_tail_launches") { *value = ctx->conv_wino4s_tail_launches + (ctx->kid ? ctx->kid->conv_wino4s_tail_launches : 0); return 0; }
    if (k == "conv_wino4_launches") { *value = ctx->conv_wino4_launches + (ctx->kid ? ctx->kid->conv_wino4_launches : 0); return 0; }
    if (k == "conv_wino2_launches") { *value = ctx->conv_wino2_launches + (ctx->kid ? ctx->kid->conv_wino2_launches : 0); return 0; }
    if (k == "comm_ranks") { *value = comm_ranks(ctx); return 0; }
    HP3D_FAIL(ctx, HP3D_ERR_ARG, "unknown counter %s", name);
}
int hp3d_get_timing(hp3d_ctx* ctx, float* ms_per_stage, int n) {
    if (!ctx || !ms_per_stage || n < 1) return HP3D_ERR_ARG;
    float acc[HP3D_TIMING_STAGES] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (const ProfRec& r : ctx->prof) {
        float ms = 0.f;
        HIPCHK(ctx, hipEventSynchronize(r.e1));
        HIPCHK(ctx, hipEventElapsedTime(&ms, r.e0, r.e1));
        int st = 1;                                                        // glue: softmax / mask / crop
        if (r.name.rfind("HandSegNet/", 0) == 0) st = 0;
        else if (r.name.rfind("PoseNet2D/", 0) == 0 || r.name == "kp_upsample") st = 2;
        else if (r.name.rfind("PosePrior", 0) == 0 || r.name.rfind("ViewpointNet/", 0) == 0 || r.name == "lift_epilogue" ||
                 r.name.rfind("fc", 0) == 0) st = 3;
        acc[st] += ms;
        acc[4] += ms;
    }
    for (int i = 0; i < n && i < HP3D_TIMING_STAGES; ++i) ms_per_stage[i] = acc[i];
    return 0;
}

// ---- RCCL (loaded lazily) ----------------------------------------------------------------------------------------
#ifndef HP3D_EMU
namespace {
struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;       // optional: the communicator's own idea of its size
};
Rccl* rccl(hp3d_ctx* ctx) {
    static Rccl R;
    if (R.h) return &R;
    // a process that already carries an RCCL (e.g. torch's) gets that one: same SONAME
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* nm : names)
        if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) { set_error(ctx, "librccl not found (dlopen)"); return nullptr; }
    R.GetUniqueId = (decltype(R.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    R.CommInitRank = (decltype(R.CommInitRank))dlsym(h, "ncclCommInitRank");
    R.Broadcast = (decltype(R.Broadcast))dlsym(h, "ncclBroadcast");
    R.AllGather = (decltype(R.AllGather))dlsym(h, "ncclAllGather");
    R.CommDestroy = (decltype(R.CommDestroy))dlsym(h, "ncclCommDestroy");
    R.GetErrorString = (decltype(R.GetErrorString))dlsym(h, "ncclGetErrorString");
    R.CommCount = (decltype(R.CommCount))dlsym(h, "ncclCommCount");
    if (!R.GetUniqueId || !R.CommInitRank || !R.Broadcast || !R.AllGather || !R.CommDestroy || !R.GetErrorString) {
        set_error(ctx, "librccl lacks an expected symbol");
        return nullptr;
    }
    R.h = h;
    return &R;
}
#define NCCLCHK(ctx, R, call)                                                          \
    do {                                                                               \
        ncclResult_t _r = (call);                                                      \
        if (_r != ncclSuccess) HP3D_FAIL(ctx, HP3D_ERR_HIP, "%s: %s", #call, (R)->GetErrorString(_r)); \
    } while (0)
}  // namespace

int hp3d_comm_unique_id(void* id128) {
    if (!id128) return HP3D_ERR_ARG;
    static_assert(sizeof(ncclUniqueId) == HP3D_COMM_ID_BYTES, "ncclUniqueId size");
    Rccl* R = rccl(nullptr);
    if (!R) return HP3D_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != ncclSuccess) { set_error(nullptr, "ncclGetUniqueId failed"); return HP3D_ERR_HIP; }
    memcpy(id128, &id, sizeof(id));
    return 0;
}
int hp3d_comm_init(hp3d_ctx* ctx, int rank, int nranks, const void* id128) {
    if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return HP3D_ERR_ARG;
    if (ctx->comm) HP3D_FAIL(ctx, HP3D_ERR_ARG, "communicator already initialised");
    Rccl* R = rccl(ctx);
    if (!R) return HP3D_ERR_UNSUPPORTED;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    NCCLCHK(ctx, R, R->CommInitRank(&c, nranks, id, rank));
    ctx->comm = c; ctx->comm_rank = rank; ctx->comm_size = nranks;
    return 0;
}
int hp3d_bcast_weights(hp3d_ctx* ctx, int root) {
    if (ctx) ++ctx->graph_epoch;
    if (!ctx || !ctx->comm || root < 0 || root >= ctx->comm_size) return HP3D_ERR_ARG;
    Rccl* R = rccl(ctx);
    if (!R) return HP3D_ERR_UNSUPPORTED;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ncclComm_t c = (ncclComm_t)ctx->comm;
    const bool is_root = ctx->comm_rank == root;
    if (is_root && !ctx->blob) HP3D_FAIL(ctx, HP3D_ERR_WEIGHTS, "root has no finalized weights");
    // header: nets mask (bit 5 = half-precision section live); the blob layout is identical on every rank
    Scratch S(ctx);
    int* d_hdr = S.alloc<int>(2); NN(ctx, d_hdr);
    int hdr[2] = {is_root ? hp3d_nets_mask(ctx) : 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(d_hdr, hdr, sizeof(hdr), hipMemcpyHostToDevice, ctx->stream));
    NCCLCHK(ctx, R, R->Broadcast(d_hdr, d_hdr, 2, ncclInt, root, c, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(hdr, d_hdr, sizeof(hdr), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const int mask = hdr[0], f16 = (mask & 32) ? 1 : 0;
    if (!ctx->blob) CHK(dev_realloc(ctx, &ctx->blob, ctx->T.blob_floats));
    NCCLCHK(ctx, R, R->Broadcast(ctx->blob, ctx->blob, ctx->T.blob_floats, ncclFloat, root, c, ctx->stream));
    if (f16) {
        if (!ctx->blob16) CHK(dev_realloc(ctx, &ctx->blob16, ctx->T.blob16_halves));
        NCCLCHK(ctx, R, R->Broadcast(ctx->blob16, ctx->blob16, ctx->T.blob16_halves, ncclHalf, root, c, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->nets = mask & 31;
    ctx->prec = f16;
    return 0;
}
static int gather_staging(hp3d_ctx* ctx, size_t floats) {
    if (floats > ctx->gather_floats) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        CHK(dev_realloc(ctx, &ctx->d_gather, floats));
        ctx->gather_floats = floats;
    }
    return 0;
}
int hp3d_allgather(hp3d_ctx* ctx, const float* send_host, int count, float* recv_host) {
    if (!ctx || !ctx->comm || !send_host || !recv_host || count < 1) return HP3D_ERR_ARG;
    Rccl* R = rccl(ctx);
    if (!R) return HP3D_ERR_UNSUPPORTED;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CHK(gather_staging(ctx, (size_t)count * (ctx->comm_size + 1)));
    float* d_send = ctx->d_gather;
    float* d_recv = ctx->d_gather + count;
    HIPCHK(ctx, hipMemcpyAsync(d_send, send_host, sizeof(float) * count, hipMemcpyHostToDevice, ctx->stream));
    NCCLCHK(ctx, R, R->AllGather(d_send, d_recv, (size_t)count, ncclFloat, (ncclComm_t)ctx->comm, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(recv_host, d_recv, sizeof(float) * count * ctx->comm_size, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
int hp3d_allgather_dev(hp3d_ctx* ctx, const float* send_dev, int count, float* recv_host) {
    if (!ctx || !ctx->comm || !send_dev || !recv_host || count < 1) return HP3D_ERR_ARG;
    Rccl* R = rccl(ctx);
    if (!R) return HP3D_ERR_UNSUPPORTED;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CHK(gather_staging(ctx, (size_t)count * ctx->comm_size));
    float* d_recv = ctx->d_gather;
    NCCLCHK(ctx, R, R->AllGather(send_dev, d_recv, (size_t)count, ncclFloat, (ncclComm_t)ctx->comm, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(recv_host, d_recv, sizeof(float) * count * ctx->comm_size, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
int hp3d_comm_destroy(hp3d_ctx* ctx) {
    if (!ctx) return HP3D_ERR_ARG;
    if (!ctx->comm) return 0;
    Rccl* R = rccl(ctx);
    if (R) R->CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr; ctx->comm_size = 1; ctx->comm_rank = 0;
    return 0;
}
#else   // the CPU interpreter build has no RCCL
int hp3d_comm_unique_id(void*) { return HP3D_ERR_UNSUPPORTED; }
int hp3d_comm_init(hp3d_ctx* ctx, int, int, const void*) { HP3D_FAIL(ctx, HP3D_ERR_UNSUPPORTED, "no RCCL in the CPU interpreter build"); }
int hp3d_bcast_weights(hp3d_ctx* ctx, int) { HP3D_FAIL(ctx, HP3D_ERR_UNSUPPORTED, "no RCCL in the CPU interpreter build"); }
int hp3d_allgather(hp3d_ctx* ctx, const float*, int, float*) { HP3D_FAIL(ctx, HP3D_ERR_UNSUPPORTED, "no RCCL in the CPU interpreter build"); }
int hp3d_allgather_dev(hp3d_ctx* ctx, const float*, int, float*) { HP3D_FAIL(ctx, HP3D_ERR_UNSUPPORTED, "no RCCL in the CPU interpreter build"); }
int hp3d_comm_destroy(hp3d_ctx*) { return 0; }
#endif

uint32_t hp3d_crc32c(const void* data, size_t n) {
    static uint32_t T[8][256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            T[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xFF];
        init = true;
    }
    const unsigned char* p = (const unsigned char*)data;
    uint32_t crc = 0xFFFFFFFFu;
    while (n >= 8) {                       // slicing-by-8
        uint32_t lo, hi;
        memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
        lo ^= crc;
        crc = T[7][lo & 0xFF] ^ T[6][(lo >> 8) & 0xFF] ^ T[5][(lo >> 16) & 0xFF] ^ T[4][lo >> 24] ^
              T[3][hi & 0xFF] ^ T[2][(hi >> 8) & 0xFF] ^ T[1][(hi >> 16) & 0xFF] ^ T[0][hi >> 24];
        p += 8; n -= 8;
    }
    while (n--) crc = T[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
    return crc ^ 0xFFFFFFFFu;
}

}  // extern "C"

static long long comm_ranks(hp3d_ctx* ctx) {
#ifndef HP3D_EMU
    if (!ctx || !ctx->comm) return 0;
    Rccl* R = rccl(ctx);
    int n = 0;
    if (R && R->CommCount && R->CommCount((ncclComm_t)ctx->comm, &n) == ncclSuccess) return n;
    return ctx->comm_size;
#else
    (void)ctx;
    return 0;
#endif
}

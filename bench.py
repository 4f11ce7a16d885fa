#!/usr/bin/env python
"""bench.py -- images/sec through the full ColorHandPose3D pipeline on N MI355X of one node.

A "step" = one pass of ColorHandPose3DNetwork.inference() (HandSegNet -> mask/bbox/crop -> PoseNet2D ->
PosePrior/Viewpoint -> heat-map upsample -> 2-D keypoints) over one synthetic batch per GPU that is already resident
in HBM.  Weak scaling: every rank runs the same per-GPU batch (BASELINE config 4: 256 images over 8 GPUs = 32 per GPU);
the only collectives are the one-off RCCL weight broadcast (untimed setup) and the per-step all-gather of the
[B,21,3] keypoints (timed).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 8 --steps 20 --warmup 5          (self-launch: spawns the 8 ranks itself, see self_launch())
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

No PyTorch in the measured path: device memory, copies, RCCL and the rendezvous go through libhp3d.so's C ABI and
hand3d_amd/dist.py (the launcher above only starts the processes and exports RANK / WORLD_SIZE / MASTER_*).  torch is
imported in exactly one place, the `cpu_baseline` leg, where torch-CPU (oneDNN) convolutions ARE the baseline.

Protocol: W untimed warm-up steps; barrier + device sync; EXACTLY K steps timed with per-launch profiling OFF; device sync
+ barrier; max over ranks -> `value`.  Then a SEPARATE pass of K steps with HIP events around every launch (on the engine
stream) gives the per-kernel durations behind `roofline` (reported with its own ms/step so the two can be compared).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide: bf16/f16 MFMA, dense
PEAK_HBM_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=32, help='images per GPU per step')
    ap.add_argument('--height', type=int, default=320)
    ap.add_argument('--width', type=int, default=320)
    ap.add_argument('--workload', default='full', choices=['full', 'posenet'])
    ap.add_argument('--cpu-seconds', type=float, default=20.0,
                    help='budget of the cpu_baseline / epe_vs_oracle leg on rank 0 at N=1 (0 = skip)')
    ap.add_argument('--layers', action='store_true', help='print the per-layer table to stderr')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f16'],
                    help="f32 = exact f32 MFMA (headline); f16 = half-precision trunks, BASELINE config 5 (looser parity)")
    ap.add_argument('--graph', action='store_true', help='replay each step as one hipGraph (hp3d_set_option graph=1)')
    ap.add_argument('--option', action='append', default=[], metavar='KEY=VALUE', help='hp3d_set_option before the run (repeatable)')
    ap.add_argument('--no-host-path', action='store_true', help='skip the PCIe-inclusive host_path measurement')
    ap.add_argument('--no-other-configs', action='store_true',
                    help="skip the `other_configs` leg (BASELINE.json's other configurations, measured after the primary line)")
    ap.add_argument('--no-pin', action='store_true', help='do not pin this rank to the CPUs of its GPU\'s NUMA node')
    return ap.parse_args()


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cgroup_cpu_quota():
    """CPUs this container may use according to its cgroup (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us); None = unlimited."""
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        return None if q == 'max' else round(float(q) / float(per), 2)
    except (OSError, ValueError):
        pass
    try:
        q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        return None if q <= 0 else round(q / per, 2)
    except (OSError, ValueError):
        return None


def oracle_leg(weights, imgs, hs, gpu_out, workload, budget_s, alg_flop_per_image):
    """rank 0, N = 1 only, two things on the very batch the GPU just processed:
    * `cpu_baseline` (kind "port"): the path as a BATCHED torch-CPU / oneDNN program (oracle/torch_port.py: the whole batch
      through every layer in one call, NCHW tensors kept in torch across layers, vectorised glue) -- SURVEY.md 8d's "CPU path
      timed beside it".  One untimed pass (oneDNN primitive creation), then whole-batch passes for about half the budget;
    * `epe_vs_oracle`: the strict oracle (oracle/nets.py: NumPy glue, one image per call) over the first images for the other
      half -- mean EPE (EvalUtil semantics) and worst heat-map / 3-D keypoint deviation of the GPU outputs."""
    import torch                                   # the baseline itself is torch-CPU; nothing else in this file uses it
    from oracle import general as OG
    from oracle import nets as onets
    from oracle import tf_ops as OT
    from oracle import torch_port as TP
    # thread count: SURVEY.md 8d says "all cores"; on many-core hosts (and in containers whose CPU quota is below nproc) that
    # over-subscribes oneDNN badly, so the count is picked by measurement, on the shape the baseline then runs at: a conv3_2-sized
    # layer (80x80, 256 -> 256) over the WHOLE batch (round 3 probed 8 images and picked 16 of 256 threads; the batch of 32 feeds more)
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    quota = cgroup_cpu_quota()
    port = TP.TorchPort(weights)
    nprobe = max(int(imgs.shape[0]), 8)
    xs = torch.from_numpy(np.random.default_rng(0).standard_normal((nprobe, 256, 80, 80)).astype(np.float32))
    probe_flop = 2.0 * 9 * 256 * 256 * 6400 * nprobe
    best = (1e30, 1)
    tried = {}
    cap = ncpu if quota is None else max(1, min(ncpu, int(quota + 0.999)))
    with torch.no_grad():
        for nt in sorted({cap, max(cap // 2, 1), max(cap // 4, 1), 128, 64, 32, 16, 8}):
            if nt > cap:
                continue
            torch.set_num_threads(nt)
            t0 = time.time()
            port.conv(xs, 'HandSegNet/conv3_2')              # untimed warm-up; a count that thrashes shows here already
            if time.time() - t0 > 8.0:
                tried[nt] = None
                continue
            t0 = time.time()
            port.conv(xs, 'HandSegNet/conv3_2')
            port.conv(xs, 'HandSegNet/conv3_2')
            tried[nt] = (time.time() - t0) / 2
            best = min(best, (tried[nt], nt))
    torch.set_num_threads(best[1])
    probe_gflops = probe_flop / best[0] / 1e9
    run = (lambda: port.inference(imgs, hs)) if workload == 'full' else (lambda: port.pose2d(imgs))
    t0 = time.time()
    run()                                          # untimed: oneDNN creates its primitives per shape on the first call
    t_first = time.time() - t0
    passes, t_used = 0, 0.0
    while passes == 0 or t_used + t_used / passes <= budget_s / 2:
        t0 = time.time()
        run()
        t_used += time.time() - t0
        passes += 1
    n_img = imgs.shape[0]
    rate = passes * n_img / t_used
    cores = torch.get_num_threads()
    cpu = {"value": round(rate, 3), "unit": "images/s", "cores": cores, "kind": "port",
           "cpu_model": _cpu_model(), "host_cores": os.cpu_count(), "affinity_cores": ncpu, "cgroup_cpu_quota": quota,
           "thread_probe_gflops": {str(k): (None if v is None else round(probe_flop / v / 1e9, 1)) for k, v in sorted(tried.items())},
           "achieved_gflops": round(rate * alg_flop_per_image / 1e9, 1), "conv_probe_gflops": round(probe_gflops, 1),
           "sample": "%d pass(es) over the same %d-image batch (%.1f s; first, untimed pass %.1f s) through the batched torch-CPU / "
                     "oneDNN float32 port of the path (oracle/torch_port.py), torch.set_num_threads(%d) = the fastest of the "
                     "counts in thread_probe_gflops on a conv3_2-sized layer over the whole batch (%.0f GFLOP/s there) -- a CPU "
                     "restatement baseline, not TensorFlow 1.3"
                     % (passes, n_img, t_used, t_first, cores, probe_gflops)}
    # ---- parity of the GPU outputs against the strict oracle (per image)
    OT.CONV_BACKEND = 'torch'
    util = OG.EvalUtil()
    worst_kp3d = worst_map = 0.0
    n, t_par = 0, 0.0
    try:
        while n < imgs.shape[0] and (n == 0 or t_par + t_par / n <= budget_s / 2):
            t0 = time.time()
            if workload == 'full':
                o = onets.inference(weights, imgs[n:n + 1], hs[n:n + 1], True)
            else:
                o = onets.posenet2d(weights, imgs[n:n + 1])
            t_par += time.time() - t0
            if workload == 'full':
                util.feed(o[5][0], np.ones(21), gpu_out['coord3d'][n])
                worst_kp3d = max(worst_kp3d, float(np.abs(o[5][0] - gpu_out['coord3d'][n]).max()))
                worst_map = max(worst_map, float(np.abs(o[4][0, ::8, ::8] - gpu_out['sm32'][n]).max()))
            else:
                worst_map = max(worst_map, float(np.abs(o[2][0] - gpu_out['sm32'][n]).max()))
            n += 1
    finally:
        OT.CONV_BACKEND = 'numpy'
    par = {"images": n, "max_abs_err_heatmap32": worst_map, "tolerance_heatmap": 1e-3}
    if workload == 'full':
        par.update({"mean_epe": float(util.get_measures(0.0, 0.05, 20)[0]), "max_abs_err_kp3d": worst_kp3d,
                    "tolerance_kp3d": 1e-4,
                    "what": "mean end-point error (EvalUtil, utils/general.py:522-611) of the GPU keypoint_coord3d against the "
                            "oracle's on the same images, normalised units"})
    return cpu, par


CONV_FAMILIES = ('conv_wino', 'conv_wino2', 'conv_wino4', 'conv_wino4s', 'conv_mfma', 'conv_h16', 'conv_first_3x3_c3', 'conv_pw2')


def families(rows):
    """Per-launch profile rows (name, kernel, ms, flops, bytes) -> {family: [ms, flops, bytes, launches, executed flops]}, total ms.
    Families: conv_wino / conv_wino2 (Winograd F(2x2,3x3) forms of the 3x3 / 7x7 layers), conv_wino4 (F(4x4,3x3), incl. the wide-item
    and 16-tile forms), conv_mfma (direct implicit GEMM), conv_h16 (half-precision trunk kernel incl. its pooled, fused-first-block,
    7x7 and 1x1 forms), conv_first (conv1_1), conv_pw2 (the 1x1 head pairs as one launch), everything else under its own kernel name."""
    fam = {}
    for name, kern, ms, fl, by in rows:
        k = ('conv_wino4s' if kern.startswith('conv_wino4s') else
             'conv_wino4' if kern.startswith(('conv_wino4', 'conv_wino7')) else 'conv_wino2' if kern.startswith('conv_wino2') else
             'conv_wino' if kern.startswith('conv_wino') else 'conv_mfma' if kern.startswith('conv_mfma') else
             'conv_h16' if kern.startswith('conv_h16') else 'conv_first_3x3_c3' if kern.startswith('conv_first') else
             'conv_pw2' if kern.startswith('conv_pw2') else kern)
        # multiply-adds the matrix cores execute per direct-form multiply-add: Winograd F(2x2,3x3) 16/36; a 7x7 filter as
        # nine 3x3 blocks of its zero-extended 9x9 form, minus the structurally zero planes of the edge blocks (round 3):
        # (4*16 + 4*12 + 9) = 121 plane products per 4*49
        # conv_wino4 (F(4x4,3x3)): 36 products per 16 outputs = 36/144 of the direct form; a 7x7 filter as nine such blocks minus
        # their structurally zero planes: (4*36 + 4*30 + 25) = 289 plane products per 16*49
        # conv_wino7 (F(4x4,4x4) over the 7x7 filter's four 4x4-tap blocks, round 5; counted in the conv_wino4 family = "Winograd with 4x4
        # output tiles on the f32 matrix cores"): (49 + 42 + 42 + 36) = 169 plane products per 16*49
        # conv_wino4s (round 6: the same F(4x4,3x3) on the bf16 matrix pipe, three bfloat16 pieces per operand, SIX products per float32 product):
        # 6 * 36/144 bf16 multiply-adds per direct-form multiply-add, priced against the dense bf16 peak
        exe = (6.0 * 36.0 / 144.0 if k == 'conv_wino4s' else
               (169.0 / 784.0 if kern.startswith('conv_wino7') else 289.0 / 784.0 if 'as7x7' in kern else 36.0 / 144.0) if k == 'conv_wino4' else
               (121.0 / 196.0 if 'as7x7' in kern else 16.0 / 36.0) if k in ('conv_wino', 'conv_wino2') else 1.0)
        f = fam.setdefault(k, [0.0, 0.0, 0.0, 0, 0.0])
        # (conv_first_touch: the read pass that warms the memory-side cache for conv1_1's gathers -- its TIME belongs to the family, it is no launch
        #  of the kernel and adds no algorithmic bytes)
        f[0] += ms; f[1] += fl; f[2] += by; f[3] += 0 if kern == 'conv_first_touch' else 1; f[4] += fl * exe
    return fam, max(sum(v[0] for v in fam.values()), 1e-9)


def roof_of_family(fam, k, total_ms, peak):
    ms, fl, by, n, fle = fam[k]
    sec = ms * 1e-3
    if k == 'conv_first_3x3_c3':          # the one HBM-bound convolution (0.2 GFLOP per 13 MB of output per image)
        ach = by / sec / 1e9 if sec > 0 else 0.0
        return {"bound": "hbm", "kernel": k, "achieved": round(ach, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                "frac": round(ach / PEAK_HBM_GBPS, 4), "launches": n, "avg_launch_ms": round(ms / max(n, 1), 4),
                "alg_mbytes_per_launch": round(by / max(n, 1) / 1e6, 2), "share_of_gpu_time": round(ms / total_ms, 4)}
    alg = fl / sec / 1e12 if sec > 0 else 0.0
    exe = fle / sec / 1e12 if sec > 0 else 0.0
    # `achieved` / `frac` = what the matrix cores EXECUTE against their dense peak (<= 1 by construction);
    # `achieved_algorithmic` = direct-form FLOPs / time, which Winograd lifts above the executed rate
    return {"bound": "mfma", "kernel": k, "achieved": round(exe, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(exe / peak, 4), "achieved_algorithmic": round(alg, 2),
            "algorithmic_over_executed": round(alg / exe, 3) if exe > 0 else None,
            "launches": n, "avg_launch_ms": round(ms / max(n, 1), 4),
            "alg_gflop_per_launch": round(fl / max(n, 1) / 1e9, 3),
            "alg_mbytes_per_launch": round(by / max(n, 1) / 1e6, 2),
            "share_of_gpu_time": round(ms / total_ms, 4)}


def measure_config(eng, tag, what, call_site, workload, B, H, W, steps, warmup, dtype, img_np, hs_np):
    """One BASELINE.json configuration on an engine whose weights are already finalized: W warm-up steps, K timed steps
    (inputs resident in HBM, per-launch profiling off, device sync after every step as in the primary line), then a separate
    event-timed pass for the dominant conv family and its executed fraction of the dense matrix-core peak.  Returns the record
    and the device outputs of one more, untimed call (for the parity spot check)."""
    from hand3d_amd import arch
    d_img, d_hs = eng.to_device(img_np), eng.to_device(hs_np)
    d_coord = eng.dev_alloc(B * 63 * 4)
    d_kphw = eng.dev_alloc(B * 42 * 8)
    d_kpmap = eng.dev_alloc(B * 256 * 256 * 21 * 4) if workload == 'full' else None
    d_sm = [eng.dev_alloc(B * 32 * 32 * 21 * 4) for _ in range(3)] if workload == 'posenet' else None
    eng.sync()

    def step():
        if workload == 'full':
            eng.infer_full_dev(B, H, W, int(d_img), int(d_hs), kpmap=int(d_kpmap), coord3d=int(d_coord), kp_hw=int(d_kphw))
        else:
            eng._chk(eng.lib.hp3d_posenet2d_dev(eng.h, B, 256, 256, int(d_img), int(d_sm[0]), int(d_sm[1]), int(d_sm[2])))      # (a failing call must not be timed as a valid one)
        eng.sync()

    eng.set_profiling(0)
    for _ in range(warmup):
        step()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    eng.sync()
    dt = time.perf_counter() - t0
    psteps = max(1, min(steps, 5))
    eng.set_profiling(2)
    for _ in range(psteps):
        step()
    eng.sync()
    rows = eng.profile()
    eng.set_profiling(0)
    fam, total_ms = families(rows)
    conv = [k for k in fam if k in CONV_FAMILIES]
    dom = max(conv, key=lambda k: fam[k][0])
    roof = roof_of_family(fam, dom, total_ms, PEAK_F32_MFMA_TFLOPS if dtype == 'f32' and dom != 'conv_wino4s' else PEAK_F16_MFMA_TFLOPS)
    fl = arch.pipeline_flops(H, W)
    rec = {"config": tag, "what": what, "call_site": call_site, "dtype": dtype, "batch": B, "height": H, "width": W,
           "images_per_s": round(B * steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps, "warmup": warmup,
           "alg_gflop_per_image": round((fl['total'] if workload == 'full' else fl['posenet']) / 1e9, 2),
           "dominant_family": dom, "executed_frac_of_dense_peak": roof["frac"], "executed_tflops": roof["achieved"],
           "family_share_of_gpu_time": roof["share_of_gpu_time"], "profiled_ms_per_step": round(total_ms / psteps, 4),
           "profiled_pass_note": "events around every launch: the lifting towers and the first-touch pass run in ONE stream there (the timed region overlaps them)"}
    step()
    out = {}
    if workload == 'full':
        out['coord3d'] = eng.to_host(d_coord, (B, 21, 3))
        out['sm32'] = eng.to_host(d_kpmap, (1, 256, 256, 21))[:, ::8, ::8]
    else:
        out['sm32'] = eng.to_host(d_sm[2], (B, 32, 32, 21))
    for b in [d_img, d_hs, d_coord, d_kphw, d_kpmap] + (d_sm or []):
        if b is not None:
            b.free()
    return rec, out


def other_configs_measure(eng, weights, a, device):
    """BASELINE.json's OTHER configurations from the command the driver runs: C1 (run.py's call, B = 1 at
    240x320), C2 (eval2d_gt_cropped.py's PoseNet2D-only call, B = 1), the C4 per-GPU shard at eval_full.py's input size (B = 32 at
    240x320), and C5's per-GPU shape (B = 128 at 480x640, half-precision trunks; on a second context of the same device).  Each
    entry: images/s and ms/step of its own timed region, the dominant conv family with its executed fraction of the dense
    matrix-core peak (separate event-timed pass).  GPU work only: this leg runs BEFORE the cpu_baseline / oracle leg, because a
    torch-CPU region leaves its OpenMP threads spinning on the cgroup's cores for a while afterwards and the B = 1 configurations
    are bound by how fast the host enqueues their ~60 launches (measured: C2 2.33 ms/step right behind an oracle call, 0.78 alone).
    Returns (records, material for the parity spot checks)."""
    from hand3d_amd import Engine, synth
    res, keep = [], []
    plan = [
        ('C1', 'run.py forward pass shape: inference(), B=1, 240x320, f32', 'run.py:44-46', 'full', 1, 240, 320, 50, 10, 0),
        ('C2', 'PoseNet2D only on a ground-truth crop: inference_pose2d(), B=1, 256x256, f32', 'eval2d_gt_cropped.py:44-46', 'posenet', 1, 256, 256, 50, 10, 100),
        ('C4-shard@240x320', "config 4's per-GPU shard at eval_full.py's input size: inference(), B=32, 240x320, f32", 'eval_full.py:50-57', 'full', 32, 240, 320, 10, 3, 300),
        # round 6: the primary line's workload with the split-operand kernel on (conv_wino4s.hip: the filled 3x3 layers with
        # Cin >= 128 on v_mfma_f32_16x16x32_bf16, three bfloat16 pieces per operand, six products, float32 accumulate).  An OPTION, not the
        # headline: its per-layer error is not below conv_wino4's on every shape (profiles/r06_split_numerics.md)
        ('C3-split', "the primary workload with option wino4_split=auto: inference(), B=32, 320x320, f32 in / out, 3x3 layers with Cin >= 128 as "
         "bf16x3 split operands (6 products, f32 accumulate)", 'eval2d.py:50-58', 'full', 32, 320, 320, 10, 3, 200, {'wino4_split': ('auto', '0')}),
    ]
    for entry in plan:
        tag, what, site, workload, B, H, W, steps, warmup, seed = entry[:10]
        opts = entry[10] if len(entry) > 10 else {}
        try:
            img = synth.make_batch(seed, B, H, W)
            hs = synth.hand_sides(B)
            for key, (on, _off) in opts.items():
                eng.set_option(key, on)
            try:
                rec, out = measure_config(eng, tag, what, site, workload, B, H, W, steps, warmup, 'f32', img, hs)
            finally:
                for key, (_on, off) in opts.items():
                    eng.set_option(key, off)
            if opts:
                rec["options"] = {key: on for key, (on, _off) in opts.items()}
            rec["parity_spot"] = None
            res.append(rec)
            keep.append((rec, workload, img[0:1].copy(), hs[0:1].copy(), out))
        except Exception as e:            # one configuration failing must not take the primary line with it
            res.append({"config": tag, "error": "%s: %s" % (type(e).__name__, str(e)[:300])})
    # ---- C5 per-GPU shape: half-precision trunks on a second context (own weights blob, own arena: ~25 GB of the 288 GB)
    eng16 = None
    try:
        eng16 = Engine(device)
        eng16.load_weight_dict(weights)
        eng16.finalize_weights('f16')
        B, H, W = 128, 480, 640
        # uniform noise frames, seed 1000 + i; 16 distinct frames tiled 8x (host-side generation time, not GPU work)
        base = np.stack([np.random.default_rng(1000 + i).uniform(-0.5, 0.5, size=(H, W, 3)).astype(np.float32) for i in range(16)], 0)
        img = np.concatenate([base] * (B // 16), 0)
        hs = synth.hand_sides(B)
        rec, _ = measure_config(eng16, 'C5-shard', "config 5's per-GPU shape: inference(), B=128, 480x640, f16 trunks (f32 accumulate, f32 heads / "
                                "lifting / outputs)", 'run.py:44-46 at 480x640', 'full', B, H, W, 10, 2, 'f16', img, hs)
        rec["data"] = "16 distinct U(-0.5,0.5) frames (seeds 1000..1015) tiled to 128"
        rec["parity_spot"] = None
        fix = os.path.join(ROOT, 'tests', 'golden', 'c5_f16_480x640.npz')
        if a.cpu_seconds > 0 and os.path.exists(fix):
            g = np.load(fix)
            n_img = int(g['seg_small'].shape[0])
            fimg = synth.make_batch(int(g['seed0']), n_img, H, W)
            eng16.set_option('f16_impl', 'h16_force')        # the kernels the B = 128 shape runs on, also at the fixture's 2 images
            _, small = eng16.handsegnet(fimg, want_small=True)
            eng16.set_option('f16_impl', 'h16')
            rec["parity_spot"] = {"images": n_img, "max_abs_err_seg_logits": float(np.abs(small - g['seg_small']).max()), "tolerance": 2e-3,
                                  "against": "tests/golden/c5_f16_480x640.npz (oracle with the f16 rounding points, scripts/make_c5_fixture.py)"}
        res.append(rec)
    except Exception as e:
        res.append({"config": "C5-shard", "error": "%s: %s" % (type(e).__name__, str(e)[:300])})
    finally:
        if eng16 is not None:
            eng16.close()
    return res, keep


def other_configs_parity(keep, weights):
    """ONE-image parity spot check of each float32 configuration measured above: the strict oracle on image 0 (heat-maps 1e-3, 3-D
    keypoints 1e-4).  (C5's check against the committed oracle fixture needs no CPU arithmetic and is done with its measurement.)
    The oracle is the checker here, never the thing timed."""
    from oracle import nets as onets
    from oracle import tf_ops as OT
    OT.CONV_BACKEND = 'torch'
    try:
        for rec, workload, img, hs, out in keep:
            try:
                if workload == 'full':
                    o = onets.inference(weights, img, hs, True)
                    rec["parity_spot"] = {"image": 0, "max_abs_err_heatmap32": float(np.abs(o[4][0, ::8, ::8] - out['sm32'][0]).max()),
                                          "tolerance_heatmap": 1e-3, "max_abs_err_kp3d": float(np.abs(o[5][0] - out['coord3d'][0]).max()),
                                          "tolerance_kp3d": 1e-4, "against": "oracle/nets.py:inference (float64-accumulating restatement)"}
                else:
                    o = onets.posenet2d(weights, img)
                    rec["parity_spot"] = {"image": 0, "max_abs_err_heatmap32": float(np.abs(o[2][0] - out['sm32'][0]).max()),
                                          "tolerance_heatmap": 1e-3, "against": "oracle/nets.py:posenet2d"}
            except Exception as e:
                rec["parity_spot"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    finally:
        OT.CONV_BACKEND = 'numpy'


def traffic_record(dom, workload_str, dtype):
    """HBM bytes per launch of the dominant kernel come from rocprofv3 PMC passes of THIS command (counters cannot be read
    in-process): scripts/gpu_round.sh <tag> pmc -> scripts/summarize_prof.py -> profiles/<family>_traffic.json.  The number
    is quoted only for the workload string it was measured on, together with the stamp the summariser wrote into the file
    (profile tag, date, commit of the tree that was profiled)."""
    tpath = os.path.join(ROOT, 'profiles', '%s_traffic.json' % dom)
    try:
        t = json.load(open(tpath))
        if t.get('workload') != workload_str or t.get('dtype', 'f32') != dtype:
            return None, None
        return round(t['hbm_bytes_per_launch']), {"file": os.path.relpath(tpath, ROOT), "stamp": t.get('stamp', t.get('source')),
                                                  "method": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) and WRITE_SIZE in separate "
                                                            "passes of this command, per launch"}
    except Exception:
        return None, None


def self_launch(a):
    """`python bench.py --gpus N` (N > 1) with no launcher in the environment: start the N ranks ourselves -- one process per
    GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT=<free port> exactly as torch.distributed.run
    would export them, plus a per-run random HP3D_RDZV_SECRET for the TCP rendezvous -- forward rank 0's JSON line and
    return the worst exit code.  (The reference has no counterpart: one tf.Session, run.py:50.)"""
    import secrets
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    entry = os.environ.get('HP3D_BENCH_ENTRY', os.path.abspath(__file__))     # tests substitute a stand-in engine
    base = dict(os.environ, WORLD_SIZE=str(a.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                HP3D_RDZV_SECRET=secrets.token_hex(16), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    procs = []
    for r in range(a.gpus):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        # rank 0's stdout carries the ONE JSON line; whatever another rank prints goes to stderr
        procs.append(subprocess.Popen([sys.executable, entry] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0, got0 = b'', False
    rcs = [None] * a.gpus
    first_fail = None
    try:
        while any(rc is None for rc in rcs):
            for r, p in enumerate(procs):
                if rcs[r] is None and p.poll() is not None:
                    rcs[r] = p.returncode
                    if p.returncode != 0 and first_fail is None:
                        first_fail = time.time()
                        sys.stderr.write('bench.py: rank %d exited with code %d\n' % (r, p.returncode))
            if rcs[0] is None:
                try:                                    # drain rank 0's pipe while waiting (one JSON line: small)
                    o, _ = procs[0].communicate(timeout=0.2)
                    out0, got0 = o or b'', True           # a communicate() that returns has ALL of rank 0's output
                except subprocess.TimeoutExpired:
                    pass
            else:
                time.sleep(0.2)
            # a rank that died leaves the others waiting at the rendezvous / in a collective: give them 20 s, then stop them
            if first_fail is not None and time.time() - first_fail > float(os.environ.get('HP3D_BENCH_GRACE_S', '20')):
                for r, p in enumerate(procs):
                    if rcs[r] is None:
                        p.kill()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    # (a communicate() that timed out keeps what it had read inside the Popen object: only another communicate() returns it --
    #  reading procs[0].stdout directly lost the line whenever rank 0 exited between two polls)
    if not got0:
        try:
            o, _ = procs[0].communicate(timeout=30)
            out0 = o or b''
        except (OSError, ValueError, subprocess.TimeoutExpired):
            pass
    sys.stdout.write(out0.decode(errors='replace'))
    sys.stdout.flush()
    bad = [rc for rc in rcs if rc]
    return 0 if not bad else max(abs(rc) for rc in bad)


def visible_devices(isolated=False):
    """HIP devices this process can see, through the C ABI (hp3d_device_count: no context, no torch).  `isolated`: ask in a
    short-lived forked child, so that THIS process never initialises the HIP runtime (the self-launch parent only starts ranks
    and must not sit on every GPU while they are measured)."""
    from hand3d_amd import _lib
    if not isolated:
        return _lib.device_count()
    import multiprocessing as mp
    ctx = mp.get_context('fork')
    rd, wr = ctx.Pipe(duplex=False)

    def probe():
        try:
            wr.send(int(_lib.device_count()))
        except Exception:
            wr.send(-1)
    pr = ctx.Process(target=probe)
    pr.start()
    n = rd.recv() if rd.poll(120) else -1
    pr.join(10)
    if n < 0:
        raise RuntimeError('hp3d_device_count failed in the probe process')
    return n


def pin_to_gpu_numa(device):
    """Pin this rank to the CPUs of its GPU's NUMA node (host dispatch is what limits batch-shard scaling, and a
    rank whose launch thread sits on the far socket pays for every one of its ~85 launches per step).  PCI address from
    hp3d_device_pci_bus_id, node from /sys/bus/pci/devices/<addr>/numa_node, CPUs from /sys/devices/system/node/nodeN/cpulist,
    intersected with the affinity the launcher / cgroup already allows.  Returns a description for `config`, None when the
    sysfs entries are absent, the node is -1 or the intersection is empty (then nothing is changed)."""
    try:
        from hand3d_amd import _lib
        addr = _lib.device_pci_bus_id(device).lower()
        node = int(open('/sys/bus/pci/devices/%s/numa_node' % addr).read().strip())
        if node < 0:
            return None
        cpus = set()
        for part in open('/sys/devices/system/node/node%d/cpulist' % node).read().strip().split(','):
            lo, _, hi = part.partition('-')
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = cpus & set(os.sched_getaffinity(0))
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
        return {"pci": addr, "numa_node": node, "cpus": len(allowed)}
    except Exception:
        return None


def main():
    a = parse()
    # ROCr reads its flags when the HIP runtime initialises (the first HIP call of the process, hp3d_device_count included): the
    # dmabuf-IPC switch RCCL needs on this driver must be in the environment BEFORE that, also for ranks an external launcher started
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    launched = 'RANK' in os.environ and 'MASTER_ADDR' in os.environ
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # `--gpus N` on a box that shows fewer than N devices: say so in ONE line and leave with a code of its own BEFORE any rank starts
    # (self-launch parent: counted in a forked probe, the parent itself never touches HIP) -- not N - n ranks dying one by one inside a
    # rendezvous while the others wait for their timeout.  A rank under a launcher only needs ITS device (LOCAL_RANK): with per-rank
    # HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES isolation or a multi-node world it legitimately sees fewer devices than WORLD_SIZE.
    if launched:
        ndev = visible_devices()
        if local >= ndev:
            sys.stderr.write('bench.py: rank %d has LOCAL_RANK %d but only %d HIP device(s) are visible to this process '
                             '(hp3d_device_count; check HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES): not joining\n' % (rank, local, ndev))
            sys.exit(3)
    else:
        ndev = visible_devices(isolated=a.gpus > 1)
        if ndev < a.gpus:
            sys.stderr.write('bench.py: --gpus %d but only %d HIP device(s) are visible to this process (hp3d_device_count; check '
                             'HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES): not starting\n' % (a.gpus, ndev))
            sys.exit(3)
    if a.gpus > 1 and not launched:
        sys.exit(self_launch(a))
    if launched and world != a.gpus:
        sys.stderr.write('bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks: refusing to report a line whose '
                         'n_gpus would not be what was asked for\n' % (a.gpus, os.environ.get('WORLD_SIZE', '1')))
        sys.exit(2)
    # keep stdout for the ONE JSON line: route everything else (RCCL's version banner, library chatter)
    # written to fd 1 during the run to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # multi-rank runs: every rank on the CPUs next to its GPU (before the engine starts its threads); recorded in `config`
    affinity = pin_to_gpu_numa(local) if (launched and not a.no_pin) else None

    from hand3d_amd import Engine, synth, arch
    from hand3d_amd.dist import Rendezvous, ShardedPipeline
    eng = Engine(local)     # raises if libhp3d.so is missing or no GPU is visible: no fallback
    rdzv = Rendezvous.from_env() if launched else Rendezvous(0, 1)
    sp = ShardedPipeline(eng, rank, world, rdzv)
    B, H, W = a.batch, a.height, a.width
    weights = synth.make_weights() if rank == 0 else None
    # native RCCL (hp3d_comm_init + hp3d_bcast_weights) whenever a launcher started us -- also at world size 1, so that a
    # single-GPU box still exercises the exchange
    comm_mode, comm_err = ('rccl' if launched else 'none'), None
    try:
        sp.sync_weights(weights, dtype=a.dtype, use_comm=launched)
    except Exception as e:                      # RCCL could not be set up on this box
        if not launched:
            raise
        comm_err = '%s: %s' % (type(e).__name__, e)
    if launched and any(rdzv.allgather(comm_err)):
        # every rank takes the same degraded path: own copy of the (seeded, identical) weights, gathers over the rendezvous
        errs = [x for x in rdzv.allgather(comm_err) if x]
        comm_mode = 'tcp-fallback (%s)' % errs[0][:200]
        sys.stderr.write('bench.py rank %d: RCCL unavailable, weights loaded per rank and keypoints gathered over TCP: %s\n' % (rank, errs[0]))
        sp.use_tcp_only()
        eng.load_weight_dict(synth.make_weights())
        eng.finalize_weights(a.dtype)
    # what RCCL itself says the communicator spans (ncclCommCount through hp3d_get_counter): 0 = no communicator
    rccl_ranks = eng.counter('comm_ranks') if comm_mode == 'rccl' else 0
    if comm_mode == 'rccl' and rccl_ranks != world:
        raise RuntimeError('RCCL communicator spans %d ranks, the launcher started %d' % (rccl_ranks, world))
    if a.graph:
        eng.set_option('graph', '1')
    for kv in a.option:
        eng.set_option(*kv.split('=', 1))

    # synthetic inputs, resident in HBM before the timed region
    Hi, Wi = (H, W) if a.workload == 'full' else (256, 256)
    img_np = synth.make_batch(1000 + rank * B, B, Hi, Wi)
    hs_np = synth.hand_sides(B)
    d_img, d_hs = eng.to_device(img_np), eng.to_device(hs_np)
    d_coord = eng.dev_alloc(B * 63 * 4)
    d_kpmap = eng.dev_alloc(B * 256 * 256 * 21 * 4)
    d_kphw = eng.dev_alloc(B * 42 * 8)
    d_sm = [eng.dev_alloc(B * 32 * 32 * 21 * 4) for _ in range(3)]
    eng.sync()

    def step():
        if a.workload == 'full':
            eng.infer_full_dev(B, H, W, int(d_img), int(d_hs), kpmap=int(d_kpmap), coord3d=int(d_coord), kp_hw=int(d_kphw))
            eng.sync()
            return sp.gather_keypoints(d_coord, B) if world > 1 else None
        eng.lib.hp3d_posenet2d_dev(eng.h, B, 256, 256, int(d_img), int(d_sm[0]), int(d_sm[1]), int(d_sm[2]))
        eng.sync()
        return None

    for _ in range(a.warmup):
        step()
    # ---- the timed region: profiling off (no per-launch events; hipGraph replay, if asked for, really replays) ----------
    # The region of EXACTLY K steps (barrier + device sync on both sides, max over ranks) is run REPEATS times back to back and the MEDIAN
    # region is the one reported (one 0.24 s sample cannot resolve the 0.5 % steps the tuning log works with;
    # box-to-box spread is +-1.5 %); `value_min` / `value_max` carry the other two.
    eng.set_profiling(0)
    REPEATS = 3
    region_dt = []
    for _ in range(REPEATS):
        rdzv.barrier()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        eng.sync()
        rdzv.barrier()
        region_dt.append(rdzv.max(time.perf_counter() - t0))
    dt = sorted(region_dt)[REPEATS // 2]

    # ---- separate profiled pass: HIP events around every launch on the engine stream, accumulated over K steps ----------
    eng.set_profiling(2)
    t1 = time.perf_counter()
    for _ in range(a.steps):
        step()
    eng.sync()
    dt_prof = time.perf_counter() - t1
    rows = eng.profile()
    eng.set_profiling(0)

    if rank == 0:
        fam, total_ms = families(rows)
        peak = PEAK_F32_MFMA_TFLOPS if a.dtype == 'f32' else PEAK_F16_MFMA_TFLOPS

        def roof_of(k):
            return roof_of_family(fam, k, total_ms, PEAK_F16_MFMA_TFLOPS if k == 'conv_wino4s' else peak)
        dom = max(fam, key=lambda k: fam[k][0])
        roof = roof_of(dom)
        if dom == 'conv_h16':
            roof["note"] = ("half-precision 3x3 trunk kernel (all instantiations: 1 / 2 / 4 cout blocks per wave, pooled, fused conv1_1 + conv1_2); "
                            "direct form, so executed = algorithmic; the fused launches count conv1_1's FLOPs too")
        if dom in ('conv_wino', 'conv_wino2'):
            roof["note"] = ("float32 Winograd F(2x2,3x3): executes 16/36 of the direct-form multiply-adds (7x7 layers as nine 3x3 "
                            "blocks without their structurally zero planes: 121/196); frac is the executed matrix-core rate over the dense f32 MFMA peak")
        if dom == 'conv_wino4s':
            roof["note"] = ("Winograd F(4x4,3x3) with split operands (option wino4_split, conv_wino4s.hip): three bfloat16 pieces per operand, six products per "
                            "float32 product on v_mfma_f32_16x16x32_bf16, float32 accumulate -- executes 6 * 36/144 bf16 multiply-adds per direct-form "
                            "multiply-add; frac is that rate over the dense bf16 MFMA peak")
        if dom == 'conv_wino4':
            roof["note"] = ("float32 Winograd with 4x4 output tiles: F(4x4,3x3) (conv_wino4.hip) executes 36/144 of the direct-form multiply-adds; "
                            "the 7x7 layers run as F(4x4,4x4) over the filter's four 4x4-tap blocks (conv_wino7.hip, kernel conv_wino7_*: 169/784 "
                            "after the structurally zero planes) where the launch fills the chip, else as nine 3x3 blocks on conv_wino4.hip "
                            "(289/784); frac is the executed matrix-core rate over the dense f32 MFMA peak, achieved_algorithmic the "
                            "direct-form FLOPs over the same time")
        workload_str = ("ColorHandPose3DNetwork.inference, %dx%dx3 f32 in HBM, %d images/GPU/step" % (H, W, B)) \
            if a.workload == 'full' else ("inference_pose2d, 256x256x3 f32 in HBM, %d images/GPU/step" % B)
        roof["traffic"], roof["traffic_source"] = traffic_record(dom, workload_str, a.dtype)
        roof["timing"] = "HIP events on the engine stream around each launch, separate pass of %d steps (%.3f ms/step profiled)" % (
            a.steps, dt_prof / a.steps * 1e3)
        others = [roof_of(k) for k in sorted(fam, key=lambda k: -fam[k][0])
                  if k != dom and k in CONV_FAMILIES]
        if a.layers:
            agg = {}
            for name, kern, ms_, fl_, by_ in rows:
                r = agg.setdefault((name, kern), [0.0, 0.0, 0.0, 0])
                r[0] += ms_; r[1] += fl_; r[2] += by_; r[3] += 1
            print("%-28s %-34s %9s %9s %9s" % ('layer', 'kernel', 'ms/step', 'TFLOP/s', 'GB/s'), file=sys.stderr)
            for (name, kern), (ms_, fl_, by_, n_) in agg.items():
                print("%-28s %-34s %9.3f %9.1f %9.0f" % (name, kern, ms_ / a.steps, fl_ / ms_ / 1e9 if ms_ else 0,
                                                         by_ / ms_ / 1e6 if ms_ else 0), file=sys.stderr)

        # ---- PCIe-inclusive rate (never `value`): pinned host frames in -> 3-D + 2-D keypoints out on the host, the upload
        #      of batch n+1 overlapped with the kernels of batch n (hp3d_upload_async on a second stream) ------------------
        host_path = None
        if world == 1 and a.workload == 'full' and not a.no_host_path:
            pin = [eng.pinned_empty((B, H, W, 3)) for _ in range(2)]
            dev = [d_img, eng.dev_alloc(B * H * W * 3 * 4)]
            for p in pin:
                p[...] = img_np
            eng.upload_async(dev[0], pin[0]); eng.wait_upload()
            outs = None
            th = time.perf_counter()
            for i in range(a.steps):
                cur, nxt = i & 1, (i + 1) & 1
                if i + 1 < a.steps:
                    eng.upload_async(dev[nxt], pin[nxt])
                eng.infer_full_dev(B, H, W, int(dev[cur]), int(d_hs), coord3d=int(d_coord), kp_hw=int(d_kphw))
                outs = (eng.to_host(d_coord, (B, 21, 3)), eng.to_host(d_kphw, (B, 21, 2), np.float64))     # blocking D2H
                eng.wait_upload()
            dth = time.perf_counter() - th
            host_path = {"value": round(B * a.steps / dth, 2), "unit": "images/s", "ms_per_step": round(dth / a.steps * 1e3, 3),
                         "what": "pinned float32 frames on the host -> keypoint_coord3d [B,21,3] + 2-D keypoints [B,21,2] on the "
                                 "host per step; H2D of step n+1 overlapped with the kernels of step n; heat-maps stay on the device"}
            dev[1].free()
            # SURVEY 8d's metric as written -- "wall-clock from host arrays in to keypoints out" -- on the entry point a caller with camera
            # frames would use: UINT8 frames in pinned host memory -> hp3d_infer_full_kp_u8 (normalise on the device, the
            # whole path, detect_keypoints / trafo_coords on the device) -> 3-D and 2-D keypoints on the host, blocking, a DIFFERENT batch
            # every step (two alternating seeds), nothing overlapped by the caller.  4x less PCIe than the float32 path above.
            u8 = [eng.pinned_empty((B, H, W, 3), np.uint8) for _ in range(2)]
            for j, p8 in enumerate(u8):
                p8[...] = np.clip(np.rint((synth.make_batch(5000 + 977 * j, B, H, W) + 0.5) * 255.0), 0, 255).astype(np.uint8)
            c3 = np.empty((B, 21, 3), np.float32)
            khw = np.empty((B, 21, 2), np.float64)

            def step_u8(i):
                eng._chk(eng.lib.hp3d_infer_full_kp_u8(eng.h, B, H, W, u8[i & 1].ctypes.data, H, W, hs_np.ctypes.data, None, None, None, None, None,
                                                       c3.ctypes.data, None, None, khw.ctypes.data))
            for i in range(2):
                step_u8(i)
            seen = []
            tu = time.perf_counter()
            for i in range(a.steps):
                step_u8(i)
                if i < 2:
                    seen.append(c3.copy())
            dtu = time.perf_counter() - tu
            host_path["value_host_u8"] = round(B * a.steps / dtu, 2)
            host_path["ms_per_step_host_u8"] = round(dtu / a.steps * 1e3, 3)
            host_path["what_host_u8"] = ("uint8 frames in pinned host memory -> hp3d_infer_full_kp_u8 -> keypoint_coord3d [B,21,3] + 2-D keypoints "
                                         "[B,21,2] on the host, blocking call per step, two different batches alternating (SURVEY 8d's metric as written)")
            host_path["host_u8_batches_differ"] = bool(len(seen) == 2 and not np.array_equal(seen[0], seen[1]))

        # BASELINE.json's other configurations: GPU measurements first (see other_configs_measure), their oracle spot checks after the CPU leg
        oc = oc_keep = None
        oc_wall = 0.0
        if world == 1 and a.workload == 'full' and a.dtype == 'f32' and not a.no_other_configs:
            t_oc = time.perf_counter()
            oc, oc_keep = other_configs_measure(eng, weights, a, local)
            oc_wall = time.perf_counter() - t_oc

        cpu = parity = None
        if world == 1 and a.cpu_seconds > 0:
            gpu_out = {'coord3d': eng.to_host(d_coord, (B, 21, 3)) if a.workload == 'full' else None}
            if a.workload == 'full':
                eng.infer_full_dev(B, H, W, int(d_img), int(d_hs), kpmap=int(d_kpmap), coord3d=int(d_coord))
                eng.sync()
                gpu_out['coord3d'] = eng.to_host(d_coord, (B, 21, 3))
                nchk = min(B, 32)
                gpu_out['sm32'] = eng.to_host(d_kpmap, (nchk, 256, 256, 21))[:, ::8, ::8]
            else:
                gpu_out['sm32'] = eng.to_host(d_sm[2], (B, 32, 32, 21))
            fl_ = arch.pipeline_flops(H, W)
            cpu, parity = oracle_leg(weights, img_np[:32], hs_np[:32], gpu_out, a.workload, a.cpu_seconds,
                                     fl_['total'] if a.workload == 'full' else fl_['posenet'])
        n_img = B * world * a.steps
        fl_img = arch.pipeline_flops(H, W)
        res = {
            "metric": "images/sec full pipeline (HandSegNet+crop+PoseNet2D+PosePrior/Viewpoint, RGB -> 21x3D kpts)"
                      if a.workload == 'full' else "images/sec PoseNet2D only (256x256 crops)",
            "value": round(n_img / dt, 2), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "value_min": round(n_img / max(region_dt), 2), "value_max": round(n_img / min(region_dt), 2), "timed_regions": REPEATS,
            "value_host_u8": host_path.get("value_host_u8") if host_path else None,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic (seeded images, seeded fan-in-scaled weights)",
            "config": {"workload": workload_str,
                       "headline": "`value` = inputs resident in HBM when the timed region starts, the MEDIAN of three K-step regions (value_min / "
                                   "value_max: the other two); SURVEY 8d's metric as written -- host arrays in, keypoints out -- from uint8 frames is "
                                   "`value_host_u8` (= host_path.value_host_u8), from float32 frames host_path.value",
                       "global_batch": B * world, "per_gpu_batch": B, "height": H, "width": W,
                       "parallelism": "batch-shard x%d, one process per GPU, no data-path collective; weights by hp3d_bcast_weights "
                                      "and a per-step keypoint all-gather (RCCL through the C ABI, TCP rendezvous; no torch)" % world,
                       "comm": comm_mode, "rccl_ranks": rccl_ranks, "hipgraph": bool(a.graph), "options": a.option,
                       "cpu_affinity": affinity,
                       "alg_gflop_per_image": round((fl_img['total'] if a.workload == 'full' else fl_img['posenet']) / 1e9, 2)},
            "roofline": roof, "roofline_other_conv": others, "cpu_baseline": cpu, "epe_vs_oracle": parity,
            "host_path": host_path,
        }
        if oc is not None:
            if a.cpu_seconds > 0:
                t_oc = time.perf_counter()
                other_configs_parity(oc_keep, weights)
                oc_wall += time.perf_counter() - t_oc
            res["other_configs"] = oc
            res["other_configs_wall_s"] = round(oc_wall, 1)
        if a.graph:
            res["config"]["hipgraph_replays"] = eng.counter('graph_replays')
        os.write(json_fd, (json.dumps(res) + '\n').encode())
    sp.close()
    if sp.comm_abandoned:
        # an RCCL set-up call is still stuck in its worker thread: tearing the engine down (hipStreamDestroy, library destructors) could
        # wait on whatever that call holds -- the line is out, leave without the destructors
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == '__main__':
    main()

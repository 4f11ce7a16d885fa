#!/usr/bin/env python
"""bench.py -- images/sec through the full ColorHandPose3D pipeline on N MI355X of one node.

A "step" = one pass of ColorHandPose3DNetwork.inference() (HandSegNet -> mask/bbox/crop ->
PoseNet2D -> PosePrior/Viewpoint -> heat-map upsample) over one synthetic batch per GPU that is
already resident in HBM.  Weak scaling: every rank runs the same per-GPU batch (BASELINE config 4:
256 images over 8 GPUs = 32 per GPU); the only collectives are the one-off RCCL weight broadcast
(untimed setup) and the per-step gather of the [B,21,3] keypoints (timed).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

torch is used for device memory and torch.distributed only (plumbing); all compute is libhp3d.so.
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
    ap.add_argument('--cpu-images', type=int, default=5, help='oracle images timed for cpu_baseline (0 = skip)')
    ap.add_argument('--layers', action='store_true', help='print the per-layer table to stderr')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f16'],
                    help="f32 = exact f32 MFMA (headline); f16 = half-precision trunks, BASELINE config 5 (looser parity)")
    ap.add_argument('--comm', default='torch', choices=['torch', 'native'],
                    help="weight broadcast: torch.distributed (RCCL) or the engine's own hp3d_comm_init + hp3d_bcast_weights")
    ap.add_argument('--graph', action='store_true', help='replay each step as one hipGraph (hp3d_set_option graph=1; small batches)')
    ap.add_argument('--streams', type=int, default=1, help='engine contexts (HIP streams) per GPU; the per-GPU batch is split across them')
    return ap.parse_args()


def cpu_baseline(weights, H, W, n_images, workload):
    """The oracle (NumPy port of the reference's TF1.3 graph; OpenBLAS threads) on the host cores."""
    from oracle import nets as onets
    from hand3d_amd import synth
    imgs = synth.make_batch(9000, n_images, H if workload == 'full' else 256, W if workload == 'full' else 256)
    hs = synth.hand_sides(n_images)
    t0 = time.time()
    for i in range(n_images):
        if workload == 'full':
            onets.inference(weights, imgs[i:i + 1], hs[i:i + 1], True)
        else:
            onets.posenet2d(weights, imgs[i:i + 1])
    dt = time.time() - t0
    threads = os.cpu_count()
    try:        # the GEMMs inside the oracle run on NumPy's BLAS pool: report the threads it really uses
        from threadpoolctl import threadpool_info
        blas = [p['num_threads'] for p in threadpool_info() if p.get('user_api') == 'blas']
        if blas:
            threads = max(blas)
    except Exception:
        pass
    return {"value": round(n_images / dt, 4), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": "%d image(s) of the same workload through the NumPy oracle (float32, BLAS pool of %d threads on a "
                      "%d-core host), %.1f s" % (n_images, threads, os.cpu_count(), dt)}


def main():
    a = parse()
    # keep stdout for the ONE JSON line: route everything else (RCCL's version banner, library chatter)
    # written to fd 1 during the run to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    import torch
    import torch.distributed as dist
    use_dist = 'RANK' in os.environ and 'MASTER_ADDR' in os.environ   # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=torch.device('cuda', local))
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)

    from hand3d_amd import Engine, synth, arch
    from hand3d_amd.dist import ShardedPipeline, gather_keypoints
    eng = Engine(local)     # raises if libhp3d.so is missing: no fallback
    B, H, W = a.batch, a.height, a.width
    weights = synth.make_weights() if rank == 0 else None
    if a.comm == 'native':
        ShardedPipeline(eng, rank, world).sync_weights_native(weights, dtype=a.dtype)
    else:
        ShardedPipeline(eng, rank, world).sync_weights(weights, device=dev, dtype=a.dtype)
    # extra contexts on the same GPU: independent HIP streams whose kernels overlap (one context's tail /
    # prologue / launch gaps are filled by the other's bulk); they get the weights by a device-to-device blob copy
    engines = [eng]
    for _ in range(a.streams - 1):
        e2 = Engine(local)
        blob = torch.empty((eng.blob_bytes() + 3) // 4, dtype=torch.float32, device=dev)
        eng.blob_export(blob.data_ptr())
        e2.blob_import(blob.data_ptr(), eng.nets_mask())
        del blob
        engines.append(e2)
    if a.graph:
        for e in engines:
            e.set_option('graph', '1')
    assert B % len(engines) == 0, "--batch must be divisible by --streams"
    Bs = B // len(engines)

    # synthetic inputs, resident in HBM before the timed region
    Hi, Wi = (H, W) if a.workload == 'full' else (256, 256)
    img = torch.from_numpy(synth.make_batch(1000 + rank * B, B, Hi, Wi)).to(dev)
    hs = torch.from_numpy(synth.hand_sides(B)).to(dev)
    coord = torch.zeros(B, 21, 3, device=dev)
    kpmap = torch.empty(B, 256, 256, 21, device=dev)
    sm = [torch.empty(B, 32, 32, 21, device=dev) for _ in range(3)]
    torch.cuda.synchronize(dev)

    def step():
        if a.workload == 'full':
            for i, e in enumerate(engines):     # stream-ordered enqueue, no host sync in between
                e.infer_full_dev(Bs, H, W, img[i * Bs:].data_ptr(), hs[i * Bs:].data_ptr(),
                                 kpmap=kpmap[i * Bs:].data_ptr(), coord3d=coord[i * Bs:].data_ptr())
            for e in engines:
                e.sync()
            return gather_keypoints(coord, n_total=B * world)
        eng.lib.hp3d_posenet2d_dev(eng.h, B, 256, 256, img.data_ptr(), sm[0].data_ptr(), sm[1].data_ptr(), sm[2].data_ptr())
        eng.sync()
        return sm[2]

    for _ in range(a.warmup):
        step()
    for e in engines:
        e.set_profiling(2)
    # HIP events on the engine stream around every launch, accumulated over the K steps
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    rows = []
    for e in engines:
        rows += e.profile()
        e.set_profiling(0)
    if rank == 0:
        # ---- roofline of the dominant kernel family ------------------------------------------------------
        # families: conv_wino (Winograd F(2x2,3x3) form of the 3x3 layers), conv_mfma (direct implicit GEMM), glue
        fam = {}
        for name, kern, ms, fl, by in rows:
            k = 'conv_wino' if kern.startswith('conv_wino') else 'conv_mfma' if kern.startswith('conv_mfma') else kern
            # multiply-adds the matrix cores execute per direct-form multiply-add: Winograd F(2x2,3x3) 16/36; a 7x7 filter
            # as nine 3x3 blocks 9*16 per 4*49
            exe = (144.0 / 196.0 if 'as7x7' in kern else 16.0 / 36.0) if k == 'conv_wino' else 1.0
            f = fam.setdefault(k, [0.0, 0.0, 0.0, 0, 0.0])
            f[0] += ms; f[1] += fl; f[2] += by; f[3] += 1; f[4] += fl * exe
        total_ms = max(sum(v[0] for v in fam.values()), 1e-9)
        peak = PEAK_F32_MFMA_TFLOPS if a.dtype == 'f32' else PEAK_F16_MFMA_TFLOPS

        def roof_of(k):
            ms, fl, by, n, fle = fam[k]
            ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            # `achieved` stays ALGORITHMIC (direct-conv FLOPs); `mfma_executed` is what the matrix cores ran
            exe = fle / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            return {"bound": "mfma", "kernel": k, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "mfma_executed": round(exe, 2), "mfma_executed_frac": round(exe / peak, 4),
                    "launches": n, "avg_launch_ms": round(ms / max(n, 1), 4),
                    "alg_gflop_per_launch": round(fl / max(n, 1) / 1e9, 3),
                    "alg_mbytes_per_launch": round(by / max(n, 1) / 1e6, 2),
                    "share_of_gpu_time": round(ms / total_ms, 4)}
        dom = max(fam, key=lambda k: fam[k][0])
        roof = roof_of(dom)
        if dom == 'conv_wino':
            roof["note"] = ("achieved = direct-form (algorithmic) FLOPs / time; the kernel is float32 Winograd F(2x2,3x3), "
                            "which executes 16/36 of them (7x7 layers as nine 3x3 blocks: 144/196), so frac can exceed 1; mfma_executed_frac is the matrix-core "
                            "utilisation against the same dense f32 peak")
        # HBM bytes per launch come from the PMC passes of the SAME command (scripts/gpu_round.sh ... pmc ->
        # scripts/summarize_prof.py -> profiles/<family>_traffic.json); counters cannot be read in-process.
        roof["traffic"] = None
        roof["traffic_unit"] = "HBM bytes per launch (PMC, profiles/%s_traffic.json)" % dom
        tpath = os.path.join(ROOT, 'profiles', '%s_traffic.json' % dom)
        if a.workload == 'full' and a.dtype == 'f32' and (B, H, W) == (32, 320, 320) and os.path.exists(tpath):
            try:
                roof["traffic"] = round(json.load(open(tpath))['hbm_bytes_per_launch'])
            except Exception:
                pass
        others = [roof_of(k) for k in sorted(fam, key=lambda k: -fam[k][0]) if k != dom and k.startswith('conv')]
        if a.layers:
            agg = {}
            for name, kern, ms_, fl_, by_ in rows:
                r = agg.setdefault((name, kern), [0.0, 0.0, 0.0, 0])
                r[0] += ms_; r[1] += fl_; r[2] += by_; r[3] += 1
            print("%-28s %-34s %9s %9s %9s" % ('layer', 'kernel', 'ms/step', 'TFLOP/s', 'GB/s'), file=sys.stderr)
            for (name, kern), (ms_, fl_, by_, n_) in agg.items():
                print("%-28s %-34s %9.3f %9.1f %9.0f" % (name, kern, ms_ / a.steps, fl_ / ms_ / 1e9 if ms_ else 0,
                                                         by_ / ms_ / 1e6 if ms_ else 0), file=sys.stderr)
        cpu = None
        if world == 1 and a.cpu_images > 0:
            cpu = cpu_baseline(weights, H, W, a.cpu_images, a.workload)
        n_img = B * world * a.steps
        fl_img = arch.pipeline_flops(H, W)
        res = {
            "metric": "images/sec full pipeline (HandSegNet+crop+PoseNet2D+PosePrior/Viewpoint, RGB -> 21x3D kpts)"
                      if a.workload == 'full' else "images/sec PoseNet2D only (256x256 crops)",
            "value": round(n_img / dt, 2), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic (seeded images, seeded fan-in-scaled weights)",
            "config": {"workload": ("ColorHandPose3DNetwork.inference, %dx%dx3 f32 in HBM, %d images/GPU/step"
                                    % (H, W, B)) if a.workload == 'full' else
                                   ("inference_pose2d, 256x256x3 f32 in HBM, %d images/GPU/step" % B),
                       "global_batch": B * world, "per_gpu_batch": B, "height": H, "width": W,
                       "parallelism": "batch-shard x%d (no data-path collective; keypoint all_gather per step)%s" % (
                           world, "" if len(engines) == 1 else "; %d HIP streams per GPU" % len(engines)),
                       "alg_gflop_per_image": round((fl_img['total'] if a.workload == 'full' else fl_img['posenet']) / 1e9, 2)},
            "roofline": roof, "roofline_other_conv": others, "cpu_baseline": cpu,
        }
        os.write(json_fd, (json.dumps(res) + '\n').encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

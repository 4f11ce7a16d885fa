"""Experiment: does running the per-GPU batch as S sub-batches on S engine contexts (= S HIP streams) fill the tails of the
persistent Winograd kernels and the launch gaps?   python scripts/exp_streams.py [B H W]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hand3d_amd import Engine, synth

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 320, 320)
w = synth.make_weights()
e0 = Engine(0); e0.load_weight_dict(w); e0.finalize_weights(0)
img = synth.make_batch(1000, B, H, W); hs = synth.hand_sides(B)
for S in (1, 2, 4):
    engs = [e0]
    blob = e0.dev_alloc(e0.blob_bytes()); e0.blob_export(int(blob))
    for _ in range(S - 1):
        e = Engine(0); e.blob_import(int(blob), e0.nets_mask()); engs.append(e)
    blob.free()
    Bs = B // S
    bufs = []
    for i, e in enumerate(engs):
        bufs.append((e.to_device(img[i * Bs:(i + 1) * Bs]), e.to_device(hs[i * Bs:(i + 1) * Bs]), e.dev_alloc(Bs * 63 * 4), e.dev_alloc(Bs * 256 * 256 * 21 * 4)))
    def step():
        for e, (di, dh, dc, dk) in zip(engs, bufs):
            e.infer_full_dev(Bs, H, W, int(di), int(dh), kpmap=int(dk), coord3d=int(dc))
        for e in engs:
            e.sync()
    for _ in range(3): step()
    t0 = time.perf_counter()
    K = 10
    for _ in range(K): step()
    dt = time.perf_counter() - t0
    print("streams %d: %.1f images/s  %.3f ms/step" % (S, B * K / dt, dt / K * 1e3), flush=True)
    for e in engs[1:]: e.close()

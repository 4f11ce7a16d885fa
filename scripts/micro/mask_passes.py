"""mask_grow per-pass cost on the GPU: the same 480 x 640 detection map with seeds whose component converges after 1, 8, 32 and
64 growth passes (horizontal bars of growing length: a pass advances 10 pixels).  Kernel time from wall-clock around the per-op
entry point minus the same call on an empty map (the copies dominate otherwise).

  python scripts/micro/mask_passes.py
"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hand3d_amd import Engine  # noqa: E402

eng = Engine(0)
H, W, B = 480, 640, 32


def run(length, fill=False):
    sm = np.zeros((B, H, W, 2), np.float32)
    sm[..., 0] = 1.0
    if fill:
        sm[:, :, :, 1] = 2.0                 # everything foreground: the worst case of a noisy map
    sm[:, 240, 0:max(length, 1), 1] = 2.0    # a bar from the left edge; the seed is its first pixel (first arg-max)
    sm[:, 240, 0, 1] = 3.0
    eng.mask_from_scoremap(sm)
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        eng.mask_from_scoremap(sm)
        t.append(time.perf_counter() - t0)
    return min(t) * 1e3


base = run(1)
for L in (1, 80, 320, 640):
    print('bar of %3d px (%2d passes to converge): %.3f ms (+%.3f over the 1-pass map)' % (L, max(1, (L + 9) // 10), run(L), run(L) - base))
print('all-foreground map: %.3f ms (+%.3f)' % (run(1, True), run(1, True) - base))

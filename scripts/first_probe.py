"""conv1_1-shaped layers (3x3, 3 -> 64) through the per-op entry point: one process per shape under `rocprofv3 --kernel-trace --stats`
reads conv_first.hip's kernel time as a function of batch and image size (does the 320 x 320 launch lose to its size or to its shape?).
usage: python scripts/first_probe.py B H W [walk]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hand3d_amd._lib import Engine
B, H, W = [int(v) for v in sys.argv[1:4]]
e = Engine(0)
if len(sys.argv) > 4:
    e.set_option('first_walk', sys.argv[4])
rng = np.random.default_rng(0)
x = rng.standard_normal((B, H, W, 3)).astype(np.float32)
w = (rng.standard_normal((3, 3, 3, 64)) / np.sqrt(27)).astype(np.float32)
b = rng.standard_normal(64).astype(np.float32)
for _ in range(5):
    y = e.conv2d(x, w, b, 1, True, False)
print(B, H, W, float(np.abs(y).mean()))

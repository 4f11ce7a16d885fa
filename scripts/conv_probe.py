"""One convolution layer through the per-op entry point, under both Winograd kernels (for rocprofv3 --pmc passes).
usage: python scripts/conv_probe.py B H W Cin Cout pool [k [wino,wino2,wino4,wino7]]"""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hand3d_amd._lib import Engine
B, H, W, Cin, Cout, pool = [int(v) for v in sys.argv[1:7]]
k = int(sys.argv[7]) if len(sys.argv) > 7 else 3
e = Engine(0)
rng = np.random.default_rng(0)
x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
b = rng.standard_normal(Cout).astype(np.float32)
modes = sys.argv[8].split(',') if len(sys.argv) > 8 else ['wino', 'wino2']
for mode in modes:          # wino = conv_wino.hip, wino2 = conv_wino2.hip, wino4 = conv_wino4.hip (F(4x4,3x3)), wino7 = conv_wino7.hip
    e.set_option('wino2', '1' if mode == 'wino2' else '0')
    e.set_option('wino4', '1' if mode == 'wino4' else '0')
    e.set_option('wino7', '1' if mode == 'wino7' else '0')        # (7x7 filters: F(4x4,4x4) over four 4x4-tap blocks, conv_wino7.hip)
    e.set_option('conv_impl', 'winograd')
    for _ in range(3):
        y = e.conv2d(x, w, b, 1, True, bool(pool))
    print(mode, float(np.abs(y).mean()))

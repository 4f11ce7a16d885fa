"""One convolution layer through the per-op entry point, under both Winograd kernels (for rocprofv3 --pmc passes).
usage: python scripts/conv_probe.py B H W Cin Cout pool [k]"""
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
for mode in ('0', '1'):
    e.set_option('wino2', mode)
    e.set_option('conv_impl', 'winograd')
    for _ in range(3):
        y = e.conv2d(x, w, b, 1, True, bool(pool))
    print('wino2 =', mode, float(np.abs(y).mean()))

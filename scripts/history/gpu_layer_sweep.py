"""Debug: every conv layer geometry of the pipeline through hp3d_conv2d vs the oracle (B=1 and B=3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hand3d_amd import Engine, arch
from oracle import tf_ops as T
e = Engine(0)
rng = np.random.default_rng(0)
def run(name, B, H, W, l, pool):
    x = rng.standard_normal((B, H, W, l.cin)).astype(np.float32)
    w = (rng.standard_normal((l.k, l.k, l.cin, l.cout)) / np.sqrt(l.k*l.k*l.cin)).astype(np.float32)
    b = rng.standard_normal(l.cout).astype(np.float32)
    r = T.bias_add(T.conv2d_same(x, w, l.stride, acc=np.float64), b)
    if l.relu: r = T.leaky_relu(r)
    if pool: r = T.max_pool_2x2(r)
    errs = []
    for rep in range(3):
        y = e.conv2d(x, w, b, l.stride, l.relu, pool)
        errs.append(float(np.abs(y - r).max()))
    flag = 'BAD' if max(errs) > 1e-4 else 'ok'
    print('%-4s %-26s B%d %3dx%-3d k%d s%d %3d->%-3d pool%d  err %s' % (flag, name, B, H, W, l.k, l.stride, l.cin, l.cout, pool, ' '.join('%.1e' % v for v in errs)), flush=True)
for B in (1, 3):
    h, w = 64, 64
    for l in arch.posenet2d_layers():
        if l.cin == 3: continue
        pool = l.name in ('conv1_2', 'conv2_2', 'conv3_4')
        run('PoseNet2D/' + l.name, B, h, w, l, pool)
        if pool: h, w = h // 2, w // 2
    h = w = 32
    for l in arch.poseprior_layers() + arch.viewpoint_layers():
        if not isinstance(l, arch.Conv): continue
        if l.name.endswith('_0_1'): h = w = 32
        run(l.scope + '/' + l.name, B, h, w, l, False)
        if l.stride == 2: h, w = h // 2, w // 2

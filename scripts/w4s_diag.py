import sys, os, numpy as np
sys.path.insert(0, '/root/repo')
from hand3d_amd._lib import Engine
from oracle import tf_ops as T
e = Engine(0)
for case in [(2, 16, 32, 64, 128, 0), (1, 16, 16, 16, 64, 0), (1, 32, 32, 256, 256, 0)]:
  for tail in ('1', '0'):
    B, H, W, Cin, Cout, pool = case
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    r = T.leaky_relu((T.conv2d_same(x, w, 1, acc=np.float64) + b).astype(np.float32))
    e.set_option('wino4_tail', tail)
    e.set_option('wino4_split', '1')
    ys = [e.conv2d(x, w, b, 1, True, False) for _ in range(6)]
    e.set_option('wino4_split', '0')
    print(case, 'tail', tail, 'err vs ref', ['%.2e' % np.abs(y - r).max() for y in ys])
    for i in range(1, 6):
        dif = np.argwhere(ys[i] != ys[0])
        if len(dif):
            print('  run', i, 'differs in', len(dif), 'elements; max diff %.3e' % np.abs(ys[i] - ys[0]).max(), 'first', dif[:6].tolist(),
                  'couts', sorted(set(dif[:, 3].tolist()))[:20], 'rows', sorted(set(dif[:, 1].tolist()))[:20], 'cols', sorted(set(dif[:, 2].tolist()))[:20])
    bad = np.argwhere(np.abs(ys[0] - r) > 1e-3)
    if len(bad): print('  WRONG elements in run 0:', len(bad), bad[:8].tolist())

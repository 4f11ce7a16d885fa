"""Per-layer error of conv_wino4s.hip (split bf16 operands) and conv_wino4.hip (float32 operands) against the FLOAT64 direct convolution,
on the 3x3 shapes of tests/test_gpu_parity.py:W4_CASES plus the trunk's own layer shapes -- VERDICT r5's headline rule 1.
Test infrastructure (imports oracle/ for the small shapes; a torch-CPU float64 convolution as the checker of the large ones, which the NumPy
oracle needs minutes for -- test_oracle_ops.py holds the two equal).     usage: python tests/helpers/split_numerics.py [out.md]"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hand3d_amd._lib import Engine
from oracle import tf_ops as T

W4_3x3 = [(2, 16, 32, 64, 128, 0), (1, 8, 8, 64, 64, 1), (1, 7, 9, 128, 64, 0), (1, 17, 21, 32, 64, 0), (1, 30, 40, 512, 512, 0), (1, 32, 32, 256, 256, 0),
          (2, 60, 80, 128, 256, 1), (3, 10, 6, 16, 64, 1), (16, 64, 64, 256, 256, 0), (16, 128, 128, 128, 128, 1), (1, 16, 16, 16, 64, 0), (1, 20, 24, 64, 192, 0),
          (1, 18, 22, 32, 64, 1), (8, 64, 64, 64, 128, 0)]
TRUNK = [(8, 80, 80, 256, 256, 0), (8, 40, 40, 512, 512, 0), (8, 64, 64, 128, 256, 0), (8, 32, 32, 512, 256, 0), (4, 160, 160, 128, 128, 1)]


def ref64(x, w, b, pool):
    if x.size * w.shape[-1] * 9 < 2.5e8:
        r = T.leaky_relu((T.conv2d_same(x, w, 1, acc=np.float64) + b).astype(np.float32))
    else:
        import torch
        torch.set_num_threads(max(1, os.cpu_count() or 1))
        xt = torch.from_numpy(x.astype(np.float64)).permute(0, 3, 1, 2)
        wt = torch.from_numpy(w.astype(np.float64)).permute(3, 2, 0, 1)
        y = torch.nn.functional.conv2d(xt, wt, torch.from_numpy(b.astype(np.float64)), padding=1).permute(0, 2, 3, 1).numpy()
        r = T.leaky_relu(y.astype(np.float32))
    return T.max_pool_2x2(r) if pool else r


def main():
    out_md = sys.argv[1] if len(sys.argv) > 1 else None
    e = Engine(0)
    rows = []
    worst = 0.0
    for case in W4_3x3 + TRUNK:
        B, H, W, Cin, Cout, pool = case
        rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
        x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
        w = (rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
        b = rng.standard_normal(Cout).astype(np.float32)
        t0 = time.time()
        r = ref64(x, w, b, pool)
        errs = {}
        for opt, ctr in (('wino4', 'conv_wino4_launches'), ('wino4_split', 'conv_wino4s_launches')):
            e.set_option(opt, '1')
            try:
                n0 = e.counter(ctr)
                y = e.conv2d(x, w, b, 1, True, bool(pool))
                assert e.counter(ctr) == n0 + 1, (opt, case)
                assert np.array_equal(y, e.conv2d(x, w, b, 1, True, bool(pool))), "not deterministic: %s %s" % (opt, case)
            finally:
                e.set_option(opt, 'auto' if opt == 'wino4' else '0')
            errs[opt] = float(np.abs(y - r).max())
        ratio = errs['wino4_split'] / errs['wino4']
        worst = max(worst, ratio)
        rows.append((case, errs['wino4'], errs['wino4_split'], ratio))
        print('%-34s wino4 %.3e  split %.3e  ratio %.2f   (%.1f s)' % (case, errs['wino4'], errs['wino4_split'], ratio, time.time() - t0), flush=True)
    print('worst ratio split / float32: %.2f' % worst)
    if out_md:
        with open(out_md, 'w') as f:
            f.write('| B, H, W, Cin, Cout, pool | conv_wino4 (float32 operands) | conv_wino4s (bf16 x3, 6 products) | ratio |\n|---|---|---|---|\n')
            for case, a, s, q in rows:
                f.write('| %s | %.3e | %.3e | %.2f |\n' % (', '.join(str(v) for v in case), a, s, q))
            f.write('\nworst ratio: %.2f (max |y - float64 direct convolution| on unit-variance data, bias + leaky-ReLU (+ pool) applied)\n' % worst)
    e.close()


if __name__ == '__main__':
    main()

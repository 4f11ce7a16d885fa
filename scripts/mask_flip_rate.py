"""How often does the kernel choice move the hand mask?  The mask is a THRESHOLD of HandSegNet's score map, so any change of
summation order can flip a pixel whose two logits are equal to rounding.  Counts, over N synthetic 320x320 images, the images whose
mask / crop centre / crop scale differ from the all-direct-kernel run (conv_impl=direct: bit-identical to an fmaf chain) under
(a) the F(2x2,3x3) default, (b) wino4=all (F(4x4,3x3) on both trunks).  usage: python scripts/mask_flip_rate.py [n_batches]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hand3d_amd._lib import Engine
from hand3d_amd import synth

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
e = Engine(0)
e.load_weight_dict(synth.make_weights())
e.finalize_weights()
hs = synth.hand_sides(32)
cfgs = [('direct', {'conv_impl': 'direct', 'wino4': '0'}), ('F(2,3) default', {'conv_impl': 'mfma', 'wino4': '0'}),
        ('wino4=auto (PoseNet2D on F(4,3))', {'conv_impl': 'mfma', 'wino4': 'auto'}), ('wino4=all', {'conv_impl': 'mfma', 'wino4': 'all'})]
ref = None
for name, opts in cfgs:
    for k, v in opts.items():
        e.set_option(k, v)
    outs = []
    for b in range(nb):
        img = synth.make_batch(5000 + 32 * b, 32, 320, 320)
        o = e.infer_full(img, hs, want_mask=True, outputs=('scoremap', 'scale', 'center', 'coord3d'))
        outs.append(o)
    if ref is None:
        ref = outs
        print('%-36s reference' % name)
        continue
    nm = nc = nk = 0
    worst = 0.0
    margins = []
    for o, r in zip(outs, ref):
        dm = (o['mask'] != r['mask']).reshape(32, -1).any(1)
        dc = (o['center'] != r['center']).any(1) | (o['scale'] != r['scale']).any(1)
        dk = np.abs(o['coord3d'] - r['coord3d']).reshape(32, -1).max(1)
        nm += int(dm.sum()); nc += int(dc.sum()); nk += int((dk > 1e-4).sum())
        worst = max(worst, float(dk[~dc].max()) if (~dc).any() else 0.0)
        worst_sm = float(np.abs(o['scoremap'] - r['scoremap']).max())
        margins.append(worst_sm)
    print('%-36s images (of %d) with a different mask: %d, different crop box: %d, 3-D keypoints off by > 1e-4: %d; '
          'worst 3-D keypoint error on same-crop images %.2e; worst score-map difference %.2e' % (name, 32 * nb, nm, nc, nk, worst, max(margins)))

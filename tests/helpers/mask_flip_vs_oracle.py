"""How often does each float32 conv kernel family land on the other side of the hand-mask threshold than the FLOAT64 oracle?

The mask is a threshold of HandSegNet's score map (s1 > s0 per pixel, utils/general.py:233-268), so any float32 summation order can flip
a pixel whose two logits agree to rounding.  For N synthetic 320x320 images this counts, per engine mode -- every layer on the direct
kernel (== an fmaf chain), F(2x2,3x3) Winograd, F(4x4,3x3) Winograd (the default) -- the det pixels / masks / crop boxes that differ
from the oracle's (oracle/nets.py:handsegnet with float64 accumulation, float32 activations between layers: the checker the GPU parity
tests use), and the oracle's own logit margin |s1 - s0| at the flipped pixels (how close to the knife edge they are).
Test infrastructure (imports oracle/).  The oracle's float64 HandSegNet is 2.5 s per image on 8 cores, so its small logit maps can be
made ahead on any CPU box:   python tests/helpers/mask_flip_vs_oracle.py --make-oracle tests/helpers/_cache/flip_oracle.npz [n_images]
then on a GPU box:           python tests/helpers/mask_flip_vs_oracle.py [n_images] [out.md]   (uses the cache when it is there)"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hand3d_amd import synth
from oracle import nets as N
from oracle import general as G
from oracle import tf_ops as T

CH = 16
CACHE = os.path.join(ROOT, 'tests', 'helpers', '_cache', 'flip_oracle.npz')
w = synth.make_weights()
if len(sys.argv) > 1 and sys.argv[1] == '--make-oracle':
    path, n = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 256
    os.makedirs(os.path.dirname(path), exist_ok=True)
    smalls = []
    t0 = time.time()
    for b0 in range(0, n, 4):
        nb = min(4, n - b0)
        small, _ = N.handsegnet(w, synth.make_batch(7000 + (b0 // CH) * CH, CH, 320, 320)[b0 % CH:b0 % CH + nb], acc=np.float64)
        smalls.append(small)
        print('%d / %d oracle images, %.0f s' % (b0 + nb, n, time.time() - t0), flush=True)
    np.savez(path, small=np.concatenate(smalls, 0), seed0=7000, chunk=CH)
    sys.exit(0)
from hand3d_amd._lib import Engine
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 256
out_md = sys.argv[2] if len(sys.argv) > 2 else None
cache = np.load(CACHE)['small'] if os.path.exists(CACHE) else None
if cache is not None:
    assert cache.shape[0] >= n_img, "the oracle cache holds fewer images than asked for"
e = Engine(0)
e.load_weight_dict(w)
e.finalize_weights()
modes = [('direct kernel everywhere (fmaf chain)', {'conv_impl': 'direct', 'wino4': '0', 'wino2': '0', 'wino4_split': '0'}),
         ('F(2x2,3x3) (conv_wino.hip)', {'conv_impl': 'mfma', 'wino4': '0', 'wino2': '0', 'wino4_split': '0'}),
         ('F(4x4,3x3) (conv_wino4.hip, the default)', {'conv_impl': 'mfma', 'wino4': 'all', 'wino2': 'auto', 'wino4_split': '0'}),
         # round 6: the same with the filled Cin >= 128 layers on the bf16 matrix pipe, three bfloat16 pieces per operand (conv_wino4s.hip)
         ('F(4x4,3x3), split bf16x3 operands where eligible (option wino4_split=auto)', {'conv_impl': 'mfma', 'wino4': 'all', 'wino2': 'auto', 'wino4_split': 'auto'})]
stat = {m[0]: dict(img_det=0, img_mask=0, img_box=0, px_det=0, margin=0.0, sm_err=0.0) for m in modes}
t0 = time.time()
e.set_option('streams', '1')
for b0 in range(0, n_img, CH):
    nb = min(CH, n_img - b0)
    img = synth.make_batch(7000 + b0, nb, 320, 320)
    hs = synth.hand_sides(nb)
    if cache is not None:
        sm = T.resize_bilinear_legacy(cache[b0:b0 + nb], 320, 320)          # what handsegnet() returns as its large score map
    else:
        sm = N.handsegnet(w, img, acc=np.float64)[1][-1]
    det_o = sm[..., 1] > sm[..., 0]
    margin = np.abs(sm[..., 1].astype(np.float64) - sm[..., 0].astype(np.float64))
    mask_o = G.single_obj_scoremap(sm, early_exit=True)
    cen_o, _, best_o = G.calc_center_bb(mask_o)
    scale_o = G.scale_from_crop_size(best_o, 256)
    for name, opts in modes:
        for k, v in opts.items():
            e.set_option(k, v)
        o = e.infer_full(img, hs, want_mask=True, outputs=('scoremap', 'scale', 'center'))
        det_g = o['scoremap'][..., 1] > o['scoremap'][..., 0]
        d = det_g != det_o
        s = stat[name]
        s['img_det'] += int(d.reshape(nb, -1).any(1).sum())
        s['px_det'] += int(d.sum())
        s['img_mask'] += int((o['mask'] != mask_o[..., 0]).reshape(nb, -1).any(1).sum())
        s['img_box'] += int(((o['center'] != cen_o).any(1) | (o['scale'] != scale_o).reshape(nb, -1).any(1)).sum())
        if d.any():
            s['margin'] = max(s['margin'], float(margin[d].max()))
        s['sm_err'] = max(s['sm_err'], float(np.abs(o['scoremap'] - sm).max()))
    print('%d / %d images, %.0f s' % (b0 + nb, n_img, time.time() - t0), flush=True)
for k in ('conv_impl', 'wino4', 'wino2', 'streams'):
    e.set_option(k, 'auto' if k != 'conv_impl' else 'mfma')
e.set_option('wino4_split', '0')
lines = ['| engine mode | images with a det pixel != oracle | det pixels != oracle (of %d) | largest oracle margin abs(s1 - s0) at such a pixel | images with another mask | images with another crop box | worst score-map error |' % (n_img * 320 * 320),
         '|---|---|---|---|---|---|---|']
for name, _ in modes:
    s = stat[name]
    lines.append('| %s | %d / %d | %d | %.2e | %d | %d | %.2e |' % (name, s['img_det'], n_img, s['px_det'], s['margin'], s['img_mask'], s['img_box'], s['sm_err']))
txt = '\n'.join(lines)
print(txt)
if out_md:
    open(out_md, 'w').write(txt + '\n')

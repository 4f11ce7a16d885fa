"""Generates tests/golden/*.npz from the oracle (seeded synthetic weights + inputs).

The reference ships no golden tensors and TF 1.3 cannot run here, so the
fixtures are outputs of the restated oracle (float64 accumulation) -- "parity unpinned" w.r.t. TF.
Run from the repo root:  python scripts/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hand3d_amd import synth  # noqa: E402
from oracle import general as G  # noqa: E402
from oracle import nets as N  # noqa: E402
from oracle import tf_ops as T  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def main():
    os.makedirs(OUT, exist_ok=True)
    w = synth.make_weights(seed=42)
    # ---- e2e, BASELINE config 1 shape -------------------------------------------------------
    img = synth.make_batch(0, 1, 240, 320)
    hs = np.array([[1.0, 0.0]], np.float32)
    taps = {}
    o = N.inference(w, img, hs, True, acc=np.float64, taps=taps)
    fg, det = G.fg_and_detmap(o[0])
    np.savez_compressed(
        os.path.join(OUT, 'e2e_240x320_seed0.npz'),
        hand_scoremap_small=taps['HandSegNet/conv6_2'], mask_rows=taps['hand_mask'][0, :, :, 0].sum(1),
        mask_cols=taps['hand_mask'][0, :, :, 0].sum(0), seed=G.find_max_location(fg),
        center=o[3], scale_crop=o[2], image_crop_sub=o[1][:, ::16, ::16, :],
        scoremap32=o[4][0, ::8, ::8, :], keypoint_coord3d=o[5],
        fc_xyz=taps['PosePrior/fc_xyz'], conv7_7=taps['PoseNet2D/conv7_7'])
    # ---- per-op fixtures (small) ---------------------------------------------------------------
    rng = np.random.default_rng(123)
    ops = {}
    for name, (B, H, W_, Cin, Cout, k, s, pool) in {
            'c3': (1, 12, 20, 40, 48, 3, 1, 0), 'c3p': (1, 16, 16, 32, 64, 3, 1, 1), 'c3s2': (1, 12, 12, 21, 32, 3, 2, 0),
            'c7': (1, 10, 12, 35, 32, 7, 1, 0), 'c1': (1, 6, 9, 64, 21, 1, 1, 0)}.items():
        x = rng.standard_normal((B, H, W_, Cin)).astype(np.float32)
        ww = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
        b = rng.standard_normal(Cout).astype(np.float32)
        y = T.bias_add(T.conv2d_same(x, ww, s, acc=np.float64), b)
        if name != 'c1':
            y = T.leaky_relu(y)
        if pool:
            y = T.max_pool_2x2(y)
        ops.update({name + '_x': x, name + '_w': ww, name + '_b': b, name + '_y': y,
                    name + '_meta': np.array([s, pool, int(name != 'c1')])})
    x = rng.standard_normal((1, 5, 7, 2)).astype(np.float32)
    ops['rs_x'], ops['rs_y'] = x, T.resize_bilinear_legacy(x, 40, 56)
    im = rng.uniform(-.5, .5, (2, 40, 56, 3)).astype(np.float32)
    c = np.array([[20, 30], [5, 50]], np.float32)
    sc = np.array([5.0, 0.7], np.float32)
    ops['cr_img'], ops['cr_center'], ops['cr_scale'], ops['cr_y'] = im, c, sc, G.crop_image_from_xy(im, c, 64, sc)
    sm = rng.standard_normal((1, 30, 40, 2)).astype(np.float32)
    sml = T.resize_bilinear_legacy(sm, 240, 320)
    m = G.single_obj_scoremap(sml)
    cen, _, siz = G.calc_center_bb(m)
    ops['mk_sm'] = sm
    ops['mk_mask_packed'] = np.packbits(m[0, :, :, 0].astype(np.uint8))
    ops['mk_center'], ops['mk_size'] = cen, siz
    np.savez_compressed(os.path.join(OUT, 'ops_small.npz'), **ops)
    # ---- lifting head ---------------------------------------------------------------------------
    sm32 = (rng.standard_normal((2, 32, 32, 21)) * 0.3).astype(np.float32)
    hs2 = synth.hand_sides(2)
    rel, can, R = N.pose3d(w, sm32, hs2, acc=np.float64)
    np.savez_compressed(os.path.join(OUT, 'pose3d_seed42.npz'), scoremap32=sm32, hand_side=hs2, rel=rel, can=can, R=R)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    main()

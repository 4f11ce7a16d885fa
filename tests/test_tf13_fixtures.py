"""Oracle (and product host helpers) vs fixtures made with REAL TensorFlow 1.x: tests/golden/tf13_*.npz.

Those files cannot be produced in the build container (no TensorFlow wheel, no network).  The recipe is one command on any
box with TF 1.x (scripts/make_tf_fixtures.py, same dump that writes the ref_*.npz files here); until someone runs it this
module SKIPS LOUDLY -- the parity chain is then: HIP == oracle == reference's Python over restated TF kernels
(tests/test_reference_pin.py, ref_*.npz), with the TF kernels themselves cross-checked only against independent
re-derivations (tests/test_oracle_ops.py).
"""
import os

import numpy as np
import pytest

from hand3d_amd import synth
from oracle import general as G
from oracle import nets as N

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
HAVE = os.path.exists(os.path.join(GOLD, 'tf13_c1_inference.npz'))
pytestmark = pytest.mark.skipif(not HAVE, reason="tests/golden/tf13_*.npz absent: PARITY AGAINST TENSORFLOW'S OWN KERNELS IS "
                                "UNPINNED until scripts/make_tf_fixtures.py is run on a TensorFlow-1.x box")


def test_oracle_vs_tensorflow_full_pipeline(synth_weights):
    g = np.load(os.path.join(GOLD, 'tf13_c1_inference.npz'))
    for s in g['seeds']:
        k = 's%d_' % s
        img = synth.make_batch(int(s), 1, 240, 320)
        taps = {}
        o = N.inference(synth_weights, img, g[k + 'hand_side'], True, taps=taps)
        assert np.array_equal(np.packbits(taps['hand_mask'][0, :, :, 0].astype(np.uint8)), g[k + 'mask_packed'])
        assert np.array_equal(o[3], g[k + 'center']) and np.array_equal(o[2], g[k + 'scale_crop'])
        assert np.abs(o[0][0, ::8, ::8, :] - g[k + 'hand_scoremap_sub']).max() < 1e-3
        assert np.abs(o[1][0, ::8, ::8, :] - g[k + 'image_crop_sub']).max() < 1e-5
        assert np.abs(o[4][0, ::8, ::8, :] - g[k + 'scoremap32']).max() < 1e-3
        assert np.abs(o[4][0, 101:104] - g[k + 'scoremap256_rows']).max() < 1e-3
        assert np.abs(o[5] - g[k + 'keypoint_coord3d']).max() < 1e-4
        assert np.array_equal(G.detect_keypoints(o[4][0]), g[k + 'kp_crop'])


def test_oracle_vs_tensorflow_mask_and_lifting(synth_weights):
    m = np.load(os.path.join(GOLD, 'tf13_mask_cases.npz'))
    tf_center = m['empty_tf_center'][0].tolist()
    rid = {(160.0, 160.0): 'inf', (0.0, 0.0): 'fltmax'}[tuple(tf_center)]
    print("TensorFlow's reduce_min/max identity on an empty tensor behaves as:", rid)
    G.EMPTY_REDUCE = rid
    try:
        for case in synth.MASK_CASES:
            mask = G.single_obj_scoremap(synth.blob_scoremap(case))
            center, _, size = G.calc_center_bb(mask)
            assert np.array_equal(np.packbits(mask[0, :, :, 0].astype(np.uint8)), m[case + '_tf_mask_packed'])
            assert np.array_equal(center, m[case + '_tf_center']) and np.array_equal(size, m[case + '_tf_size'])
    finally:
        G.EMPTY_REDUCE = 'inf'
    p = np.load(os.path.join(GOLD, 'tf13_poseprior_variants.npz'))
    sm, hs = synth.lifting_scoremaps(5, 2), synth.hand_sides(2)
    for v in ('direct', 'bottleneck', 'local', 'local_w_xyz_loss', 'proposed'):
        w = synth.make_weights(bottleneck=True) if v == 'bottleneck' else synth_weights
        rel, c3d, R = N.poseprior_network(w, v, sm, hs)
        assert np.abs(rel - p[v + '_rel']).max() < 1e-4 and np.abs(c3d - p[v + '_coord3d']).max() < 1e-4


def test_product_evalutil_vs_tensorflow_box():
    from hand3d_amd.utils import general as PG
    e = np.load(os.path.join(GOLD, 'tf13_evalutil.npz'))
    u = PG.EvalUtil()
    for gt, vis, pr in zip(e['gt'], e['vis'], e['pred']):
        u.feed(gt, vis, pr)
    mean, median, auc, pck, thr = u.get_measures(0.0, 30.0, 20)
    assert np.isclose(mean, e['mean'], rtol=1e-12) and np.isclose(auc, e['auc'], rtol=1e-12) and np.allclose(pck, e['pck'])

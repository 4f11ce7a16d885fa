"""Oracle vs the committed golden fixtures (tests/golden, scripts/make_golden.py) and the same
fixtures through the kernel sources on the CPU interpreter."""
import os

import numpy as np
import pytest

from hand3d_amd import synth
from oracle import general as G
from oracle import nets as N
from oracle import tf_ops as T

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def ops():
    return np.load(os.path.join(GOLD, 'ops_small.npz'))


def _conv_cases(ops):
    for name in ('c3', 'c3p', 'c3s2', 'c7', 'c1'):
        s, pool, act = [int(v) for v in ops[name + '_meta']]
        yield name, ops[name + '_x'], ops[name + '_w'], ops[name + '_b'], s, pool, act, ops[name + '_y']


def test_oracle_f32_matches_golden_ops(ops):
    for name, x, w, b, s, pool, act, y in _conv_cases(ops):
        r = T.bias_add(T.conv2d_same(x, w, s, acc=np.float32), b)
        if act:
            r = T.leaky_relu(r)
        if pool:
            r = T.max_pool_2x2(r)
        assert np.abs(r - y).max() < 2e-5, name
    assert np.array_equal(T.resize_bilinear_legacy(ops['rs_x'], 40, 56), ops['rs_y'])
    assert np.array_equal(G.crop_image_from_xy(ops['cr_img'], ops['cr_center'], 64, ops['cr_scale']), ops['cr_y'])
    m = G.single_obj_scoremap(T.resize_bilinear_legacy(ops['mk_sm'], 240, 320))
    assert np.array_equal(np.packbits(m[0, :, :, 0].astype(np.uint8)), ops['mk_mask_packed'])


def test_kernels_on_interpreter_match_golden_ops(ops, emu_engine):
    e = emu_engine
    for name, x, w, b, s, pool, act, y in _conv_cases(ops):
        assert np.abs(e.conv2d(x, w, b, s, bool(act), bool(pool)) - y).max() < 2e-5, name
    assert np.array_equal(e.resize_bilinear(ops['rs_x'], 40, 56), ops['rs_y'])
    assert np.array_equal(e.crop_and_resize(ops['cr_img'], ops['cr_center'], ops['cr_scale'], 64), ops['cr_y'])
    large = T.resize_bilinear_legacy(ops['mk_sm'], 240, 320)
    mask, center, size, _, _ = e.mask_from_scoremap(large)
    assert np.array_equal(np.packbits(mask[0].astype(np.uint8)), ops['mk_mask_packed'])
    assert np.array_equal(center, ops['mk_center']) and np.array_equal(size, ops['mk_size'])


def test_pose3d_golden(emu_engine, synth_weights):
    g = np.load(os.path.join(GOLD, 'pose3d_seed42.npz'))
    rel, can, R = N.pose3d(synth_weights, g['scoremap32'], g['hand_side'])
    assert np.abs(rel - g['rel']).max() < 1e-5 and np.abs(R - g['R']).max() < 1e-5
    emu_engine.load_weight_dict({k: v for k, v in synth_weights.items() if k.startswith(('PosePrior', 'ViewpointNet'))})
    emu_engine.finalize_weights()
    rel2, can2, R2 = emu_engine.pose3d(g['scoremap32'], g['hand_side'])
    assert np.abs(rel2 - g['rel']).max() < 1e-5 and np.abs(can2 - g['can']).max() < 1e-5 and np.abs(R2 - g['R']).max() < 1e-5


def test_e2e_golden_oracle_glue(synth_weights):
    """Stages downstream of HandSegNet recomputed by the f32 oracle from the golden small scoremap."""
    g = np.load(os.path.join(GOLD, 'e2e_240x320_seed0.npz'))
    large = T.resize_bilinear_legacy(g['hand_scoremap_small'], 240, 320)
    fg, _ = G.fg_and_detmap(large)
    assert np.array_equal(G.find_max_location(fg), g['seed'])
    m = G.single_obj_scoremap(large, early_exit=True)
    assert np.array_equal(m[0, :, :, 0].sum(1), g['mask_rows']) and np.array_equal(m[0, :, :, 0].sum(0), g['mask_cols'])
    c, _, s = G.calc_center_bb(m)
    assert np.array_equal(c, g['center']) and np.array_equal(G.scale_from_crop_size(s), g['scale_crop'])
    crop = G.crop_image_from_xy(synth.make_batch(0, 1, 240, 320), c, 256, G.scale_from_crop_size(s))
    assert np.array_equal(crop[:, ::16, ::16, :], g['image_crop_sub'])
    rel, _, _ = N.pose3d(synth_weights, g['conv7_7'], np.array([[1.0, 0.0]], np.float32))
    assert np.abs(rel - g['keypoint_coord3d']).max() < 1e-5

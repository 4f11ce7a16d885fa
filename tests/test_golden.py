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


# ---- fixtures written by executing the reference's own code (scripts/make_ref_fixtures.py) -----------------------------
def test_oracle_matches_reference_code_fixtures(synth_weights):
    """oracle == tests/golden/ref_*.npz.  Runs everywhere (the fixtures travel; /root/reference does not)."""
    g = np.load(os.path.join(GOLD, 'ref_c1_inference.npz'))
    for s in (0, 3):
        k = 's%d_' % s
        img = synth.make_batch(s, 1, 240, 320)
        taps = {}
        o = N.inference(synth_weights, img, g[k + 'hand_side'], True, taps=taps)
        assert np.array_equal(np.packbits(taps['hand_mask'][0, :, :, 0].astype(np.uint8)), g[k + 'mask_packed'])
        assert np.array_equal(o[3], g[k + 'center']) and np.array_equal(o[2], g[k + 'scale_crop'])
        assert np.array_equal(o[4][0, ::8, ::8, :], g[k + 'scoremap32'])
        assert np.abs(o[5] - g[k + 'keypoint_coord3d']).max() < 1e-6
        assert np.array_equal(G.detect_keypoints(o[4][0]), g[k + 'kp_crop'])
    # the batch of 8 (ref_c4_b8_inference.npz): the oracle run per image equals the reference's BATCHED run (two images here, for time)
    g8 = np.load(os.path.join(GOLD, 'ref_c4_b8_inference.npz'))
    img8 = synth.make_batch(int(g8['seed0']), 8, int(g8['shape'][0]), int(g8['shape'][1]))
    for i in (1, 6):
        taps = {}
        o = N.inference(synth_weights, img8[i:i + 1], g8['hand_side'][i:i + 1], True, taps=taps)
        assert np.array_equal(np.packbits(taps['hand_mask'][0, :, :, 0].astype(np.uint8)), g8['mask_packed'][i])
        assert np.array_equal(o[3], g8['center'][i:i + 1]) and np.array_equal(o[2], g8['scale_crop'][i:i + 1])
        assert np.array_equal(o[4][0, ::8, ::8, :], g8['scoremap32'][i])
        assert np.abs(o[5][0] - g8['keypoint_coord3d'][i]).max() < 1e-6
        assert np.array_equal(G.detect_keypoints(o[4][0]), g8['kp_crop'][i])
    m = np.load(os.path.join(GOLD, 'ref_mask_cases.npz'))
    for rid in ('inf', 'fltmax'):
        G.EMPTY_REDUCE = rid
        try:
            for case in synth.MASK_CASES:
                mask = G.single_obj_scoremap(synth.blob_scoremap(case))
                center, _, size = G.calc_center_bb(mask)
                key = '%s_%s_' % (case, rid)
                assert np.array_equal(np.packbits(mask[0, :, :, 0].astype(np.uint8)), m[key + 'mask_packed'])
                assert np.array_equal(center, m[key + 'center']) and np.array_equal(size, m[key + 'size'])
        finally:
            G.EMPTY_REDUCE = 'inf'
    p = np.load(os.path.join(GOLD, 'ref_poseprior_variants.npz'))
    sm, hs = synth.lifting_scoremaps(5, 2), synth.hand_sides(2)
    for v in ('direct', 'local', 'proposed'):
        rel, c3d, R = N.poseprior_network(synth_weights, v, sm, hs)
        assert np.abs(rel - p[v + '_rel']).max() < 2e-6 and np.abs(c3d - p[v + '_coord3d']).max() < 1e-6


def test_kernels_on_interpreter_match_reference_code_fixtures(emu_engine, synth_weights):
    """The product's kernels (CPU interpreter build) and host helpers vs the reference-code fixtures."""
    from hand3d_amd.utils import general as PG
    m = np.load(os.path.join(GOLD, 'ref_mask_cases.npz'))
    for case in synth.MASK_CASES:
        mask, center, size, _, _ = emu_engine.mask_from_scoremap(synth.blob_scoremap(case))
        key = '%s_inf_' % case
        assert np.array_equal(np.packbits(mask[0].astype(np.uint8)), m[key + 'mask_packed']), case
        assert np.array_equal(center, m[key + 'center']) and np.array_equal(size, m[key + 'size']), case
    p = np.load(os.path.join(GOLD, 'ref_poseprior_variants.npz'))
    sm, hs = synth.lifting_scoremaps(5, 2), synth.hand_sides(2)
    emu_engine.load_weight_dict({k: v for k, v in synth_weights.items() if k.startswith(('PosePrior', 'ViewpointNet'))})
    emu_engine.finalize_weights()
    for v in ('local', 'proposed'):
        rel, c3d, R = emu_engine.poseprior(v, sm, hs)
        assert np.abs(rel - p[v + '_rel']).max() < 1e-4 and np.abs(c3d - p[v + '_coord3d']).max() < 1e-4
    e = np.load(os.path.join(GOLD, 'ref_evalutil.npz'))
    u = PG.EvalUtil()
    for gt, vis, pr in zip(e['gt'], e['vis'], e['pred']):
        u.feed(gt, vis, pr)
    mean, median, auc, pck, thr = u.get_measures(0.0, 30.0, 20)
    assert np.isclose(mean, e['mean'], rtol=1e-13) and np.isclose(median, e['median'], rtol=1e-13)
    assert np.isclose(auc, e['auc'], rtol=1e-13) and np.allclose(pck, e['pck'], rtol=1e-13) and np.array_equal(thr, e['thresholds'])


def test_c5_fixture_is_the_oracle_at_config5_size_and_precision(synth_weights):
    """tests/golden/c5_f16_480x640.npz (scripts/make_c5_fixture.py) re-derived for its second image: 480x640, trunks rounding to
    half where the engine stores halves (oracle/nets.py f16=True), float64 accumulation -- the fixture the GPU test of BASELINE
    config 5 reads is exactly what the oracle computes (~20 s)."""
    g = np.load(os.path.join(GOLD, 'c5_f16_480x640.npz'))
    i = 1
    img = synth.make_batch(int(g['seed0']) + i, 1, 480, 640)
    hs = synth.hand_sides(2)[i:i + 1]
    small, large = N.handsegnet(synth_weights, img, acc=np.float64, f16=True)
    assert np.array_equal(small, g['seg_small'][i:i + 1])
    fg, det = G.fg_and_detmap(large[-1])
    assert np.array_equal(np.packbits(det.reshape(1, -1).astype(np.uint8), axis=1)[0], g['det'][i])
    margin = np.abs(large[-1][..., 1] - large[-1][..., 0])
    assert np.array_equal(np.minimum(np.floor(margin * 1e4), 255).astype(np.uint8)[0], g['margin_q'][i])
    mask = G.single_obj_scoremap(large[-1], early_exit=True)
    assert np.array_equal(np.packbits(mask.reshape(1, -1).astype(np.uint8), axis=1)[0], g['mask'][i])
    cen, _, best = G.calc_center_bb(mask)
    sc = G.scale_from_crop_size(best, 256)
    assert np.array_equal(cen, g['center'][i:i + 1]) and np.array_equal(sc, g['scale_crop'][i:i + 1])
    sms = N.posenet2d(synth_weights, G.crop_image_from_xy(img, cen, 256, scale=sc), acc=np.float64, f16=True)
    for k in range(3):
        assert np.array_equal(sms[k], g['sm32'][k][i:i + 1])
    coord3d, _, _ = N.pose3d(synth_weights, sms[-1], hs, acc=np.float64)
    assert np.array_equal(coord3d, g['coord3d'][i:i + 1])
    # the half-precision configuration stays inside its own bar against the float32 path on these frames
    assert np.abs(g['sm32'][2] - g['sm32_f32']).max() < 5e-3 and np.abs(g['coord3d'] - g['coord3d_f32']).max() < 5e-3

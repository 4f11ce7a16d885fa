"""GPU parity against fixtures written by EXECUTING THE REFERENCE'S OWN CODE (tests/golden/ref_*.npz, generator
scripts/make_ref_fixtures.py: /root/reference run unmodified over the NumPy TensorFlow stand-in oracle/tfshim).

/root/reference does not exist on the GPU box, so nothing here reads it: only the committed .npz files.  Tolerances are
the north star's: heat-maps 1e-3, 3-D keypoints 1e-4 (max-abs); masks, centres, scales, arg-max keypoints bit-exact -- a mask may differ
from the reference's only through det pixels the reference's own run lists as knife-edge (|p_fg - 1/2| < 4e-6: _mask_gate below), and the
mask stage is always held exactly on the device's own score map.
"""
import os

import numpy as np
import pytest

from hand3d_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')
TOL_HEATMAP, TOL_KP3D = 1e-3, 1e-4


@pytest.fixture(params=['ref_', 'tf13_'])
def gold(request):
    """ref_: the reference's code over the NumPy TF stand-in (always present).  tf13_: the same dump made with real
    TensorFlow 1.x by scripts/make_tf_fixtures.py -- not producible in the build container; skipped LOUDLY until supplied."""
    pre = request.param
    if not os.path.exists(os.path.join(GOLD, pre + 'c1_inference.npz')):
        pytest.skip("tests/golden/%s*.npz absent: run scripts/make_tf_fixtures.py on a box with TensorFlow 1.x "
                    "(inputs: scripts/make_ref_fixtures.py --export-inputs) to pin the TF kernels themselves" % pre)
    return lambda name: np.load(os.path.join(GOLD, pre + name))


def _mask_gate(o, i, mask_packed, scoremap_sub, knife_yx):
    """The hand mask is a THRESHOLD of HandSegNet's score map (utils/general.py:240-245) and the synthetic weights' maps hold pixels whose
    foreground probability sits within float32 rounding of 1/2: which side such a pixel lands on depends on the summation order of the
    convolution kernels, not on their correctness ("Hard parts").  Until round 5 this file demanded bit-equal masks and so
    pinned HandSegNet's kernel PLAN to whatever the fixtures had been checked with.  Now, for image i of engine output `o`:
      * always: the mask STAGE is exact on the device's own score map (seed, growth, box, centre, scale == oracle/general.py on o['scoremap']);
      * the mask equals the reference's -> True (every crop-dependent gate applies to this image);
      * else every det pixel that differs from the reference's det map must be one of the knife-edge pixels the fixture lists
        (|p_fg - 1/2| < 4e-6 in the reference's own run, scripts/make_tf_fixtures.py:knife_edge_pixels) -> False (its crop moved: skip them)."""
    from oracle import general as G
    from oracle import tf_ops as T
    H, W = o['mask'].shape[1:3]
    m = G.single_obj_scoremap(o['scoremap'][i:i + 1], early_exit=True)
    cen, _, best = G.calc_center_bb(m)
    assert np.array_equal(o['mask'][i], m[0, :, :, 0]), "image %d: mask growth differs from the oracle's on the device's own score map" % i
    assert np.array_equal(o['center'][i:i + 1], cen) and np.array_equal(o['scale'][i:i + 1], G.scale_from_crop_size(best, 256)), i
    if np.array_equal(np.packbits(o['mask'][i].astype(np.uint8)), mask_packed):
        return True
    ref_full = T.resize_bilinear_legacy(np.asarray(scoremap_sub)[None], H, W)       # what the reference up-sampled (nets/ColorHandPose3DNetwork.py:166)
    det_e = G.fg_and_detmap(o['scoremap'][i:i + 1])[1][0]
    det_r = G.fg_and_detmap(ref_full)[1][0]
    diff = np.argwhere(det_e != det_r)
    allowed = set((int(y), int(x)) for y, x in knife_yx)
    stray = [tuple(int(v) for v in p) for p in diff if (int(p[0]), int(p[1])) not in allowed]
    assert len(diff) > 0 and not stray, "image %d: hand mask differs and det pixels %s are not knife-edge pixels of the reference run" % (i, stray[:5])
    return False


@pytest.fixture(scope='module')
def net(gpu_engine, synth_weights):
    from hand3d_amd import ColorHandPose3DNetwork
    n = ColorHandPose3DNetwork(engine=gpu_engine)
    n.init_from_dict(synth_weights)
    return n


def test_reference_fixtures_full_pipeline_c1(net, gold):
    """BASELINE config 1: the five 240x320 images through the reference's inference() + run.py's post-processing."""
    from hand3d_amd.utils.general import EvalUtil, detect_keypoints, trafo_coords
    g = gold('c1_inference.npz')
    ev = EvalUtil()
    undecided = []
    for s in g['seeds']:
        k = 's%d_' % s
        img = synth.make_batch(int(s), 1, 240, 320)
        hs = g[k + 'hand_side']
        o = net.engine.infer_full(img, hs, want_mask=True)
        assert np.abs(o['scoremap'][0, ::8, ::8, :] - g[k + 'hand_scoremap_sub']).max() < TOL_HEATMAP
        if not _mask_gate(o, 0, g[k + 'mask_packed'], g[k + 'hand_scoremap_sub'], g[k + 'knife_iyx'][:, 1:]):
            undecided.append(int(s))
            continue
        assert np.array_equal(o['center'], g[k + 'center']) and np.array_equal(o['scale'], g[k + 'scale_crop'])
        assert np.abs(o['crop'][0, ::8, ::8, :] - g[k + 'image_crop_sub']).max() < 1e-5
        assert np.abs(o['kpmap'][0, ::8, ::8, :] - g[k + 'scoremap32']).max() < TOL_HEATMAP
        assert np.abs(o['kpmap'][0, 101:104] - g[k + 'scoremap256_rows']).max() < TOL_HEATMAP
        assert np.abs(o['kpmap'][0].sum(axis=(0, 1), dtype=np.float64) - g[k + 'scoremap256_sum']).max() < 65536 * 1e-5
        err3d = np.abs(o['coord3d'] - g[k + 'keypoint_coord3d']).max()
        print("seed %d: coord3d err %.2e" % (s, err3d))
        assert err3d < TOL_KP3D
        kp = detect_keypoints(np.squeeze(o['kpmap']))
        assert np.array_equal(kp, g[k + 'kp_crop'])
        assert np.array_equal(trafo_coords(kp, o['center'], o['scale'], 256), g[k + 'kp_uv'])
        # the same two host steps evaluated on the device (hp3d_infer_full_kp)
        od = net.engine.infer_full(img, hs, outputs=('kp_crop', 'kp_hw'))
        assert np.array_equal(od['kp_crop'][0], g[k + 'kp_crop']) and np.array_equal(od['kp_hw'][0], g[k + 'kp_uv'])
        ev.feed(g[k + 'keypoint_coord3d'][0], np.ones(21), o['coord3d'][0])
    mean_epe = ev.get_measures(0.0, 0.05, 20)[0]
    print("mean EPE engine vs reference-code fixtures: %.3e; images whose mask moved through a knife-edge pixel: %s" % (mean_epe, undecided))
    assert mean_epe < TOL_KP3D and len(undecided) <= 1


def test_reference_fixtures_batch8_on_the_headline_kernel(net, gold):
    """One batch of 8 at 240x320 through the reference's inference() (BASELINE config 4's per-GPU shard in small; utils/general.py:210
    only needs B < H, W): from this batch size on the engine's default policy runs the 3x3 trunk layers on conv_wino4.hip (and the 7x7 layers on
    conv_wino7.hip's channel-split form), so these fixtures -- written by executing nets/ColorHandPose3DNetwork.py:61-99 -- meet the F(4x4,3x3) kernel directly, not through the
    oracle chain.  Gates as everywhere: mask / centre / scale / arg-max keypoints exact, heat-maps 1e-3, 3-D keypoints 1e-4."""
    from hand3d_amd.utils.general import detect_keypoints, trafo_coords
    if not os.path.exists(os.path.join(GOLD, 'ref_c4_b8_inference.npz')):
        pytest.fail("tests/golden/ref_c4_b8_inference.npz is missing: scripts/make_ref_fixtures.py writes it")
    try:
        g = gold('c4_b8_inference.npz')
    except FileNotFoundError:
        pytest.skip("tf13_c4_b8_inference.npz absent (needs a TensorFlow 1.x box)")
    H, W = [int(v) for v in g['shape']]
    img = synth.make_batch(int(g['seed0']), 8, H, W)
    hs = g['hand_side']
    c0, c7 = net.engine.counter('conv_wino4_launches'), net.engine.counter('conv_wino7_split_launches')
    o = net.engine.infer_full(img, hs, want_mask=True)
    n4, n7 = net.engine.counter('conv_wino4_launches') - c0, net.engine.counter('conv_wino7_split_launches') - c7
    print("batch of 8: %d conv_wino4 launches, %d conv_wino7 launches in the channel-split form" % (n4, n7))
    # (round 5: the ten 7x7 layers of a batch this size run on conv_wino7.hip's channel-split form instead of conv_wino4's nine-block form --
    #  so these fixtures hold that form to the reference as well)
    assert n4 >= 10 and n4 + n7 >= 20, "the F(4x4,3x3) / F(4x4,4x4) kernels did not run: this test would not be holding them to the reference"
    assert np.abs(o['scoremap'][:, ::8, ::8, :] - g['hand_scoremap_sub']).max() < TOL_HEATMAP
    knife = g['knife_iyx']
    ok = [i for i in range(8) if _mask_gate(o, i, g['mask_packed'][i], g['hand_scoremap_sub'][i], knife[knife[:, 0] == i][:, 1:])]
    print("batch of 8: masks equal to the reference's on images %s (the others moved through a knife-edge pixel)" % ok)
    assert len(ok) >= 6
    assert np.array_equal(o['center'][ok], g['center'][ok]) and np.array_equal(o['scale'][ok], g['scale_crop'][ok])
    assert np.abs(o['crop'][ok][:, ::8, ::8, :] - g['image_crop_sub'][ok]).max() < 1e-5
    e_map = np.abs(o['kpmap'][ok][:, ::8, ::8, :] - g['scoremap32'][ok]).max()
    e_3d = np.abs(o['coord3d'][ok] - g['keypoint_coord3d'][ok]).max()
    print("batch of 8 vs reference-code fixtures: heat-maps %.2e, coord3d %.2e" % (e_map, e_3d))
    assert e_map < TOL_HEATMAP and e_3d < TOL_KP3D
    assert np.abs(o['kpmap'][ok].sum(axis=(1, 2), dtype=np.float64) - g['scoremap256_sum'][ok]).max() < 65536 * 1e-5
    od = net.engine.infer_full(img, hs, outputs=('kp_crop', 'kp_hw'))
    for i in ok:
        kp = detect_keypoints(o['kpmap'][i])
        assert np.array_equal(kp, g['kp_crop'][i]) and np.array_equal(od['kp_crop'][i], g['kp_crop'][i])
        assert np.array_equal(trafo_coords(kp, o['center'][i:i + 1], o['scale'][i:i + 1], 256), g['kp_uv'][i])
        assert np.array_equal(od['kp_hw'][i], g['kp_uv'][i])


def test_reference_fixtures_inference2d_c3(net, gold):
    """BASELINE config 3 shape: inference2d on raw 320x320 frames."""
    from hand3d_amd.utils.general import detect_keypoints
    g = gold('c3_inference2d.npz')
    img = synth.make_batch(int(g['seed0']), 2, 320, 320)
    kp, crop, scale, center = net.inference2d(img)
    assert np.array_equal(scale, g['scale_crop']) and np.array_equal(center, g['center'])
    assert np.abs(kp[:, ::8, ::8, :] - g['scoremap32']).max() < TOL_HEATMAP
    assert np.abs(crop[:, ::8, ::8, :] - g['image_crop_sub']).max() < 1e-5
    for i in range(2):
        assert np.array_equal(detect_keypoints(kp[i]), g['kp_crop'][i])


def test_reference_fixtures_mask_cases(gpu_engine, gold):
    """single_obj_scoremap + calc_center_bb of the reference on engineered score maps, both reducer identities."""
    g = gold('mask_cases.npz')
    try:
        for rid in ('inf', 'fltmax'):
            gpu_engine.set_option('empty_reduce', rid)
            for case in synth.MASK_CASES:
                mask, center, size, scale, seed = gpu_engine.mask_from_scoremap(synth.blob_scoremap(case))
                key = '%s_%s_' % (case, rid)
                if key + 'center' not in g:       # real TF: one behaviour for empty reductions (suffix 'tf')
                    key = '%s_tf_' % case
                    if case == 'empty':
                        which = [r for r, c in (('inf', [160.0, 160.0]), ('fltmax', [0.0, 0.0])) if g[key + 'center'][0].tolist() == c]
                        print("TensorFlow's empty reduce_min/max identity:", which)
                        if which != [rid]:
                            continue
                assert np.array_equal(np.packbits(mask[0].astype(np.uint8)), g[key + 'mask_packed']), key
                assert np.array_equal(center, g[key + 'center']) and np.array_equal(size, g[key + 'size']), key
    finally:
        gpu_engine.set_option('empty_reduce', 'inf')


def test_reference_fixtures_poseprior_variants(gpu_engine, synth_weights, gold):
    from hand3d_amd import PosePriorNetwork
    g = gold('poseprior_variants.npz')
    sm, hs = synth.lifting_scoremaps(5, 2), synth.hand_sides(2)
    try:
        for v in ('direct', 'bottleneck', 'local', 'local_w_xyz_loss', 'proposed'):
            w = synth.make_weights(bottleneck=True) if v == 'bottleneck' else synth_weights
            p = PosePriorNetwork(v, engine=gpu_engine)
            p.init_from_dict({k: a for k, a in w.items() if k.startswith(('PosePrior', 'ViewpointNet'))})
            rel, c3d, R = p.inference(sm, hs, True)
            assert np.abs(rel - g[v + '_rel']).max() < TOL_KP3D and np.abs(c3d - g[v + '_coord3d']).max() < TOL_KP3D, v
            assert (R is None) == (v + '_R' not in g)
            if R is not None:
                assert np.abs(R - g[v + '_R']).max() < TOL_KP3D
    finally:
        gpu_engine.load_weight_dict(synth_weights)
        gpu_engine.finalize_weights()

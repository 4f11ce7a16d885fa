"""The product kernel sources + executor, interpreted on the CPU (tests/emu), vs the oracle on
small shapes: tile maps, halo zero-fill, weight packing, concat-channel permutation, masked edge
tiles, stride-2 SAME asymmetry, fused pool, mask growth, lifting head.  Runs without a GPU."""
import os

import numpy as np
import pytest

from hand3d_amd import synth
from oracle import general as G
from oracle import nets as N
from oracle import tf_ops as T

CASES = [(1, 16, 16, 32, 32, 3, 1, 0), (2, 10, 20, 40, 70, 3, 1, 0), (1, 16, 32, 64, 128, 3, 1, 1),
         (1, 12, 12, 21, 32, 3, 2, 0), (1, 9, 9, 33, 64, 3, 2, 0), (1, 8, 8, 32, 21, 1, 1, 0),
         (1, 9, 11, 35, 32, 7, 1, 0), (1, 6, 10, 64, 160, 1, 1, 0), (1, 10, 18, 32, 64, 3, 1, 1)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B%d_%dx%d_%d-%d_k%ds%dp%d" % c)
def test_conv_kernel_on_interpreter(emu_engine, case):
    B, H, W, Cin, Cout, k, s, pool = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    r = T.leaky_relu(T.bias_add(T.conv2d_same(x, w, s, acc=np.float64), b))
    if pool:
        r = T.max_pool_2x2(r)
    assert np.abs(emu_engine.conv2d(x, w, b, s, True, bool(pool)) - r).max() < 1e-5


def test_glue_kernels_on_interpreter(emu_engine):
    e = emu_engine
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 6, 8, 5)).astype(np.float32)
    assert np.array_equal(e.maxpool2(x), T.max_pool_2x2(x))
    x = rng.standard_normal((1, 16, 24, 3)).astype(np.float32)
    assert np.abs(e.avgpool8(x) - T.avg_pool_8x8(x)).max() < 1e-6
    x = rng.standard_normal((2, 5, 7, 3)).astype(np.float32)
    assert np.array_equal(e.resize_bilinear(x, 40, 56), T.resize_bilinear_legacy(x, 40, 56))      # (one workgroup per output row: glue.hip rows kernel)
    x21 = rng.standard_normal((2, 8, 8, 21)).astype(np.float32)                                  # the key-point maps' shape class: 8x, 21 channels
    assert np.array_equal(e.resize_bilinear(x21, 64, 64), T.resize_bilinear_legacy(x21, 64, 64))
    assert np.array_equal(e.resize_bilinear(x, 41, 57), T.resize_bilinear_legacy(x, 41, 57))      # 57 x 3 values per row: the element kernel
    assert np.array_equal(e.resize_bilinear(x, 9, 12), T.resize_bilinear_legacy(x, 9, 12))        # small factor: the element kernel
    img = rng.uniform(-.5, .5, (3, 40, 56, 3)).astype(np.float32)
    c = np.array([[20, 30], [5, 50], [39.5, 2]], np.float32)
    s = np.array([5.0, 1.3, 0.25], np.float32)
    assert np.array_equal(e.crop_and_resize(img, c, s, 64), G.crop_image_from_xy(img, c, 64, s))
    x = rng.standard_normal((5, 300)).astype(np.float32)
    w = (rng.standard_normal((300, 70)) / 17).astype(np.float32)
    b = rng.standard_normal(70).astype(np.float32)
    assert np.abs(e.fc(x, w, b, True) - T.leaky_relu(T.fully_connected(x, w, b, np.float64))).max() < 1e-5
    x = rng.standard_normal((2, 16, 16, 21)).astype(np.float32)
    x[0, 3, 4, 2] = x[0, 9, 9, 2] = 50
    ref = np.array([[np.unravel_index(np.argmax(x[b, :, :, c]), (16, 16)) for c in range(21)] for b in range(2)])
    assert np.array_equal(e.argmax2d(x), ref)


def test_mask_growth_on_interpreter(emu_engine):
    rng = np.random.default_rng(2)
    small = rng.standard_normal((2, 8, 12, 2)).astype(np.float32)
    sm = T.resize_bilinear_legacy(small, 64, 96)
    mask, center, size, scale, seed = emu_engine.mask_from_scoremap(sm)
    rm = G.single_obj_scoremap(sm)[..., 0]
    rc, _, rs = G.calc_center_bb(rm[..., None])
    assert np.array_equal(mask, rm) and np.array_equal(center, rc) and np.array_equal(size, rs)
    assert np.array_equal(seed, G.find_max_location(G.fg_and_detmap(sm)[0]))
    assert np.array_equal(scale, G.scale_from_crop_size(rs))
    # empty mask fallbacks (utils/general.py:311-320) and the 32-pass cap on a 1-px-wide spiral
    empty = np.zeros((1, 40, 64, 2), np.float32)
    empty[..., 1] = -2
    _, c, s_, _, _ = emu_engine.mask_from_scoremap(empty)
    assert c.tolist() == [[160.0, 160.0]] and s_.tolist() == [[100.0]]
    with pytest.raises(AssertionError):
        emu_engine.mask_from_scoremap(np.zeros((5, 4, 9, 2), np.float32))   # reference asserts B < H, W


def test_networks_on_interpreter(emu_engine, synth_weights):
    from hand3d_amd import ColorHandPose3DNetwork
    net = ColorHandPose3DNetwork(engine=emu_engine)
    net.init_from_dict(synth_weights)
    img = synth.make_batch(3, 1, 16, 24)
    large, small = emu_engine.handsegnet(img, want_small=True)
    rs, rl = N.handsegnet(synth_weights, img, acc=np.float64)
    assert np.abs(small - rs).max() < 1e-5 and np.abs(large - rl[0]).max() < 1e-5
    crop = synth.make_batch(9, 1, 16, 16)
    sms = net.inference_pose2d(crop)           # exercises the concat-channel permutation of conv6_1/conv7_1
    ref = N.posenet2d(synth_weights, crop, acc=np.float64)
    for a, b in zip(sms, ref):
        assert a.shape == b.shape and np.abs(a - b).max() < 1e-5
    # the same two nets with every eligible layer forced onto conv_wino.hip: 3x3 layers in both item shapes, the 7x7
    # refinement units as nine 3x3 blocks including the concat-channel permutation of conv6_1 / conv7_1
    emu_engine.set_option('conv_impl', 'winograd')
    try:
        _, small_w = emu_engine.handsegnet(img, want_small=True)
        sms_w = net.inference_pose2d(crop)
    finally:
        emu_engine.set_option('conv_impl', 'mfma')
    assert np.abs(small_w - rs).max() < 1e-5
    for a, b in zip(sms_w, ref):
        assert np.abs(a - b).max() < 1e-5
    # ... and onto conv_wino4.hip (F(4x4,3x3)): the executor's packed 36-plane filters, pooled layers, ragged 4x4 tiles at 16 x 24 and
    # its pooled sizes, the 7x7 units with the concat-channel permutation
    emu_engine.set_option('wino4', '1')
    emu_engine.set_option('wino7', '0')           # (the 7x7 units on the nine-block form here; their F(4x4,4x4) form follows)
    try:
        n0 = emu_engine.counter('conv_wino4_launches')
        _, small_4 = emu_engine.handsegnet(img, want_small=True)
        sms_4 = net.inference_pose2d(crop)
        assert emu_engine.counter('conv_wino4_launches') >= n0 + 30
    finally:
        emu_engine.set_option('wino4', 'auto')
        emu_engine.set_option('wino7', 'auto')
    assert np.abs(small_4 - rs).max() < 3e-5
    for a, b in zip(sms_4, ref):
        assert np.abs(a - b).max() < 3e-5
    # ... and PoseNet2D's 7x7 refinement layers on conv_wino7.hip (F(4x4,4x4) over the filter's four 4x4-tap blocks): the executor's packed
    # filters [chunk][169 products], the concat-channel permutation of conv6_1 / conv7_1 over the 160-channel concat buffer
    emu_engine.set_option('wino7', '1')
    try:
        n0 = emu_engine.counter('conv_wino7_launches')
        sms_7 = net.inference_pose2d(crop)
        assert emu_engine.counter('conv_wino7_launches') == n0 + 10
    finally:
        emu_engine.set_option('wino7', 'auto')
    for a, b in zip(sms_7, ref):
        assert np.abs(a - b).max() < 3e-5
    # ... and the 1x1 head pairs (conv6_1 + conv6_2 of HandSegNet; conv5_1 + conv5_2, conv6_6 + conv6_7, conv7_6 + conv7_7 of PoseNet2D) as one launch
    # each on conv_pw2.hip: a partial 64-pixel tile, the 160-channel concat buffer as input, hidden widths 512 and 128, 2 / 21 real couts
    emu_engine.set_option('pw2', 'force')
    try:
        n0 = emu_engine.counter('conv_pw2_launches')
        _, small_p = emu_engine.handsegnet(img, want_small=True)
        sms_p = net.inference_pose2d(crop)
        assert emu_engine.counter('conv_pw2_launches') == n0 + 4
    finally:
        emu_engine.set_option('pw2', '1')
    assert np.abs(small_p - rs).max() < 1e-5
    for a, b in zip(sms_p, ref):
        assert np.abs(a - b).max() < 1e-5
    rng = np.random.default_rng(5)
    sm32 = (rng.standard_normal((2, 32, 32, 21)) * 0.3).astype(np.float32)
    hs = synth.hand_sides(2)
    rel, can, R = emu_engine.pose3d(sm32, hs)
    rrel, rcan, rR = N.pose3d(synth_weights, sm32, hs, acc=np.float64)
    assert np.abs(rel - rrel).max() < 1e-5 and np.abs(can - rcan).max() < 1e-5 and np.abs(R - rR).max() < 1e-5


def test_lifting_tail_and_stride2_gemm_on_interpreter(emu_engine, synth_weights):
    """Round 6, the unfused lifting stage (batches above 4; forced here with lift_fused = 0): ViewpointNet's three FC layers as fc_partial + ONE
    fc_tail launch (reduction of fc_vp0's K slices, fc_vp1, fc_vp_u with the activations in LDS; glue.hip) and its last stride-2 layer
    conv_vp_2_2 as a split-K GEMM over the 16 output pixels per image (the HWIO filter is the matrix; SAME padding of a stride-2 layer on an even
    map: 0 before, 1 after).  Every combination of the two options against the float64 oracle (nets/ColorHandPose3DNetwork.py:249-309), an odd
    batch (the tail takes two images per workgroup), counters prove which path ran."""
    w = {k: v for k, v in synth_weights.items() if k.startswith(('PosePrior/', 'ViewpointNet/'))}
    emu_engine.load_weight_dict(w)
    emu_engine.finalize_weights()
    rng = np.random.default_rng(11)
    sm32 = (rng.standard_normal((3, 32, 32, 21)) * 0.3).astype(np.float32)
    hs = synth.hand_sides(3)
    ref = N.pose3d(synth_weights, sm32, hs, acc=np.float64)
    emu_engine.set_option('lift_fused', '0')
    outs = {}
    try:
        for tail in ('0', '1'):
            for gemm in ('0', '1'):
                emu_engine.set_option('fc_tail', tail)
                emu_engine.set_option('tiny_gemm', gemm)
                n0, g0 = emu_engine.counter('fc_tail_launches'), emu_engine.counter('conv_s2_gemm_launches')
                o = emu_engine.pose3d(sm32, hs)
                assert emu_engine.counter('fc_tail_launches') - n0 == int(tail) and emu_engine.counter('conv_s2_gemm_launches') - g0 == int(gemm)
                for a, b in zip(o, ref):
                    assert np.abs(a - b).max() < 1e-5, (tail, gemm)
                outs[tail + gemm] = o
    finally:
        emu_engine.set_option('lift_fused', 'auto')
        emu_engine.set_option('fc_tail', '1')
        emu_engine.set_option('tiny_gemm', '1')
        emu_engine.load_weight_dict(synth_weights)
        emu_engine.finalize_weights()
    # PosePrior's tower is untouched by both options: its canonical coordinates are bit-identical across them
    assert all(np.array_equal(outs['00'][1], o[1]) for o in outs.values())


def test_arbitrary_image_sizes_on_interpreter(emu_engine, synth_weights):
    """Input sizes that are not multiples of 8, odd ones included: the VALID 2x2 max-pools floor (utils/general.py:61-65), the legacy resize
    maps the floor(H/8) x floor(W/8) logits back to H x W (nets/ColorHandPose3DNetwork.py:165-166), the mask / crop stage takes the image
    extent as it is -- HandSegNet and the mask / box / crop stage at two such sizes against the oracle (the whole path at a third size: the
    HP3D_SLOW test below and tests/test_gpu_parity.py::test_full_pipeline_arbitrary_image_sizes)."""
    from hand3d_amd import ColorHandPose3DNetwork
    net = ColorHandPose3DNetwork(engine=emu_engine)
    net.init_from_dict(synth_weights)
    for (H, W) in ((37, 45), (50, 68)):
        img = synth.make_batch(3, 1, H, W)
        large, small = emu_engine.handsegnet(img, want_small=True)
        rs, rl = N.handsegnet(synth_weights, img, acc=np.float64)
        assert small.shape == rs.shape == (1, H // 8, W // 8, 2) and large.shape == (1, H, W, 2)
        assert np.abs(small - rs).max() < 1e-5 and np.abs(large - rl[0]).max() < 1e-5
        # mask growth / bounding box / crop on the odd extent: bit-exact like every other size
        mask, center, size, scale, seed = emu_engine.mask_from_scoremap(large)
        rm = G.single_obj_scoremap(large)[..., 0]
        rc, _, rsz = G.calc_center_bb(rm[..., None])
        assert np.array_equal(mask, rm) and np.array_equal(center, rc) and np.array_equal(size, rsz)
        crop = emu_engine.crop_and_resize(img, center, scale, 64)
        assert np.array_equal(crop, G.crop_image_from_xy(img, center, 64, scale))


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get('HP3D_SLOW') != '1', reason="~3 min on the CPU interpreter; set HP3D_SLOW=1")
def test_full_pipeline_on_interpreter(emu_engine, synth_weights):
    from hand3d_amd import ColorHandPose3DNetwork
    net = ColorHandPose3DNetwork(engine=emu_engine)
    net.init_from_dict(synth_weights)
    for (H, W) in ((48, 64), (45, 37)):            # the second: odd extents (any size from 16 x 16 is accepted)
        img = synth.make_batch(1, 1, H, W)
        hs = synth.hand_sides(1)
        out = net.inference(img, hs, True)
        ref = N.inference(synth_weights, img, hs, True, acc=np.float64)
        for a, b in zip(out, ref):
            assert a.shape == b.shape and np.abs(a - b).max() < 1e-4
    img = synth.make_batch(1, 1, 48, 64)
    out = net.inference(img, hs, True)
    # keypoints detected on the device == the reference's host functions on the returned maps
    from hand3d_amd.utils import general as PG
    c3d, kp_hw, kp_crop, scale, center = net.inference_keypoints(img, hs, True)
    assert np.array_equal(c3d, out[5]) and np.array_equal(scale, out[2]) and np.array_equal(center, out[3])
    assert np.array_equal(kp_crop[0], PG.detect_keypoints(out[4][0]))
    assert np.array_equal(kp_hw[0], PG.trafo_coords(PG.detect_keypoints(out[4][0]), out[3], out[2], 256))
    kp_hw2, kp_crop2, scale2, center2 = net.inference2d_keypoints(img)
    assert np.array_equal(kp_hw2, kp_hw) and np.array_equal(kp_crop2, kp_crop)


def test_local_variant_bone_rel_trafo_inv(emu_engine, synth_weights):
    """PosePriorNetwork('local'): PosePrior net + bone_rel_trafo_inv (utils/relative_trafo.py:243-295)."""
    from hand3d_amd import PosePriorNetwork
    from oracle import relative_trafo as RT
    rng = np.random.default_rng(12)
    xyz = rng.standard_normal((3, 21, 3)).astype(np.float32)
    assert np.abs(RT.bone_rel_trafo_inv(RT.bone_rel_trafo(xyz)) - xyz).max() < 1e-5     # round trip pins the restatement
    sm = np.maximum(rng.standard_normal((2, 256, 256, 21)).astype(np.float32), 0) * 0.2
    hs = synth.hand_sides(2)
    net = PosePriorNetwork('local', engine=emu_engine)
    net.init_from_dict({k: v for k, v in synth_weights.items() if k.startswith('PosePrior')})
    rel, c3d, R = net.inference(sm, hs, True)
    rrel, rc3d, rR = N.poseprior_network(synth_weights, 'local', sm, hs, acc=np.float64)
    assert R is None and rR is None
    assert np.abs(c3d - rc3d).max() < 1e-5 and np.abs(rel - rrel).max() < 1e-5


def test_preprocess_u8_on_interpreter(emu_engine):
    """uint8 -> x/255-0.5 -> legacy bilinear resize, bit-exact vs the oracle."""
    rng = np.random.default_rng(21)
    u8 = rng.integers(0, 256, size=(2, 48, 64, 3), dtype=np.uint8)
    assert np.array_equal(emu_engine.preprocess_u8(u8, 24, 32), G.preprocess_u8(u8, 24, 32))
    assert np.array_equal(emu_engine.preprocess_u8(u8, 48, 64), u8.astype(np.float32) / np.float32(255) - np.float32(0.5))
    assert np.array_equal(emu_engine.preprocess_u8(u8, 30, 50), G.preprocess_u8(u8, 30, 50))


def test_f16_trunk_mode_on_interpreter(emu_engine, synth_weights):
    """hp3d_finalize_weights(dtype=1): half-precision trunks (same kernel template, K=16 MFMA) vs the oracle with
    the same rounding points; f32 heads; weights restored to f32 afterwards."""
    from hand3d_amd import ColorHandPose3DNetwork
    net = ColorHandPose3DNetwork(engine=emu_engine)
    net.init_from_dict(synth_weights, dtype='f16')
    try:
        assert emu_engine.nets_mask() & 32
        img = synth.make_batch(3, 1, 16, 24)
        _, small = emu_engine.handsegnet(img, want_small=True)
        rs, _ = N.handsegnet(synth_weights, img, acc=np.float64, f16=True)
        assert np.abs(small - rs).max() < 2e-3
        crop = synth.make_batch(9, 1, 16, 16)
        for a, b in zip(net.inference_pose2d(crop), N.posenet2d(synth_weights, crop, acc=np.float64, f16=True)):
            assert np.abs(a - b).max() < 2e-3
    finally:
        net.init_from_dict(synth_weights, dtype=0)
    assert not (emu_engine.nets_mask() & 32)


def test_f16_trunks_on_conv_h16_kernel_on_interpreter(emu_engine, synth_weights):
    """conv_h16.hip (option f16_impl=h16_force: taken whenever the shape allows, also below the grid-fill threshold):
    HandSegNet's 3x3 trunk incl. the pooled layers, ragged 16 x 16 tiles (24 x 40 and 16 x 72 images, and their pooled sizes) and all three cout-block
    widths (64 / 128 / 256+), vs the oracle with the same rounding points AND vs the general f16 kernel on the same input
    (same MFMA, same packed weights: differences are accumulation-order only)."""
    from hand3d_amd import ColorHandPose3DNetwork
    net = ColorHandPose3DNetwork(engine=emu_engine)
    net.init_from_dict(synth_weights, dtype='f16')
    try:
        for (h, w) in ((24, 40), (16, 72)):
            img = synth.make_batch(5, 1, h, w)
            emu_engine.set_option('f16_impl', 'mfma')
            _, small_ref = emu_engine.handsegnet(img, want_small=True)
            emu_engine.set_option('f16_impl', 'h16_force')
            _, small = emu_engine.handsegnet(img, want_small=True)
            rs, _ = N.handsegnet(synth_weights, img, acc=np.float64, f16=True)
            assert np.abs(small - rs).max() < 2e-3
            assert np.abs(small - small_ref).max() < 5e-4
            # conv1_1 computed inside conv1_2's patch stage (the default) == the two-launch form, bit for bit
            emu_engine.set_option('f16_fuse12', '0')
            _, small_unfused = emu_engine.handsegnet(img, want_small=True)
            emu_engine.set_option('f16_fuse12', '1')
            assert np.array_equal(small, small_unfused)
        # the fused block's two forms (round 6): filter ring, two workgroups per CU / conv1_2's filters resident in registers, one workgroup per
        # CU walking its items with the next patch built between the K-steps -- 3 x 4 tiles on the interpreter's 3 CUs = 4 items per
        # workgroup (prologue, steady state, the dry run past the last item); border tiles (zero padding inside the halo) and two interior
        # ones, ragged bottom row and right column (40 = 2.5, 56 = 3.5 tiles)
        img = synth.make_batch(6, 1, 40, 56)
        outs = {}
        for form in ('ring', 'resident'):
            emu_engine.set_option('f16_fuse12', form)
            n0 = emu_engine.counter('conv_h16_first_resident_launches')
            _, outs[form] = emu_engine.handsegnet(img, want_small=True)
            assert emu_engine.counter('conv_h16_first_resident_launches') - n0 == (1 if form == 'resident' else 0)
        emu_engine.set_option('f16_fuse12', '0')
        _, unfused = emu_engine.handsegnet(img, want_small=True)
        assert np.array_equal(outs['ring'], unfused) and np.array_equal(outs['resident'], unfused)
        emu_engine.set_option('f16_fuse12', '1')
        crop = synth.make_batch(9, 1, 16, 16)
        for a, b in zip(net.inference_pose2d(crop), N.posenet2d(synth_weights, crop, acc=np.float64, f16=True)):
            assert np.abs(a - b).max() < 2e-3
    finally:
        emu_engine.set_option('f16_impl', 'h16')
        emu_engine.set_option('f16_fuse12', '1')
        net.init_from_dict(synth_weights, dtype=0)


def test_f16_7x7_and_1x1_layers_on_conv_h16_kernel_on_interpreter(emu_engine, synth_weights):
    """conv_h16.hip's 7x7 / 1x1 forms (option f16_k7k1, round 4): PoseNet2D's score-map stages (Mconv1_stage2: 149 -> 128 over the concat
    buffer, Mconv2..5: 128 -> 128, nets/ColorHandPose3DNetwork.py:206-215) and the 1x1 layers with >= 64 couts (conv6_1_CPM, Mconv6), on a
    crop whose score maps are 17 x 3 (a second, ragged 16 x 16 tile below the first -- the GPU test has 2 x 2 tiles --, the 7x7 halo crossing both the image border and the tile border), vs the
    same layers on the general f16 kernel (same MFMA, same packed weights: accumulation order only) and vs the oracle with the same
    rounding points; the launch counter shows the layers really moved."""
    from hand3d_amd import ColorHandPose3DNetwork
    net = ColorHandPose3DNetwork(engine=emu_engine)
    net.init_from_dict(synth_weights, dtype='f16')
    try:
        emu_engine.set_option('f16_impl', 'h16_force')
        for (h, w) in ((136, 24),):
            crop = synth.make_batch(11, 1, h, w)
            emu_engine.set_option('f16_k7k1', '0')
            n0 = emu_engine.counter('conv_h16_launches')
            ref = net.inference_pose2d(crop)
            n1 = emu_engine.counter('conv_h16_launches')
            emu_engine.set_option('f16_k7k1', '1')
            got = net.inference_pose2d(crop)
            n2 = emu_engine.counter('conv_h16_launches')
            # + conv6_1_CPM (1x1), 2 stages x (5 x 7x7 + Mconv6 1x1)
            assert (n2 - n1) - (n1 - n0) == 1 + 2 * 6
            orc = N.posenet2d(synth_weights, crop, acc=np.float64, f16=True)
            for a, b, c in zip(got, ref, orc):
                assert a.shape == (1, h // 8, w // 8, 21)
                assert np.abs(a - b).max() < 5e-4
                assert np.abs(a - c).max() < 2e-3
    finally:
        emu_engine.set_option('f16_impl', 'h16')
        emu_engine.set_option('f16_k7k1', '1')
        net.init_from_dict(synth_weights, dtype=0)


@pytest.mark.parametrize("case", [(2, 16, 32, 64, 128, 0), (1, 17, 21, 128, 256, 0), (2, 14, 20, 64, 128, 1),
                                  # 64-tile x 64-cout items (16-channel steps, swizzled V): conv1_2-like, odd sizes, 3 cout blocks
                                  (2, 16, 32, 64, 64, 1), (1, 17, 21, 32, 64, 0), (1, 12, 20, 96, 192, 0)],
                         ids=lambda c: "B%d_%dx%d_%d-%d_p%d" % c)
def test_winograd_kernel_on_interpreter(emu_engine, case):
    """conv_wino.hip forced on (F(2x2,3x3)): input/weight/output transforms, fused pool, masked tiles."""
    B, H, W, Cin, Cout, pool = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    r = T.leaky_relu(T.bias_add(T.conv2d_same(x, w, 1, acc=np.float64), b))
    if pool:
        r = T.max_pool_2x2(r)
    emu_engine.set_option('conv_impl', 'winograd')
    try:
        emu_engine.set_option('wino_splitk', '0')
        y = emu_engine.conv2d(x, w, b, 1, True, bool(pool))
        emu_engine.set_option('wino_splitk', '1')       # small shapes under-fill the chip: channel steps split; a pooled layer pools in the reduce
        ys = emu_engine.conv2d(x, w, b, 1, True, bool(pool))
    finally:
        emu_engine.set_option('conv_impl', 'mfma')
        emu_engine.set_option('wino_splitk', '1')
    assert np.abs(y - r).max() < 1e-5 and np.abs(ys - r).max() < 1e-5


@pytest.mark.parametrize("case", [(2, 16, 32, 64, 128, 0), (3, 18, 22, 48, 128, 1), (1, 30, 40, 256, 128, 0), (4, 13, 11, 16, 64, 0), (2, 32, 32, 128, 64, 1),
                                  (1, 7, 9, 128, 64, 0), (3, 10, 6, 16, 64, 1)], ids=lambda c: "B%d_%dx%d_%d-%d_p%d" % c)
def test_winograd_f4x4_split_operands_on_interpreter(emu_engine, case):
    """conv_wino4s.hip (option wino4_split = 1): F(4x4,3x3) with the plane products on v_mfma_f32_16x16x32_bf16 over three bfloat16 pieces per
    operand.  Held on the interpreter: the pre-split filter layout ([U1|U0] + U2 fragments), the plane-per-wave accumulators and their
    exchange through LDS in the epilogue (swizzled [plane][cout][tile]), pooled / ragged / multi-image tiles, items that run as tail pieces;
    the result against the float64 oracle within conv_wino4's gate and close to conv_wino4's own (the interpreter sums an MFMA's 32 products
    wide and rounds once: the hardware's order is its own, the GPU test holds the gate there)."""
    B, H, W, Cin, Cout, pool = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    r = T.leaky_relu(T.bias_add(T.conv2d_same(x, w, 1, acc=np.float64), b))
    if pool:
        r = T.max_pool_2x2(r)
    emu_engine.set_option('wino4', '1')
    try:
        y4 = emu_engine.conv2d(x, w, b, 1, True, bool(pool))
    finally:
        emu_engine.set_option('wino4', 'auto')
    emu_engine.set_option('wino4_split', '1')
    try:
        n0, t0 = emu_engine.counter('conv_wino4s_launches'), emu_engine.counter('conv_wino4s_tail_launches')
        y = emu_engine.conv2d(x, w, b, 1, True, bool(pool))
        assert emu_engine.counter('conv_wino4s_launches') == n0 + 1
        tails = emu_engine.counter('conv_wino4s_tail_launches') - t0
        emu_engine.set_option('wino4_tail', '0')
        y_nt = emu_engine.conv2d(x, w, b, 1, True, bool(pool))
    finally:
        emu_engine.set_option('wino4_split', '0')
        emu_engine.set_option('wino4_tail', '1')
    err, err4 = np.abs(y - r).max(), np.abs(y4 - r).max()
    print('conv_wino4s %s: %.2e (conv_wino4 %.2e), tail pieces %d' % (case, err, err4, tails))
    assert y.shape == r.shape and err < 2e-4 and np.abs(y - y4).max() < 1e-4
    assert np.abs(y_nt - r).max() < 2e-4          # whole items only: the same sums in another order
    if case in ((2, 16, 32, 64, 128, 0), (2, 32, 32, 128, 64, 1), (1, 7, 9, 128, 64, 0)):
        assert tails == 1, "this shape leaves an under-filled last round on the interpreter's 3 CUs: it must run as tail pieces"


@pytest.mark.parametrize("case", [(1, 12, 14, 32, 128, 0), (2, 9, 11, 40, 64, 1)], ids=lambda c: "B%d_%dx%d_%d-%d_a%d" % c)
def test_winograd_7x7_as_3x3_blocks_on_interpreter(emu_engine, case):
    """7x7 filter on conv_wino.hip: nine 3x3 blocks of the zero-extended 9x9 filter accumulate into the same planes
    (shifted windows, padding 3, both item shapes)."""
    B, H, W, Cin, Cout, act = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((7, 7, Cin, Cout)) / np.sqrt(49 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    r = T.bias_add(T.conv2d_same(x, w, 1, acc=np.float64), b)
    if act:
        r = T.leaky_relu(r)
    emu_engine.set_option('conv_impl', 'winograd')
    try:
        emu_engine.set_option('wino_splitk', '0')
        y = emu_engine.conv2d(x, w, b, 1, bool(act), False)
        emu_engine.set_option('wino_splitk', '1')       # 9 x Cin/32 (or /16) steps split over up to 16 workgroups
        ys = emu_engine.conv2d(x, w, b, 1, bool(act), False)
    finally:
        emu_engine.set_option('conv_impl', 'mfma')
        emu_engine.set_option('wino_splitk', '1')
    assert np.abs(y - r).max() < 1e-5 and np.abs(ys - r).max() < 1e-5
    assert not np.array_equal(y, ys), "the split-K variant did not run (different summation order expected)"


@pytest.mark.parametrize("case", [(1, 16, 16, 32, 64, 1), (2, 9, 11, 48, 64, 0), (1, 20, 24, 16, 128, 1), (5, 16, 16, 32, 64, 1), (1, 32, 32, 160, 128, 1),
                                  (1, 13, 18, 32, 192, 1), (9, 8, 8, 16, 64, 1)], ids=lambda c: "B%d_%dx%d_%d-%d_a%d" % c)
def test_winograd_f4x4_4x4_for_7x7_filters_on_interpreter(emu_engine, case):
    """conv_wino7.hip (round 5): a 7x7 filter as the four 4x4-tap blocks of its zero-extended 8x8 form, Winograd F(4x4,4x4) each over the points
    {0, +-1, +-2, 1/2, inf} -- 49 planes, the structurally zero planes of the edge blocks left out at compile time, the transformed input of a
    4x4 tile block + halo (25 windows) shared by the four blocks, work item = 16 tiles x 64 couts.  One tile block with ragged edges, blocks
    at the image border (zero padding 3), several images per launch and more items than the interpreter's CUs (the persistent loop), Cin up
    to the 160-channel concat buffer, three cout blocks.  Against the float64 oracle, and of the same order as the nine-block
    F(4x4,3x3) form of conv_wino4.hip on the same input."""
    B, H, W, Cin, Cout, act = case
    rng = np.random.default_rng(sum(case) + 7)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((7, 7, Cin, Cout)) / np.sqrt(49 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    r = T.bias_add(T.conv2d_same(x, w, 1, acc=np.float64), b)
    if act:
        r = T.leaky_relu(r)
    emu_engine.set_option('wino7', '1')
    try:
        n0, o0 = emu_engine.counter('conv_wino7_launches'), emu_engine.counter('emu_soff_overreads')
        y = emu_engine.conv2d(x, w, b, 1, bool(act), False)
        assert emu_engine.counter('conv_wino7_launches') == n0 + 1
        assert np.array_equal(y, emu_engine.conv2d(x, w, b, 1, bool(act), False)), "not deterministic"
        #  the weight ring runs 13 fragments past an item's last chunk through the SCALAR offset, which the hardware's range check
        # does not cover: the packed filters carry that much slack (wino7_packed_floats), so no 16-byte load leaves its buffer -- the
        # per-op entry point allocates exactly the packed size, where the GPU could have faulted
        assert emu_engine.counter('emu_soff_overreads') == o0, "a weight fetch left the filter buffer through the scalar offset"
    finally:
        emu_engine.set_option('wino7', 'auto')
    emu_engine.set_option('wino4', '1')
    emu_engine.set_option('wino_splitk', '0')
    try:
        y9 = emu_engine.conv2d(x, w, b, 1, bool(act), False)
    finally:
        emu_engine.set_option('wino4', 'auto')
        emu_engine.set_option('wino_splitk', '1')
    e7, e9 = np.abs(y - r).max(), np.abs(y9 - r).max()
    assert y.shape == r.shape and e7 < 1e-4 and e7 < 3 * e9 + 2e-5, (e7, e9)


@pytest.mark.parametrize("case", [(1, 16, 16, 48, 64, 1, 0), (1, 11, 14, 160, 64, 0, 3), (2, 20, 9, 64, 128, 1, 2), (3, 16, 16, 32, 64, 1, 7)],
                         ids=lambda c: "B%d_%dx%d_%d-%d_a%d_ks%d" % c)
def test_winograd_f4x4_4x4_channel_split_on_interpreter(emu_engine, case):
    """conv_wino7.hip, SPLITK form (round 5, small batches: a B = 1 PoseNet2D 7x7 layer is 8 work items on 256 CUs): an item owns a contiguous
    range of the 16-channel chunks and stores the RAW 4x4 sums of its range into [ksplit][B*Ho*Wo][Cout]; conv_splitk_reduce adds the slices
    in order + bias + leaky-ReLU.  The automatic split of a one-item launch (3 interpreter CUs: 3 splits of 3 chunks), forced splits that cut
    10 chunks unevenly (3+3+4), two cout blocks x two tile blocks, and a forced split larger than the number of chunks (clamped to 2).
    Against the float64 oracle and the unsplit launch (another summation order: close, not equal); deterministic; the counter proves it ran."""
    B, H, W, Cin, Cout, act, ks = case
    rng = np.random.default_rng(sum(case) + 71)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((7, 7, Cin, Cout)) / np.sqrt(49 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    r = T.bias_add(T.conv2d_same(x, w, 1, acc=np.float64), b)
    if act:
        r = T.leaky_relu(r)
    emu_engine.set_option('wino7', '1')
    try:
        emu_engine.set_option('wino_splitk', '0')
        y1 = emu_engine.conv2d(x, w, b, 1, bool(act), False)
        emu_engine.set_option('wino_splitk', '1')
        emu_engine.set_option('wino7_ksplit', str(ks) if ks else 'auto')
        n0 = emu_engine.counter('conv_wino7_split_launches')
        y = emu_engine.conv2d(x, w, b, 1, bool(act), False)
        assert emu_engine.counter('conv_wino7_split_launches') == n0 + 1
        assert np.array_equal(y, emu_engine.conv2d(x, w, b, 1, bool(act), False)), "not deterministic"
    finally:
        emu_engine.set_option('wino7', 'auto')
        emu_engine.set_option('wino7_ksplit', 'auto')
        emu_engine.set_option('wino_splitk', '1')
    e, e1 = np.abs(y - r).max(), np.abs(y1 - r).max()
    assert y.shape == r.shape and e < 1e-4 and e < 3 * e1 + 2e-5, (e, e1)
    assert np.abs(y - y1).max() < 1e-4 and not np.array_equal(y, y1)


@pytest.mark.parametrize("case", [(3, 24, 40), (1, 16, 16), (2, 19, 37), (7, 8, 16), (1, 40, 100)], ids=lambda c: "B%d_%dx%d" % c)
def test_first_layer_kernel_balanced_tile_runs_on_interpreter(emu_engine, case):
    """conv_first.hip (conv1_1: 3x3, 3 -> 64, nets/ColorHandPose3DNetwork.py:144,183): a workgroup walks a RUN of consecutive 8 x 16 tiles in
    (image, tile row, tile) order, all runs within one tile of the same length, one workgroup per resident slot (round 5; rounds 2-4 walked
    whole tile rows).  Runs that continue into the next tile row and the next image, run lengths that differ by one, ragged right / bottom
    tiles, fewer tiles than slots; bit-identical to the row walk (the same arithmetic per pixel), and against the float64 oracle."""
    B, H, W = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, H, W, 3)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, 64)) / np.sqrt(27)).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    r = T.leaky_relu(T.bias_add(T.conv2d_same(x, w, 1, acc=np.float64), b))
    n0 = emu_engine.counter('conv_first_launches')
    y = emu_engine.conv2d(x, w, b, 1, True, False)
    assert emu_engine.counter('conv_first_launches') == n0 + 1
    emu_engine.set_option('first_walk', 'rows')
    try:
        y_rows = emu_engine.conv2d(x, w, b, 1, True, False)
    finally:
        emu_engine.set_option('first_walk', 'balanced')
    assert y.shape == r.shape and np.abs(y - r).max() < 1e-5
    assert np.array_equal(y, y_rows)


def _near_tie_scoremaps(trial, rng):
    """[2,32,32,21] score maps whose peak has a neighbour 1 ulp below it (an EARLIER interpolated position of the x8
    up-sampled map can then round up to the peak value) or an exact earlier tie."""
    sm = rng.standard_normal((2, 32, 32, 21)).astype(np.float32)
    if trial == 0:
        return sm
    for b in range(2):
        for c in range(21):
            i, j = int(rng.integers(1, 31)), int(rng.integers(1, 31))
            peak = np.float32(3.0 + c)
            sm[b, i, j, c] = peak
            below = np.nextafter(peak, np.float32(0))
            if trial == 1:
                sm[b, i, j - 1, c] = below
            elif trial == 2:
                sm[b, i - 1, j, c] = below
            else:
                sm[b, i - 1, j - 1, c] = peak
    return sm


def test_device_detect_keypoints_equals_reference_on_upsampled_map(emu_engine):
    """hp3d_detect_keypoints == detect_keypoints(resize_images(map, (256,256))) (utils/general.py:331-344 after
    CHP3D.py:97) without the large map -- including the cases where 8 x argmax(small map) is NOT the answer."""
    from hand3d_amd.utils import general as PG
    rng = np.random.default_rng(0)
    shortcut_wrong = 0
    for trial in range(4):
        sm = _near_tie_scoremaps(trial, rng)
        up = T.resize_bilinear_legacy(sm, 256, 256)
        ref = np.stack([PG.detect_keypoints(up[b]) for b in range(2)])
        got = emu_engine.detect_keypoints(sm)
        assert got.dtype == np.int32 and np.array_equal(ref, got), trial
        shortcut_wrong += int((np.stack([PG.detect_keypoints(sm[b]) for b in range(2)]) * 8 != ref).any(axis=2).sum())
    assert shortcut_wrong > 0, "the engineered cases no longer exercise the rounding hazard"
    # a non-square, non-x8 geometry
    sm = rng.standard_normal((1, 30, 40, 5)).astype(np.float32)
    up = T.resize_bilinear_legacy(sm, 100, 90)
    assert np.array_equal(emu_engine.detect_keypoints(sm, (100, 90))[0], PG.detect_keypoints(up[0]))


def test_argmax_ordering_of_signed_zeros_and_nans_is_numpys(emu_engine):
    """np.argmax treats -0.0 and +0.0 as equal (the first one wins) and returns the first NaN; the device keys do too
    (the raw bit pattern ordered -0.0 below +0.0 and a sign-bit NaN lowest)."""
    from hand3d_amd.utils import general as PG
    x = np.full((1, 8, 8, 4), -1.0, np.float32)
    x[0, :, :, 0] = 0.0
    x[0, 2, 3, 0] = -0.0          # an all-zero map with mixed signs: index 0 wins
    x[0, 0, 0, 1] = -0.0
    x[0, 5, 5, 1] = 0.0           # -0.0 at (0,0) comes first and is not smaller
    x[0, 3, 1, 2] = 7.0
    x[0, 4, 4, 2] = np.float32(np.nan)                      # a NaN beats the finite maximum
    x[0, 6, 6, 3] = -np.float32(np.nan)                     # ... with either sign
    x[0, 1, 1, 3] = np.inf
    got = emu_engine.argmax2d(x)
    for c in range(4):
        v, u = np.unravel_index(np.argmax(x[0, :, :, c]), (8, 8))
        assert tuple(got[0, c]) == (v, u), c
    # the same through the fused up-sample + arg-max: zeros of both signs interpolate to zeros of both signs
    z = np.zeros((1, 4, 4, 2), np.float32)
    z[0, 1:, :, 0] = -0.0
    z[0, 2, 2, 1] = -0.0
    up = T.resize_bilinear_legacy(z, 32, 32)
    assert np.array_equal(emu_engine.detect_keypoints(z, (32, 32))[0], PG.detect_keypoints(up[0]))


@pytest.mark.parametrize("case", [(2, 16, 32, 64, 128, 0, 3), (1, 8, 8, 64, 64, 1, 3), (1, 7, 9, 128, 64, 0, 3), (1, 8, 12, 96, 128, 1, 3),
                                  (1, 17, 21, 32, 64, 0, 3), (1, 12, 14, 32, 128, 0, 7), (2, 9, 11, 40, 64, 0, 7)],
                         ids=lambda c: "B%d_%dx%d_%d-%d_p%d_k%d" % c)
def test_winograd_two_workgroups_per_cu_kernel_on_interpreter(emu_engine, case):
    """conv_wino2.hip (option wino2 = 1) on the interpreter: v_mfma_f32_16x16x4_f32 lane maps, the swizzled V rows, pair loader,
    weight packing [plane][step][Cout/16][q][n][e], 3x3 and 7x7 (nine blocks, zero planes skipped), fused pool, ragged tile
    grids, and the channel split (the interpreter's "chip" has 6 slots, so the small cases split)."""
    B, H, W, Cin, Cout, pool, k = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    r = T.leaky_relu(T.bias_add(T.conv2d_same(x, w, 1, acc=np.float64), b))
    if pool:
        r = T.max_pool_2x2(r)
    emu_engine.set_option('wino2', '1')
    try:
        for sk in ('0', '1'):
            emu_engine.set_option('wino_splitk', sk)
            n0 = emu_engine.counter('conv_wino2_launches')
            y = emu_engine.conv2d(x, w, b, 1, True, bool(pool))
            assert emu_engine.counter('conv_wino2_launches') == n0 + 1
            assert np.abs(y - r).max() < 1e-5, sk
    finally:
        emu_engine.set_option('wino2', 'auto')
        emu_engine.set_option('wino_splitk', '1')


@pytest.mark.parametrize("case", [(2, 16, 32, 64, 128, 0, 3), (1, 8, 8, 64, 64, 1, 3), (1, 7, 9, 128, 64, 0, 3), (1, 8, 12, 96, 128, 1, 3),
                                  (1, 17, 21, 32, 64, 0, 3), (1, 12, 14, 32, 128, 0, 7), (2, 9, 11, 48, 64, 0, 7), (3, 10, 6, 16, 64, 1, 3),
                                  # 8 tile blocks: the XCD-affine item order; one 16-channel step per item; three cout blocks; a pooled layer
                                  # whose pooled extent is odd; a 7x7 layer whose channel split cuts through the nine blocks
                                  (4, 32, 32, 32, 128, 0, 3), (1, 16, 16, 16, 64, 0, 3), (1, 20, 24, 64, 192, 0, 3), (1, 18, 22, 32, 64, 1, 3),
                                  (1, 16, 16, 80, 64, 0, 7)],
                         ids=lambda c: "B%d_%dx%d_%d-%d_p%d_k%d" % c)
def test_winograd_f4x4_kernel_on_interpreter(emu_engine, case):
    """conv_wino4.hip (option wino4 = 1) on the interpreter: F(4x4,3x3) with 36 planes -- 6x6 windows from row + column offset terms
    (image borders, the 7x7 block shifts), in-place B^T d B, weight packing [36][step][Cout/16][q][n][e] from G evaluated in double,
    A^T M A with ragged 4x4 tiles (sizes that are not multiples of 4), the fused pool as four maxima per tile, items that run on
    into the next image, and the channel split."""
    B, H, W, Cin, Cout, pool, k = case
    rng = np.random.default_rng(sum(case) + 4)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    r = T.leaky_relu(T.bias_add(T.conv2d_same(x, w, 1, acc=np.float64), b))
    if pool:
        r = T.max_pool_2x2(r)
    emu_engine.set_option('wino4', '1')
    try:
        for sk in ('0', '1'):
            emu_engine.set_option('wino_splitk', sk)
            n0 = emu_engine.counter('conv_wino4_launches')
            y = emu_engine.conv2d(x, w, b, 1, True, bool(pool))
            assert emu_engine.counter('conv_wino4_launches') == n0 + 1
            # F(4x4,3x3) in float32: ~10x the rounding error of F(2x2,3x3) per layer on unit-variance data
            assert np.abs(y - r).max() < 1e-4, (sk, np.abs(y - r).max())
    finally:
        emu_engine.set_option('wino4', 'auto')
        emu_engine.set_option('wino_splitk', '1')


def test_lift_fused_one_launch_lifting_stage_on_interpreter(emu_engine, synth_weights):
    """lift_fused.hip (phases as launches on the interpreter): PosePrior + ViewpointNet conv / fc chains, stride-2 SAME padding,
    hand-side concat, K-slice partial sums finished by the consumer, bottleneck variant -- against the oracle and the layer-by-layer
    kernels."""
    rng = np.random.default_rng(1)
    sm = (rng.standard_normal((3, 32, 32, 21)) * 0.3).astype(np.float32)
    hs = synth.hand_sides(3)
    emu_engine.load_weight_dict({k: v for k, v in synth_weights.items() if k.startswith(('PosePrior', 'ViewpointNet'))})
    emu_engine.finalize_weights()
    ref = N.pose3d(synth_weights, sm, hs, acc=np.float64)
    try:
        for mode in ('0', '1'):
            emu_engine.set_option('lift_fused', mode)
            n0 = emu_engine.counter('lift_fused_launches')
            out = emu_engine.pose3d(sm, hs)
            assert emu_engine.counter('lift_fused_launches') - n0 == int(mode)
            for a, b in zip(out, ref):
                assert np.abs(a - b).max() < 1e-5, mode
        wb = synth.make_weights(bottleneck=True)
        emu_engine.load_weight_dict({k: v for k, v in wb.items() if k.startswith('PosePrior')})
        emu_engine.finalize_weights()
        sm256 = synth.lifting_scoremaps(5, 2)
        r = N.poseprior_network(wb, 'bottleneck', sm256, synth.hand_sides(2))
        o = emu_engine.poseprior('bottleneck', sm256, synth.hand_sides(2))
        assert np.abs(o[0] - r[0]).max() < 1e-5
    finally:
        emu_engine.set_option('lift_fused', 'auto')


@pytest.mark.parametrize("case", [(1, 16, 32, 128, 256, 0), (1, 16, 32, 128, 256, 1), (4, 18, 22, 128, 64, 0), (4, 18, 22, 128, 64, 1), (7, 16, 32, 192, 64, 0),
                                  # two tail items: the middle workgroup's run crosses from one into the other (two pieces); less than one round in all
                                  (5, 16, 32, 128, 64, 0), (5, 16, 32, 128, 64, 1), (2, 16, 32, 128, 64, 0), (1, 16, 32, 64, 128, 1)],
                         ids=lambda c: "B%d_%dx%d_%d-%d_p%d" % c)
def test_winograd_f4x4_tail_pieces_on_interpreter(emu_engine, case):
    """conv_wino4.hip's TAIL: the interpreter's "chip" has 3 CUs, so 4 or 7 work items are one / two full rounds + ONE item, 5 items
    one round + TWO, 2 items less than a round.  The remainder's item-steps are shared out in equal runs, one per workgroup (a run
    may cross from one item into the next: two pieces), raw 4x4 sums go to the compact scratch, wino4_tail_reduce adds an item's
    pieces in step order (+ bias, leaky-ReLU, the pooled form, ragged edges, a tile block that runs past the last image).  The
    counter proves the path ran; with the option off the same layer runs unsplit and both agree with the float64 oracle."""
    B, H, W, Cin, Cout, pool = case
    rng = np.random.default_rng(sum(case) + 11)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    r = T.leaky_relu(T.bias_add(T.conv2d_same(x, w, 1, acc=np.float64), b))
    if pool:
        r = T.max_pool_2x2(r)
    emu_engine.set_option('wino4', '1')
    try:
        outs = {}
        for tail in ('1', '0'):
            emu_engine.set_option('wino4_tail', tail)
            n0, t0 = emu_engine.counter('conv_wino4_launches'), emu_engine.counter('conv_wino4_tail_launches')
            outs[tail] = emu_engine.conv2d(x, w, b, 1, True, bool(pool))
            assert emu_engine.counter('conv_wino4_launches') == n0 + 1
            assert emu_engine.counter('conv_wino4_tail_launches') == t0 + (1 if tail == '1' else 0)
            assert outs[tail].shape == r.shape and np.abs(outs[tail] - r).max() < 1e-4, (tail, np.abs(outs[tail] - r).max())
        # same products, another summation order -- in the tail item only: everything outside it is bit-identical
        diff = np.abs(outs['1'] - outs['0'])
        assert diff.max() < 1e-4 and 0 < (diff > 0).mean() <= (1.0 if B * Cout <= 128 else 0.6), (diff.max(), (diff > 0).mean())
    finally:
        emu_engine.set_option('wino4', 'auto')
        emu_engine.set_option('wino4_tail', '1')

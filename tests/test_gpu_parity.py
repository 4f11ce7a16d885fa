"""GPU parity tests proper: libhp3d.so (hand-written HIP, gfx950) vs the oracle, through the C ABI.

Tolerances (BASELINE.json north_star): heat-maps 1e-3, 3-D keypoints 1e-4, max-abs, against the
float64-accumulating oracle; integer/index outputs (masks, seeds, arg-max) bit-exact.
The oracle itself is pinned to the reference's executed code (tests/test_reference_pin.py, tests/golden/ref_*.npz);
TF 1.3's own kernels remain restated (oracle/__init__.py).
"""
import numpy as np
import pytest

from hand3d_amd import synth
from oracle import general as G
from oracle import nets as N
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu

TOL_HEATMAP = 1e-3
TOL_KP3D = 1e-4


def _conv_ref(x, w, b, stride, act, pool):
    r = T.bias_add(T.conv2d_same(x, w, stride, acc=np.float64), b)
    if act:
        r = T.leaky_relu(r)
    if pool:
        r = T.max_pool_2x2(r)
    return r


# (B,H,W,Cin,Cout,k,stride,pool): the layer geometries of SURVEY.md App. A + ragged/edge cases
CONV_CASES = [
    (1, 240, 320, 64, 64, 3, 1, 1),     # HandSegNet conv1_2 + pool1
    (2, 120, 160, 64, 128, 3, 1, 0),    # conv2_1
    (1, 60, 80, 256, 256, 3, 1, 1),     # conv3_4 + pool3 (60 rows: masked 8-row tiles)
    (2, 30, 40, 512, 512, 3, 1, 0),     # conv4_x (30x40: 8x8 tiles with masking)
    (1, 30, 40, 512, 128, 3, 1, 0),     # conv5_2
    (1, 30, 40, 128, 512, 1, 1, 0),     # conv6_1 (1x1)
    (1, 30, 40, 512, 2, 1, 1, 0),       # conv6_2 head (Cout=2)
    (1, 32, 32, 149, 128, 7, 1, 0),     # PoseNet conv6_1 (7x7, Cin=149)
    (2, 32, 32, 128, 128, 7, 1, 0),     # conv6_2..5
    (1, 32, 32, 128, 21, 1, 1, 0),      # conv6_7 head (Cout=21)
    (3, 32, 32, 21, 32, 3, 1, 0),       # conv_pose_0_1
    (3, 32, 32, 32, 32, 3, 2, 0),       # conv_pose_0_2 stride 2 (asymmetric SAME pad)
    (2, 8, 8, 128, 128, 3, 2, 0),       # conv_pose_2_2 -> 4x4
    (2, 8, 8, 256, 256, 3, 2, 0),       # conv_vp_2_2
    (1, 17, 23, 37, 45, 3, 1, 0),       # ragged everything
    (1, 18, 22, 40, 64, 3, 1, 1),       # ragged + pool
    (1, 15, 15, 32, 32, 3, 2, 0),       # odd size stride 2 (symmetric pad case)
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "B%d_%dx%d_%d-%d_k%ds%dp%d" % c)
def test_conv_mfma_vs_oracle(gpu_engine, case):
    B, H, W, Cin, Cout, k, stride, pool = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    act = Cout not in (2, 21)
    y = gpu_engine.conv2d(x, w, b, stride, act, pool)
    r = _conv_ref(x, w, b, stride, act, pool)
    assert y.shape == r.shape
    err = np.abs(y - r).max()
    print("conv %s max|err| %.3e" % (case, err))
    assert err < 5e-5, err   # f32 accumulation over K <= 7301 vs the f64 oracle, unit-variance data


WINO_CASES = [(1, 60, 80, 256, 256, 1), (2, 30, 40, 512, 512, 0), (1, 64, 64, 128, 256, 0), (1, 17, 23, 40, 128, 0),
              (2, 18, 22, 64, 128, 1), (1, 120, 160, 64, 128, 0), (4, 32, 32, 512, 256, 0),
              # more work items than CUs: the persistent grid walks several items per workgroup (work counter,
              # next-item prefetch, tile-table parity), three cout blocks, items that straddle images
              (8, 64, 64, 128, 256, 0), (6, 48, 56, 64, 384, 1),
              # Cout % 128 != 0: 64-tile x 64-cout items with 16-channel steps (conv1_2 is 64 -> 64 with the fused pool)
              (2, 64, 64, 64, 64, 1), (1, 33, 47, 32, 64, 0), (3, 40, 40, 96, 192, 0), (8, 128, 128, 64, 64, 1)]


@pytest.mark.parametrize("case", WINO_CASES, ids=lambda c: "B%d_%dx%d_%d-%d_p%d" % c)
def test_conv_winograd_vs_oracle(gpu_engine, case):
    """conv_wino.hip (F(2x2,3x3), float32) forced on: same tolerance as the direct kernel; masked edge tiles,
    fused pool, Cin not a multiple of 64 (zero-padded)."""
    B, H, W, Cin, Cout, pool = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    r = _conv_ref(x, w, b, 1, True, pool)
    gpu_engine.set_option('conv_impl', 'winograd')
    try:
        y = gpu_engine.conv2d(x, w, b, 1, True, bool(pool))
    finally:
        gpu_engine.set_option('conv_impl', 'mfma')
    gpu_engine.set_option('conv_impl', 'direct')
    try:
        yd = gpu_engine.conv2d(x, w, b, 1, True, bool(pool))
    finally:
        gpu_engine.set_option('conv_impl', 'mfma')
    err, errd = np.abs(y - r).max(), np.abs(yd - r).max()
    print("winograd %s max|err| %.3e (direct %.3e)" % (case, err, errd))
    assert y.shape == r.shape and err < 5e-5
    assert not np.array_equal(y, yd), "the Winograd kernel did not run"


W2_CASES = WINO_CASES + [(1, 32, 32, 256, 256, 0), (1, 32, 32, 512, 512, 0), (1, 128, 128, 64, 64, 1), (32, 160, 160, 64, 128, 0),
            (1, 64, 64, 128, 128, 1), (1, 30, 40, 512, 512, 0), (2, 60, 80, 128, 256, 1)]


@pytest.mark.parametrize("case", W2_CASES, ids=lambda c: "B%d_%dx%d_%d-%d_p%d" % c)
def test_conv_winograd_two_workgroups_per_cu_vs_oracle(gpu_engine, case):
    """conv_wino2.hip (option wino2 = 1: F(2x2,3x3) on v_mfma_f32_16x16x4_f32, two workgroups per CU, 16-channel steps) on the
    shapes of the conv_wino.hip test plus batch-1 PoseNet2D shapes (split channel steps) and the full-size Cin = 64 layers: same
    tolerance against the float64-accumulating oracle, and the launch counter proves which kernel ran."""
    B, H, W, Cin, Cout, pool = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    big = B * H * W * Cout > 3e7          # the NumPy oracle needs a minute there: compare with conv_wino.hip instead (itself oracle-checked)
    gpu_engine.set_option('wino2', '1')
    try:
        n0 = gpu_engine.counter('conv_wino2_launches')
        y = gpu_engine.conv2d(x, w, b, 1, True, bool(pool))
        assert gpu_engine.counter('conv_wino2_launches') == n0 + 1
        for _ in range(1 if big else 8):      # split launches reduce INSIDE the launch (agent-scope hand-off): a stale read would show here
            y2 = gpu_engine.conv2d(x, w, b, 1, True, bool(pool))
            assert np.array_equal(y, y2), "not deterministic"
    finally:
        gpu_engine.set_option('wino2', 'auto')
    if big:
        gpu_engine.set_option('wino2', '0')
        gpu_engine.set_option('conv_impl', 'winograd')
        try:
            r = gpu_engine.conv2d(x, w, b, 1, True, bool(pool))
        finally:
            gpu_engine.set_option('conv_impl', 'mfma')
            gpu_engine.set_option('wino2', 'auto')
    else:
        r = _conv_ref(x, w, b, 1, True, pool)
    err = np.abs(y - r).max()
    print("conv_wino2 %s max|err| %.3e" % (case, err))
    assert y.shape == r.shape and err < 5e-5


@pytest.mark.parametrize("case", [(1, 32, 32, 160, 128), (4, 32, 32, 128, 128), (32, 32, 32, 128, 128)], ids=lambda c: "B%d_%dx%d_%d-%d" % c)
def test_conv_winograd_two_workgroups_per_cu_7x7(gpu_engine, case):
    """7x7 layers of the PoseNet2D refinement units on conv_wino2.hip (nine 3x3 blocks, shifted windows; split channel steps at
    small batches)."""
    B, H, W, Cin, Cout = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((7, 7, Cin, Cout)) / np.sqrt(49 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    gpu_engine.set_option('wino2', '1')
    try:
        n0 = gpu_engine.counter('conv_wino2_launches')
        y = gpu_engine.conv2d(x, w, b, 1, True, False)
        assert gpu_engine.counter('conv_wino2_launches') == n0 + 1
    finally:
        gpu_engine.set_option('wino2', 'auto')
    if B <= 4:
        r = _conv_ref(x, w, b, 1, True, 0)
    else:
        gpu_engine.set_option('wino2', '0')
        gpu_engine.set_option('conv_impl', 'winograd')
        try:
            r = gpu_engine.conv2d(x, w, b, 1, True, False)
        finally:
            gpu_engine.set_option('conv_impl', 'mfma')
            gpu_engine.set_option('wino2', 'auto')
    err = np.abs(y - r).max()
    print("conv_wino2 7x7 %s max|err| %.3e" % (case, err))
    assert err < 5e-5


@pytest.mark.parametrize("case", [(10, 64, 64, 256, 256, 0), (32, 40, 40, 512, 512, 0), (32, 80, 80, 256, 256, 1), (5, 126, 158, 128, 128, 0), (3, 40, 40, 512, 128, 0)],
                         ids=lambda c: "B%d_%dx%d_%d-%d_p%d" % c)
def test_conv_winograd_f4x4_tail_pieces(gpu_engine, case):
    """conv_wino4.hip's TAIL (option wino4_tail, default on): a launch whose last round of work items is not full -- HandSegNet's 40x40
    layers at B = 32 are 800 items on 256 CUs -- shares that round's item-steps out in equal runs, one per CU (a run may cross from one
    item into the next), and wino4_tail_reduce adds an item's raw pieces in step order.  Shapes: a quarter round left, an eighth, a
    pooled layer, more than half a round with ragged tiles, less than one round in all.  Against the same kernel with the option off
    (only the tail items may differ, by summation order) and against conv_wino.hip (F(2x2,3x3), itself oracle-checked); deterministic;
    the counter proves the path ran."""
    B, H, W, Cin, Cout, pool = case
    rng = np.random.default_rng(sum(case) + 21)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    gpu_engine.set_option('wino4', '1')
    gpu_engine.set_option('wino_splitk', '0')       # (a launch of less than one round would otherwise take the whole-launch channel split)
    try:
        t0 = gpu_engine.counter('conv_wino4_tail_launches')
        y = gpu_engine.conv2d(x, w, b, 1, True, bool(pool))
        assert gpu_engine.counter('conv_wino4_tail_launches') == t0 + 1, "the launch has no tail pieces: wrong test shape for this chip"
        assert np.array_equal(y, gpu_engine.conv2d(x, w, b, 1, True, bool(pool))), "not deterministic"
        gpu_engine.set_option('wino4_tail', '0')
        y0 = gpu_engine.conv2d(x, w, b, 1, True, bool(pool))
        assert gpu_engine.counter('conv_wino4_tail_launches') == t0 + 2
    finally:
        gpu_engine.set_option('wino4_tail', '1')
        gpu_engine.set_option('wino_splitk', '1')
        gpu_engine.set_option('wino4', '0')
    gpu_engine.set_option('conv_impl', 'winograd')
    try:
        r = gpu_engine.conv2d(x, w, b, 1, True, bool(pool))
    finally:
        gpu_engine.set_option('conv_impl', 'mfma')
        gpu_engine.set_option('wino4', 'auto')
    d = np.abs(y - y0)
    frac = float((d > 0).mean())
    print("tail pieces %s: vs unsplit %.2e on %.1f %% of the outputs, vs F(2x2,3x3) %.2e" % (case, d.max(), 100 * frac, np.abs(y - r).max()))
    assert y.shape == r.shape and d.max() < 1e-4 and 0 < frac < (1.01 if B * H < 200 else 0.5) and np.abs(y - r).max() < 2e-4


W4_CASES = [(2, 16, 32, 64, 128, 0, 3), (1, 8, 8, 64, 64, 1, 3), (1, 7, 9, 128, 64, 0, 3), (1, 17, 21, 32, 64, 0, 3), (1, 30, 40, 512, 512, 0, 3),
            (1, 32, 32, 256, 256, 0, 3), (2, 60, 80, 128, 256, 1, 3), (3, 10, 6, 16, 64, 1, 3), (16, 64, 64, 256, 256, 0, 3), (16, 128, 128, 128, 128, 1, 3),
            (1, 32, 32, 160, 128, 0, 7), (4, 32, 32, 128, 128, 0, 7), (16, 32, 32, 128, 128, 0, 7), (2, 9, 11, 48, 64, 0, 7),
            # one 16-channel step per item; three cout blocks; a pooled layer with an odd pooled extent; a 7x7 split that cuts through the nine blocks
            (1, 16, 16, 16, 64, 0, 3), (1, 20, 24, 64, 192, 0, 3), (1, 18, 22, 32, 64, 1, 3), (1, 16, 16, 80, 64, 0, 7), (8, 64, 64, 64, 128, 0, 3)]


@pytest.mark.parametrize("case", W4_CASES, ids=lambda c: "B%d_%dx%d_%d-%d_p%d_k%d" % c)
def test_conv_winograd_f4x4_vs_oracle(gpu_engine, case):
    """conv_wino4.hip (option wino4 = 1: Winograd F(4x4,3x3), 36 planes, 288 accumulators per lane) against the float64-accumulating
    oracle on ragged sizes (not multiples of 4), the fused pool, items that continue into the next image, split channel steps, the
    7x7 layers as nine blocks and the real PoseNet2D shapes; deterministic; the launch counter proves which kernel ran.  Tolerance:
    F(4x4,3x3) multiplies by up to 8 / divides by up to 24 in float32 -- per layer on unit-variance data it is ~10x F(2x2,3x3)'s
    error (gate here 2e-4; the conv_wino.hip test's is 5e-5); end to end the executor's gates stay 1e-3 / 1e-4."""
    B, H, W, Cin, Cout, pool, k = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    big = B * H * W * Cout * k * k > 2.5e8          # the NumPy oracle needs minutes there: compare with conv_wino.hip instead (itself oracle-checked)
    gpu_engine.set_option('wino4', '1')
    try:
        n0 = gpu_engine.counter('conv_wino4_launches')
        y = gpu_engine.conv2d(x, w, b, 1, True, bool(pool))
        assert gpu_engine.counter('conv_wino4_launches') == n0 + 1
        for _ in range(1 if big else 4):
            assert np.array_equal(y, gpu_engine.conv2d(x, w, b, 1, True, bool(pool))), "not deterministic"
    finally:
        gpu_engine.set_option('wino4', 'auto')
    if big:
        gpu_engine.set_option('wino4', '0')
        gpu_engine.set_option('conv_impl', 'winograd')
        try:
            r = gpu_engine.conv2d(x, w, b, 1, True, bool(pool))
        finally:
            gpu_engine.set_option('conv_impl', 'mfma')
            gpu_engine.set_option('wino4', 'auto')
    else:
        r = _conv_ref(x, w, b, 1, True, pool)
    err = np.abs(y - r).max()
    print("conv_wino4 %s max|err| %.3e" % (case, err))
    assert y.shape == r.shape and err < 2e-4


W4S_CASES = [c[:6] for c in W4_CASES if c[6] == 3] + [(8, 80, 80, 256, 256, 0), (8, 40, 40, 512, 512, 0), (4, 160, 160, 128, 128, 1)]


@pytest.mark.parametrize("case", W4S_CASES, ids=lambda c: "B%d_%dx%d_%d-%d_p%d" % c)
def test_conv_winograd_f4x4_split_operands_vs_oracle(gpu_engine, case):
    """conv_wino4s.hip (option wino4_split = 1, round 6): F(4x4,3x3) with the 36 plane products on v_mfma_f32_16x16x32_bf16 over three bfloat16
    pieces per operand (six products, float32 accumulate) against the float64-accumulating oracle on conv_wino4's 3x3 shapes (ragged, pooled,
    tail pieces, items that continue into the next image) and the trunk's own; deterministic; the launch counter proves which kernel ran.
    Gate: conv_wino4's (2e-4 on unit-variance data) -- measured on the GPU the two kernels' errors are within 0.7 ... 1.6x of each other
    (profiles/r06_split_numerics.md: the matrix pipe's accumulation is no more exact than a float32 fmaf chain), so the split form keeps the
    gate but does NOT halve the error as its CPU emulation had (profiles/r05_splithalf_numerics.md)."""
    B, H, W, Cin, Cout, pool = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    big = B * H * W * Cout * 9 > 2.5e8          # the NumPy oracle needs minutes there: compare with conv_wino.hip instead (itself oracle-checked)
    gpu_engine.set_option('wino4_split', '1')
    try:
        n0 = gpu_engine.counter('conv_wino4s_launches')
        y = gpu_engine.conv2d(x, w, b, 1, True, bool(pool))
        assert gpu_engine.counter('conv_wino4s_launches') == n0 + 1
        for _ in range(1 if big else 4):
            assert np.array_equal(y, gpu_engine.conv2d(x, w, b, 1, True, bool(pool))), "not deterministic"
    finally:
        gpu_engine.set_option('wino4_split', '0')
    if big:
        gpu_engine.set_option('wino4', '0')
        gpu_engine.set_option('conv_impl', 'winograd')
        try:
            r = gpu_engine.conv2d(x, w, b, 1, True, bool(pool))
        finally:
            gpu_engine.set_option('conv_impl', 'mfma')
            gpu_engine.set_option('wino4', 'auto')
    else:
        r = _conv_ref(x, w, b, 1, True, pool)
    err = np.abs(y - r).max()
    print("conv_wino4s %s max|err| %.3e" % (case, err))
    assert y.shape == r.shape and err < 2e-4


def test_full_pipeline_split_operands_option(gpu_engine, synth_weights):
    """Option wino4_split = auto on the whole path (B = 32, 240x320: every 3x3 trunk layer with Cin >= 128 whose launch fills the chip goes to
    conv_wino4s.hip, the rest stays): against the default run score-map logits, heat-maps and 3-D keypoints agree to Winograd rounding,
    crop boxes and 2-D keypoints are identical unless the image holds a knife-edge mask pixel (the same margin rule as
    test_full_pipeline_batch32_winograd_active), and the option is OFF by default (VERDICT r5's headline rule: its per-layer error is not below
    conv_wino4's on every shape, so it stays an option)."""
    gpu_engine.load_weight_dict(synth_weights)
    gpu_engine.finalize_weights()
    img = synth.make_batch(900, 32, 240, 320)
    hs = synth.hand_sides(32)
    names = ('scoremap', 'scale', 'center', 'kpmap', 'coord3d', 'kp_crop')
    n0 = gpu_engine.counter('conv_wino4s_launches')
    r = gpu_engine.infer_full(img, hs, want_mask=True, outputs=names)
    assert gpu_engine.counter('conv_wino4s_launches') == n0, "the split-operand kernel must be off by default"
    gpu_engine.set_option('wino4_split', 'auto')
    try:
        o = gpu_engine.infer_full(img, hs, want_mask=True, outputs=names)
        ns = gpu_engine.counter('conv_wino4s_launches') - n0
    finally:
        gpu_engine.set_option('wino4_split', '0')
    print("conv_wino4s launches per call:", ns)
    assert ns >= 18, "the filled 3x3 layers with Cin >= 128 of both trunks should run on conv_wino4s.hip"
    d_sm = np.abs(o['scoremap'] - r['scoremap']).max()
    det_o = o['scoremap'][..., 1] > o['scoremap'][..., 0]
    det_r = r['scoremap'][..., 1] > r['scoremap'][..., 0]
    margin = np.abs(r['scoremap'][..., 1] - r['scoremap'][..., 0])
    assert d_sm < 1e-4 and (margin[det_o != det_r] < 1e-4).all(), "a pixel with a clear logit margin changed class"
    same = [i for i in range(32) if np.array_equal(o['mask'][i], r['mask'][i])]
    print("score map %.2e; identical masks on %d of 32 images" % (d_sm, len(same)))
    assert len(same) >= 24          # (a knife-edge flip moves a mask on ~5 % of the synthetic images per kernel plan: profiles/r06_split_numerics.md)
    assert np.array_equal(o['center'][same], r['center'][same]) and np.array_equal(o['scale'][same], r['scale'][same])
    d_hm = np.abs(o['kpmap'][same] - r['kpmap'][same]).max()
    d_kp = np.abs(o['coord3d'][same] - r['coord3d'][same]).max()
    print("heat-maps %.2e, 3-D keypoints %.2e (images with identical masks)" % (d_hm, d_kp))
    assert d_hm < 1e-4 and d_kp < 2e-5


def test_full_pipeline_winograd_f4x4_policy(gpu_engine, synth_weights):
    """Option wino4 on the whole path (B = 16, 240x320): "0" runs no conv_wino4 launch, "pose" only PoseNet2D's, "auto" (the default)
    both trunks'; all three give the same crop box and 2-D keypoints, score maps / heat-maps / 3-D keypoints agree to float32 Winograd
    rounding (1e-4 score-map logits... measured 1e-5, 2e-5 heat-maps, 1e-5 keypoints: two orders inside the north-star gates 1e-3 / 1e-4),
    and a hand mask may differ from the F(2x2,3x3) run's only in pixels whose two logits are equal to rounding."""
    gpu_engine.load_weight_dict(synth_weights)
    gpu_engine.finalize_weights()
    img = synth.make_batch(700, 16, 240, 320)
    hs = synth.hand_sides(16)
    outs, counts = {}, {}
    try:
        for mode in ('0', 'pose', 'auto'):
            gpu_engine.set_option('wino4', mode)
            n0 = gpu_engine.counter('conv_wino4_launches')
            outs[mode] = gpu_engine.infer_full(img, hs, want_mask=True, outputs=('scoremap', 'scale', 'center', 'kpmap', 'coord3d', 'kp_crop'))
            counts[mode] = gpu_engine.counter('conv_wino4_launches') - n0
    finally:
        gpu_engine.set_option('wino4', 'auto')
    print("conv_wino4 launches per call:", counts)
    assert counts['0'] == 0 and 0 < counts['pose'] < counts['auto']
    r = outs['0']
    for mode in ('pose', 'auto'):
        o = outs[mode]
        assert np.array_equal(o['center'], r['center']) and np.array_equal(o['scale'], r['scale']), mode
        assert np.array_equal(o['kp_crop'], r['kp_crop']), mode
        d_sm = np.abs(o['scoremap'] - r['scoremap']).max()
        d_hm = np.abs(o['kpmap'] - r['kpmap']).max()
        d_kp = np.abs(o['coord3d'] - r['coord3d']).max()
        print("wino4=%s vs F(2x2,3x3): score map %.2e, heat-maps %.2e, 3-D keypoints %.2e" % (mode, d_sm, d_hm, d_kp))
        assert d_sm < 1e-4 and d_hm < 1e-4 and d_kp < 2e-5, mode
        if mode == 'pose':
            assert np.array_equal(o['mask'], r['mask'])          # HandSegNet untouched: identical masks
        else:
            # the class decision per pixel may only change where the two logits are equal to rounding
            det_o = o['scoremap'][..., 1] > o['scoremap'][..., 0]
            det_r = r['scoremap'][..., 1] > r['scoremap'][..., 0]
            margin = np.abs(r['scoremap'][..., 1] - r['scoremap'][..., 0])
            assert (margin[det_o != det_r] < 1e-4).all(), "a pixel with a clear logit margin changed class"


def test_conv_mfma_vs_naive_kernel(gpu_engine):
    """Same op through the obviously-correct one-thread-per-output kernel (debug path)."""
    rng = np.random.default_rng(7)
    x = rng.standard_normal((1, 24, 40, 96)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 96, 160)) / 30).astype(np.float32)
    b = rng.standard_normal(160).astype(np.float32)
    y = gpu_engine.conv2d(x, w, b, 1, True, False)
    gpu_engine.set_option('conv_impl', 'naive')
    try:
        y2 = gpu_engine.conv2d(x, w, b, 1, True, False)
    finally:
        gpu_engine.set_option('conv_impl', 'mfma')
    assert np.abs(y - y2).max() < 2e-5


def test_conv_linearity_full_size(gpu_engine):
    """Size-independent property at a BASELINE-size layer (B=8, 80x80, 256->256): without bias
    and activation conv(a*x1 + x2) == a*conv(x1) + conv(x2) to rounding."""
    rng = np.random.default_rng(11)
    x1 = rng.standard_normal((8, 80, 80, 256)).astype(np.float32)
    x2 = rng.standard_normal((8, 80, 80, 256)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 256, 256)) / 48).astype(np.float32)
    b = np.zeros(256, np.float32)
    y1 = gpu_engine.conv2d(x1, w, b, 1, False, False)
    y2 = gpu_engine.conv2d(x2, w, b, 1, False, False)
    y3 = gpu_engine.conv2d(2.0 * x1 + x2, w, b, 1, False, False)
    assert np.abs(y3 - (2.0 * y1 + y2)).max() < 5e-5


def test_glue_ops_bit_exact(gpu_engine):
    e = gpu_engine
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 60, 80, 37)).astype(np.float32)
    assert np.array_equal(e.maxpool2(x), T.max_pool_2x2(x))
    x = rng.standard_normal((2, 256, 256, 21)).astype(np.float32)
    assert np.abs(e.avgpool8(x) - T.avg_pool_8x8(x)).max() < 1e-6
    x = rng.standard_normal((2, 30, 40, 2)).astype(np.float32)
    assert np.array_equal(e.resize_bilinear(x, 240, 320), T.resize_bilinear_legacy(x, 240, 320))
    x = rng.standard_normal((1, 32, 32, 21)).astype(np.float32)
    assert np.array_equal(e.resize_bilinear(x, 256, 256), T.resize_bilinear_legacy(x, 256, 256))
    assert np.array_equal(e.resize_bilinear(x, 32, 32), x)   # equal sizes => identity


def test_crop_and_resize_vs_oracle(gpu_engine):
    rng = np.random.default_rng(2)
    img = rng.uniform(-.5, .5, (6, 240, 320, 3)).astype(np.float32)
    center = np.array([[120, 160], [5, 300], [239.5, 2], [160, 160], [0, 0], [53, 41]], np.float32)
    scale = np.array([1.0, 5.0, 0.25, 3.1030303, 0.5, 2.2], np.float32)   # incl. both clip ends
    y = gpu_engine.crop_and_resize(img, center, scale, 256)
    r = G.crop_image_from_xy(img, center, 256, scale)
    assert np.array_equal(y == 0, r == 0), "extrapolated (zero) region differs"
    assert np.abs(y - r).max() < 1e-6


def test_fc_vs_oracle(gpu_engine):
    rng = np.random.default_rng(3)
    for B, Cin, Cout, act in [(1, 2050, 512, True), (32, 4098, 256, True), (5, 512, 63, False), (3, 128, 3, False)]:
        x = rng.standard_normal((B, Cin)).astype(np.float32)
        w = (rng.standard_normal((Cin, Cout)) / np.sqrt(Cin)).astype(np.float32)
        b = rng.standard_normal(Cout).astype(np.float32)
        r = T.fully_connected(x, w, b, np.float64)
        if act:
            r = T.leaky_relu(r)
        assert np.abs(gpu_engine.fc(x, w, b, act) - r).max() < 1e-5


def test_argmax2d_first_index(gpu_engine):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 256, 256, 21)).astype(np.float32)
    x[0, 10, 20, 3] = x[0, 200, 5, 3] = 99.0        # tie: first (row-major) wins
    x[1, :, :, 7] = -1.5                            # all equal -> index 0
    got = gpu_engine.argmax2d(x)
    for b in range(2):
        for c in range(21):
            v, u = np.unravel_index(np.argmax(x[b, :, :, c]), (256, 256))
            assert tuple(got[b, c]) == (v, u)


def _blob_scoremap(H, W, boxes, seedpix=None, base=-3.0):
    sm = np.zeros((1, H, W, 2), np.float32)
    sm[..., 1] = base
    for (y0, y1, x0, x1, v) in boxes:
        sm[0, y0:y1, x0:x1, 1] = v
    if seedpix:
        sm[0, seedpix[0], seedpix[1], 1] = 9.0
    return sm


def test_mask_engineered_cases(gpu_engine):
    H, W = 240, 320
    cases = {
        'two_blobs_far': _blob_scoremap(H, W, [(20, 60, 30, 90, 2.0), (150, 220, 200, 300, 2.5)]),
        'two_blobs_gap10': _blob_scoremap(H, W, [(50, 80, 50, 100, 2.0), (50, 80, 110, 160, 3.0)]),   # gap 10 px: bridged
        'two_blobs_gap11': _blob_scoremap(H, W, [(50, 80, 50, 100, 2.0), (50, 80, 111, 160, 3.0)]),   # gap 11 px: not bridged
        'empty': _blob_scoremap(H, W, []),
        'full': _blob_scoremap(H, W, [(0, H, 0, W, 1.0)]),
        'single_pixel': _blob_scoremap(H, W, [(77, 78, 123, 124, 4.0)]),
        'border': _blob_scoremap(H, W, [(0, 5, 0, W, 2.0), (0, H, W - 3, W, 2.0)], seedpix=(0, 0)),
    }
    # a serpentine needing more than 32 passes: the pass cap must bite identically
    sm = _blob_scoremap(H, W, [])
    for i, y in enumerate(range(4, 236, 24)):
        sm[0, y:y + 2, 4:316, 1] = 2.0
        xs = 314 if i % 2 == 0 else 4
        sm[0, y:y + 26, xs:xs + 2, 1] = 2.0
    sm[0, 4, 4, 1] = 9.0
    cases['serpentine_pass_cap'] = sm
    for name, sm in cases.items():
        mask, center, size, scale, seed = gpu_engine.mask_from_scoremap(sm)
        rm = G.single_obj_scoremap(sm)[..., 0]
        rc, _, rs = G.calc_center_bb(rm[..., None])
        fg, _ = G.fg_and_detmap(sm)
        assert np.array_equal(seed, G.find_max_location(fg)), name
        assert np.array_equal(mask, rm), (name, mask.sum(), rm.sum())
        assert np.array_equal(center, rc) and np.array_equal(size, rs), (name, center, rc, size, rs)
        assert np.array_equal(scale, G.scale_from_crop_size(rs)), name
    rm = G.single_obj_scoremap(cases['serpentine_pass_cap'])[..., 0]
    full = G.grow_objectmap(G.fg_and_detmap(cases['serpentine_pass_cap'])[1][0], (4, 4), num_passes=500, early_exit=True)[0]
    assert rm.sum() < full.sum(), "test input does not exercise the 32-pass cap"


def test_mask_empty_reduce_option(gpu_engine):
    sm = _blob_scoremap(240, 320, [])
    gpu_engine.set_option('empty_reduce', 'fltmax')
    try:
        _, center, size, _, _ = gpu_engine.mask_from_scoremap(sm)
    finally:
        gpu_engine.set_option('empty_reduce', 'inf')
    assert center.tolist() == [[0.0, 0.0]] and size.tolist() == [[100.0]]
    _, center, size, _, _ = gpu_engine.mask_from_scoremap(sm)
    assert center.tolist() == [[160.0, 160.0]] and size.tolist() == [[100.0]]


@pytest.fixture(scope='module')
def net(gpu_engine, synth_weights):
    from hand3d_amd import ColorHandPose3DNetwork
    n = ColorHandPose3DNetwork(engine=gpu_engine)
    n.init_from_dict(synth_weights)
    return n


def test_handsegnet_parity(net, synth_weights):
    img = synth.make_batch(0, 2, 240, 320)
    large, small = net.engine.handsegnet(img, want_small=True)
    rs, rl = N.handsegnet(synth_weights, img, acc=np.float64)
    print("HandSegNet small err %.3e large err %.3e" % (np.abs(small - rs).max(), np.abs(large - rl[0]).max()))
    assert np.abs(small - rs).max() < TOL_HEATMAP / 10
    assert np.abs(large - rl[0]).max() < TOL_HEATMAP / 10
    assert np.array_equal(net.inference_detection(img)[0], large)


def test_posenet_parity_config_c2(net, synth_weights):
    """BASELINE config 2: PoseNet-only, 256x256 crop, batch 1 (seed 100)."""
    crop = synth.make_batch(100, 1, 256, 256)
    sms = net.inference_pose2d(crop)
    ref = N.posenet2d(synth_weights, crop, acc=np.float64)
    assert len(sms) == 3
    for a, b in zip(sms, ref):
        assert a.shape == (1, 32, 32, 21)
        print("PoseNet2D scoremap err %.3e (|ref| max %.3f)" % (np.abs(a - b).max(), np.abs(b).max()))
        assert np.abs(a - b).max() < TOL_HEATMAP / 10


@pytest.mark.parametrize("case", [(32, 320, 320), (32, 256, 256), (3, 240, 320), (1, 37, 53), (5, 200, 264)], ids=lambda c: "B%d_%dx%d" % c)
def test_first_layer_kernel_balanced_tile_runs(gpu_engine, case):
    """conv_first.hip (conv1_1, nets/ColorHandPose3DNetwork.py:144,183), round 5: workgroups walk balanced runs of consecutive tiles (one
    workgroup per resident slot) instead of whole tile rows.  The bench shapes (25 600 and 16 384 tiles on 768 slots: runs of 33 / 34 and
    21 / 22 tiles that cross tile rows and images), C1's, ragged small ones.  Bit-identical to the row walk; float64 oracle on a sample."""
    B, H, W = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, H, W, 3)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, 64)) / np.sqrt(27)).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    n0 = gpu_engine.counter('conv_first_launches')
    y = gpu_engine.conv2d(x, w, b, 1, True, False)
    assert gpu_engine.counter('conv_first_launches') == n0 + 1
    gpu_engine.set_option('first_walk', 'rows')
    try:
        y_rows = gpu_engine.conv2d(x, w, b, 1, True, False)
    finally:
        gpu_engine.set_option('first_walk', 'balanced')
    assert np.array_equal(y, y_rows)
    for i in sorted({0, B // 2, B - 1}):
        r = T.leaky_relu(T.bias_add(T.conv2d_same(x[i:i + 1], w, 1, acc=np.float64), b))
        assert np.abs(y[i:i + 1] - r).max() < 1e-5


def test_cold_input_image_is_streamed_through_the_cache_first(gpu_engine, synth_weights):
    """Option "first_touch" (round 5): HandSegNet's conv1_1 gathers a COLD image (the caller's buffer / an upload) one tile ahead and loses a
    third of its store rate to the misses; a read pass over the image in front of it warms the memory-side cache.  The pass runs once per
    whole-path call for a batch whose image is 8 ... 128 MB (not for PoseNet2D's crop, which crop_and_resize has just written; not for
    small batches), reads only -- results bit-identical to the run without it."""
    from hand3d_amd import ColorHandPose3DNetwork
    net = ColorHandPose3DNetwork(engine=gpu_engine)
    net.init_from_dict(synth_weights)
    img = synth.make_batch(5100, 16, 240, 320)          # 14.7 MB
    hs = synth.hand_sides(16)
    gpu_engine.set_option('first_touch', '0')
    try:
        n0 = gpu_engine.counter('first_touch_launches')
        ref = net.inference(img, hs, True)
        assert gpu_engine.counter('first_touch_launches') == n0
    finally:
        gpu_engine.set_option('first_touch', 'auto')
    n0 = gpu_engine.counter('first_touch_launches')
    out = net.inference(img, hs, True)
    assert gpu_engine.counter('first_touch_launches') == n0 + 1
    for a, b in zip(out, ref):
        assert np.array_equal(a, b)
    n0 = gpu_engine.counter('first_touch_launches')
    net.inference(img[:2], hs[:2], True)                  # 1.8 MB: below the threshold
    assert gpu_engine.counter('first_touch_launches') == n0


def test_lifting_towers_on_two_streams_equal_serial(gpu_engine, synth_weights):
    """Unfused lifting stage (batches above 4), option "lift_overlap" (round 5): ViewpointNet on the child context's stream beside PosePrior
    (nets/ColorHandPose3DNetwork.py:231-235 -- the towers share only the pooled score map and the hand side).  Same kernels, so the
    results are bit-identical to the one-stream run, call after call (a missing fork / join dependency would show as a difference or as
    garbage in the rotation), growing and shrinking batches (the side tower's buffers follow), and within tolerance of the float64 oracle;
    the counter proves the two-stream form ran."""
    gpu_engine.load_weight_dict(synth_weights)
    gpu_engine.finalize_weights()
    rng = np.random.default_rng(81)
    for B in (8, 32, 6):
        sm = np.maximum(rng.standard_normal((B, 32, 32, 21)).astype(np.float32), 0) * 0.3
        hs = synth.hand_sides(B)
        gpu_engine.set_option('lift_overlap', '0')
        try:
            n0 = gpu_engine.counter('lift_overlap_calls')
            ref = gpu_engine.pose3d(sm, hs)
            assert gpu_engine.counter('lift_overlap_calls') == n0
        finally:
            gpu_engine.set_option('lift_overlap', '1')
        for it in range(6):
            n0 = gpu_engine.counter('lift_overlap_calls')
            out = gpu_engine.pose3d(sm, hs)
            assert gpu_engine.counter('lift_overlap_calls') == n0 + 1
            for a, b in zip(out, ref):
                assert np.array_equal(a, b), (B, it)
        rrel, rcan, rR = N.pose3d(synth_weights, sm[:2], hs[:2], acc=np.float64)
        assert np.abs(out[0][:2] - rrel).max() < TOL_KP3D and np.abs(out[2][:2] - rR).max() < TOL_KP3D


def test_pose3d_and_poseprior_variants(gpu_engine, synth_weights):
    from hand3d_amd import PosePriorNetwork
    rng = np.random.default_rng(8)
    B = 5
    sm256 = np.maximum(rng.standard_normal((B, 256, 256, 21)).astype(np.float32), 0) * 0.2
    hs = synth.hand_sides(B)
    wprior = {k: v for k, v in synth_weights.items() if k.startswith(('PosePrior', 'ViewpointNet'))}
    p = PosePriorNetwork('proposed', engine=gpu_engine)
    p.init_from_dict(wprior)
    rel, c3d, R = p.inference(sm256, hs, True)
    rrel, rc3d, rR = N.poseprior_network(synth_weights, 'proposed', sm256, hs, acc=np.float64)
    assert np.abs(rel - rrel).max() < TOL_KP3D and np.abs(c3d - rc3d).max() < TOL_KP3D and np.abs(R - rR).max() < TOL_KP3D
    assert np.abs(np.einsum('bij,bkj->bik', R, R) - np.eye(3)).max() < 1e-5   # rotation matrices
    d = PosePriorNetwork('direct', engine=gpu_engine)
    rel, c3d, R = d.inference(sm256, hs, True)
    rrel, _, _ = N.poseprior_network(synth_weights, 'direct', sm256, hs, acc=np.float64)
    assert R is None and np.abs(rel - rrel).max() < TOL_KP3D and np.array_equal(rel, c3d)
    wb = synth.make_weights(bottleneck=True)
    wbp = {k: v for k, v in wb.items() if k.startswith('PosePrior')}
    bn = PosePriorNetwork('bottleneck', engine=gpu_engine)
    bn.init_from_dict(wbp)
    rel, _, R = bn.inference(sm256, hs, True)
    rrel, _, _ = N.poseprior_network(wb, 'bottleneck', sm256, hs, acc=np.float64)
    assert R is None and np.abs(rel - rrel).max() < TOL_KP3D
    lo = PosePriorNetwork('local', engine=gpu_engine)          # PosePrior net + bone_rel_trafo_inv
    lo.init_from_dict(wprior)
    rel, c3d, R = lo.inference(sm256, hs, True)
    rrel, rc3d, _ = N.poseprior_network(synth_weights, 'local_w_xyz_loss', sm256, hs, acc=np.float64)
    assert R is None and np.abs(c3d - rc3d).max() < TOL_KP3D and np.abs(rel - rrel).max() < TOL_KP3D
    # restore the standard weight set for later tests
    gpu_engine.load_weight_dict(synth_weights)
    gpu_engine.finalize_weights()


def test_lift_fused_one_launch_lifting_stage(gpu_engine, synth_weights):
    """lift_fused.hip: PosePrior + ViewpointNet (12 convolutions, 6 fully connected layers) as ONE launch with grid barriers
    between the layers -- against the oracle (float64 accumulation), against the layer-by-layer kernels, for every batch size the
    auto policy sends there (B <= 4) and a larger forced one, all variants; repeated calls must be bit-identical (a stale read
    across a grid barrier would show as a difference between calls)."""
    from hand3d_amd import PosePriorNetwork
    rng = np.random.default_rng(18)
    for B in (1, 3, 4, 9):
        sm = (rng.standard_normal((B, 32, 32, 21)) * 0.3).astype(np.float32)
        hs = synth.hand_sides(B)
        ref = N.pose3d(synth_weights, sm, hs, acc=np.float64)
        outs = {}
        for mode in ('0', '1'):
            gpu_engine.set_option('lift_fused', mode)
            n0 = gpu_engine.counter('lift_fused_launches')
            outs[mode] = gpu_engine.pose3d(sm, hs)
            assert gpu_engine.counter('lift_fused_launches') - n0 == int(mode)
            for a, b in zip(outs[mode], ref):
                assert np.abs(a - b).max() < TOL_KP3D, (B, mode, float(np.abs(a - b).max()))
        for _ in range(10):
            again = gpu_engine.pose3d(sm, hs)
            for a, b in zip(again, outs['1']):
                assert np.array_equal(a, b), "the one-launch lifting stage is not deterministic"
        print("lift_fused B=%d: vs oracle %s, vs layer-by-layer %s" % (B, ['%.1e' % np.abs(a - b).max() for a, b in zip(outs['1'], ref)],
                                                                       ['%.1e' % np.abs(a - b).max() for a, b in zip(outs['1'], outs['0'])]))
    gpu_engine.set_option('lift_fused', '1')
    try:
        sm256 = np.maximum(rng.standard_normal((3, 256, 256, 21)).astype(np.float32), 0) * 0.2
        hs = synth.hand_sides(3)
        wprior = {k: v for k, v in synth_weights.items() if k.startswith(('PosePrior', 'ViewpointNet'))}
        for variant, ref_variant, wts in (('proposed', 'proposed', None), ('direct', 'direct', None), ('local', 'local', None), ('bottleneck', 'bottleneck', 'bn')):
            w = synth_weights
            if wts == 'bn':
                w = synth.make_weights(bottleneck=True)
                net = PosePriorNetwork(variant, engine=gpu_engine)
                net.init_from_dict({k: v for k, v in w.items() if k.startswith('PosePrior')})
            else:
                net = PosePriorNetwork(variant, engine=gpu_engine)
                net.init_from_dict(wprior)
            n0 = gpu_engine.counter('lift_fused_launches')
            got = net.inference(sm256, hs, True)
            assert gpu_engine.counter('lift_fused_launches') == n0 + 1
            exp = N.poseprior_network(w, ref_variant, sm256, hs, acc=np.float64)
            for a, b in zip(got, exp):
                assert (a is None) == (b is None)
                if a is not None:
                    assert np.abs(a - b).max() < TOL_KP3D, variant
    finally:
        gpu_engine.set_option('lift_fused', 'auto')
        gpu_engine.load_weight_dict(synth_weights)
        gpu_engine.finalize_weights()


def _full_parity(net, weights, img, hs):
    o = net.engine.infer_full(img, hs, want_mask=True)
    taps = {}
    ref = N.inference(weights, img, hs, True, acc=np.float64, taps=taps)
    rmask = taps['hand_mask'][..., 0]
    rep = dict(
        scoremap=np.abs(o['scoremap'] - ref[0]).max(),
        mask_equal=bool(np.array_equal(o['mask'], rmask)),
        mask_diff_px=int((o['mask'] != rmask).sum()),
        center=np.abs(o['center'] - ref[3]).max(), scale=np.abs(o['scale'] - ref[2]).max(),
        crop=np.abs(o['crop'] - ref[1]).max(), kpmap=np.abs(o['kpmap'] - ref[4]).max(),
        coord3d=np.abs(o['coord3d'] - ref[5]).max())
    return o, ref, rep


def test_full_pipeline_parity_config_c1(net, synth_weights):
    """BASELINE config 1 shape: inference(), B=1, 240x320, 5 seeded images, hand_side [[1,0]]."""
    from hand3d_amd.utils.general import EvalUtil
    ev = EvalUtil()
    for seed in range(5):
        img = synth.make_batch(seed, 1, 240, 320)
        hs = np.array([[1.0, 0.0]], np.float32)
        o, ref, rep = _full_parity(net, synth_weights, img, hs)
        print("seed %d: %s" % (seed, rep))
        assert rep['scoremap'] < TOL_HEATMAP
        assert rep['mask_equal'], "mask differs in %d px (knife-edge?)" % rep['mask_diff_px']
        assert rep['center'] == 0 and rep['scale'] == 0
        assert rep['crop'] < 1e-5
        assert rep['kpmap'] < TOL_HEATMAP
        assert rep['coord3d'] < TOL_KP3D
        ev.feed(ref[5][0], np.ones(21), o['coord3d'][0])
    mean_epe, _, _, _, _ = ev.get_measures(0.0, 0.05, 20)
    print("mean EPE engine vs oracle: %.3e" % mean_epe)
    assert mean_epe < TOL_KP3D


def test_full_pipeline_batch_and_2d(net, synth_weights):
    """B=4 with alternating hand sides == four B=1 calls; inference2d return order."""
    img = synth.make_batch(300, 4, 240, 320)
    hs = synth.hand_sides(4)
    o = net.inference(img, hs, True)
    assert [a.shape for a in o] == [(4, 240, 320, 2), (4, 256, 256, 3), (4, 1), (4, 2), (4, 256, 256, 21), (4, 21, 3)]
    # B=1 takes the small-batch plan (narrower cout tiles, split-K): same math, different summation
    # order -> equal to rounding, with identical discrete decisions (centre, scale)
    for i in range(4):
        oi = net.inference(img[i:i + 1], hs[i:i + 1], True)
        assert np.array_equal(o[3][i:i + 1], oi[3]) and np.array_equal(o[2][i:i + 1], oi[2])
        for a, b in zip(o, oi):
            assert np.abs(a[i:i + 1] - b).max() < 2e-5, "batched result differs from single-image result"
    kp, crop, scale, center = net.inference2d(img)
    assert np.array_equal(kp, o[4]) and np.array_equal(crop, o[1]) and np.array_equal(scale, o[2]) and np.array_equal(center, o[3])
    ref = N.inference(synth_weights, img, hs, True, acc=np.float64)
    assert np.abs(o[5] - ref[5]).max() < TOL_KP3D and np.abs(o[4] - ref[4]).max() < TOL_HEATMAP


def test_full_pipeline_320x320_and_determinism(net, synth_weights):
    """BASELINE config 3 shape (raw 320x320 RHD frame) + run-to-run bit reproducibility."""
    img = synth.make_batch(200, 2, 320, 320)
    hs = synth.hand_sides(2)
    o1 = net.inference(img, hs, True)
    o2 = net.inference(img, hs, True)
    for a, b in zip(o1, o2):
        assert np.array_equal(a, b)
    ref = N.inference(synth_weights, img, hs, True, acc=np.float64)
    assert np.abs(o1[0] - ref[0]).max() < TOL_HEATMAP
    assert np.array_equal(o1[3], ref[3]) and np.array_equal(o1[2], ref[2])
    assert np.abs(o1[4] - ref[4]).max() < TOL_HEATMAP and np.abs(o1[5] - ref[5]).max() < TOL_KP3D


def test_full_pipeline_batch32_winograd_active(net, synth_weights):
    """The bench workload shape (B=32, 320x320): at this size the default policy puts every 3x3 / stride-1 trunk layer with
    Cout % 64 == 0 on conv_wino4.hip (Winograd F(4x4,3x3), the headline kernel) and the ten 7x7 layers on conv_wino7.hip (F(4x4,4x4) over
    the filter's four 4x4-tap blocks, round 5; the nine-3x3-block form on conv_wino4.hip until round 4) -- asserted by kernel name
    and by the engine's launch counter, so a silent fall-back to another kernel fails here.  EVERY image of the batch is checked
    against the oracle (two of them against its float64-accumulating form, the rest against the float32 one), then the whole batch
    against the direct-kernel engine."""
    from hand3d_amd.utils.general import EvalUtil
    img = synth.make_batch(3000, 32, 320, 320)
    hs = synth.hand_sides(32)
    c0, c7, cp = net.engine.counter('conv_wino4_launches'), net.engine.counter('conv_wino7_launches'), net.engine.counter('conv_pw2_launches')
    o = net.engine.infer_full(img, hs, want_mask=True)
    assert net.engine.counter('conv_pw2_launches') - cp == 4, "the 1x1 head pairs did not run as one launch each (conv_pw2.hip)"
    assert net.engine.counter('conv_wino4_launches') - c0 >= 26, "the F(4x4,3x3) kernel did not take the 3x3 trunk layers"
    assert net.engine.counter('conv_wino7_launches') - c7 == 10, "the F(4x4,4x4) kernel did not take the ten 7x7 layers"
    net.engine.set_profiling(1)
    net.engine.infer_full(img, hs)
    prof = net.engine.profile()
    net.engine.set_profiling(0)
    w4 = [n for n, k, _, _, _ in prof if k.startswith(('conv_wino4_', 'conv_wino7_'))]          # (the 7x7 layers' F(4x4,4x4) form counts: the same family)
    assert len(w4) >= 36 and 'HandSegNet/conv3_2' in w4 and 'PoseNet2D/conv4_2' in w4 and 'PoseNet2D/conv6_3' in w4, sorted(set(k for _, k, _, _, _ in prof))
    assert not [k for _, k, _, _, _ in prof if k.startswith(('conv_wino_', 'conv_wino2_'))], "a trunk layer fell back to an F(2x2,3x3) kernel"
    ev = EvalUtil()
    worst = dict(scoremap=0.0, kpmap=0.0, coord3d=0.0)
    undecidable = []
    for i in range(32):
        taps = {}
        ref = N.inference(synth_weights, img[i:i + 1], hs[i:i + 1], True, acc=np.float64 if i in (0, 17) else np.float32, taps=taps)
        worst['scoremap'] = max(worst['scoremap'], float(np.abs(o['scoremap'][i:i + 1] - ref[0]).max()))
        # the mask stage is exact on the engine's own score map ...
        m = G.single_obj_scoremap(o['scoremap'][i:i + 1], early_exit=True)
        cen, _, best = G.calc_center_bb(m)
        assert np.array_equal(o['mask'][i], m[0, :, :, 0]), "image %d: mask growth differs from the oracle's on the same score map" % i
        assert np.array_equal(o['center'][i:i + 1], cen) and np.array_equal(o['scale'][i:i + 1], G.scale_from_crop_size(best, 256)), i
        # ... and equals the oracle's mask unless a pixel's fg probability sits within float32 rounding of 1/2 (random-weight
        # logits do that on about one image in thirty; which way it rounds then depends on the summation order of the kernel)
        if not np.array_equal(o['mask'][i], taps['hand_mask'][0, :, :, 0]):
            fg, _ = G.fg_and_detmap(ref[0])
            assert np.abs(fg - 0.5).min() < 1e-6, "image %d: hand mask differs although no pixel is near the rounding threshold" % i
            undecidable.append(i)
            continue
        assert np.array_equal(o['center'][i:i + 1], ref[3]) and np.array_equal(o['scale'][i:i + 1], ref[2]), i
        worst['kpmap'] = max(worst['kpmap'], float(np.abs(o['kpmap'][i:i + 1] - ref[4]).max()))
        worst['coord3d'] = max(worst['coord3d'], float(np.abs(o['coord3d'][i:i + 1] - ref[5]).max()))
        ev.feed(ref[5][0], np.ones(21), o['coord3d'][i])
    print("B=32 320x320, all 32 images vs oracle: worst %s, mean EPE %.3e, images with a knife-edge mask pixel: %s"
          % (worst, ev.get_measures(0.0, 0.05, 20)[0], undecidable))
    assert len(undecidable) <= 2
    assert worst['scoremap'] < TOL_HEATMAP and worst['kpmap'] < TOL_HEATMAP and worst['coord3d'] < TOL_KP3D
    net.engine.set_option('conv_impl', 'direct')
    try:
        od = net.engine.infer_full(img, hs)
    finally:
        net.engine.set_option('conv_impl', 'mfma')
    assert np.array_equal(o['center'], od['center']) and np.array_equal(o['scale'], od['scale'])
    assert np.abs(o['kpmap'] - od['kpmap']).max() < 1e-4 and np.abs(o['coord3d'] - od['coord3d']).max() < 1e-5


def test_full_pipeline_480x640_config_c5_shape(net, synth_weights):
    """BASELINE config 5 input shape (640x480 RGB stream), float32 path: 64 growth passes allowed, bigger mask."""
    img = synth.make_batch(1000, 3, 480, 640)
    hs = synth.hand_sides(3)
    o = net.engine.infer_full(img, hs, want_mask=True)
    for i in range(3):          # every image against the oracle (~20 s of CPU each at this size)
        taps = {}
        ref = N.inference(synth_weights, img[i:i + 1], hs[i:i + 1], True, acc=np.float64, taps=taps)
        assert np.array_equal(o['mask'][i], taps['hand_mask'][0, :, :, 0]), i
        assert np.array_equal(o['center'][i:i + 1], ref[3]) and np.array_equal(o['scale'][i:i + 1], ref[2]), i
        assert np.abs(o['scoremap'][i:i + 1] - ref[0]).max() < TOL_HEATMAP and np.abs(o['kpmap'][i:i + 1] - ref[4]).max() < TOL_HEATMAP
        assert np.abs(o['coord3d'][i:i + 1] - ref[5]).max() < TOL_KP3D
    assert np.isfinite(o['coord3d']).all() and np.isfinite(o['kpmap']).all()


def test_uint8_frontend_matches_float_path(net, synth_weights):
    """320x320 uint8 RHD-sized frames -> normalise + resize to 240x320 on device ->
    inference(); identical to feeding the oracle-preprocessed float image (eval_full.py:50 order)."""
    rng = np.random.default_rng(31)
    base = (synth.make_batch(400, 2, 320, 320) + 0.5) * 255.0
    u8 = np.clip(np.rint(base + rng.normal(0, 2, base.shape)), 0, 255).astype(np.uint8)
    hs = synth.hand_sides(2)
    pre = G.preprocess_u8(u8, 240, 320)
    assert np.array_equal(net.engine.preprocess_u8(u8, 240, 320), pre)
    a = net.inference_from_uint8(u8, hs, True, net_size=(240, 320))
    b = net.inference(pre, hs, True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    ref = N.inference(synth_weights, pre, hs, True, acc=np.float64)
    assert np.array_equal(a[3], ref[3]) and np.abs(a[5] - ref[5]).max() < TOL_KP3D


def test_golden_fixtures_gpu(net, synth_weights):
    """The committed golden vectors (tests/golden, made by scripts/make_golden.py from the oracle)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'e2e_240x320_seed0.npz'))
    img = synth.make_batch(0, 1, 240, 320)
    o = net.inference(img, np.array([[1.0, 0.0]], np.float32), True)
    assert np.array_equal(o[3], g['center']) and np.array_equal(o[2], g['scale_crop'])
    assert np.abs(o[5] - g['keypoint_coord3d']).max() < TOL_KP3D
    assert np.abs(o[4][0, ::8, ::8, :] - g['scoremap32']).max() < TOL_HEATMAP


def test_f16_trunk_kernel_conv_h16_vs_general_kernel(gpu_engine, synth_weights):
    """conv_h16.hip (the half-precision 3x3 trunk kernel: 1 / 2 / 4 cout blocks per wave, one to three workgroups per CU,
    pooled and plain epilogues through the LDS slabs) on shapes whose 16 x 16 tiles are ragged at every level (232 x 312 and
    its pooled sizes), forced onto every eligible layer ("h16_force"), against the f16-rounding oracle and against the general
    kernel (same MFMA, same packed weights: accumulation order only); the launch counter proves which kernel ran."""
    from hand3d_amd import ColorHandPose3DNetwork
    net16 = ColorHandPose3DNetwork(engine=gpu_engine)
    net16.init_from_dict(synth_weights, dtype='f16')
    try:
        img = synth.make_batch(510, 3, 232, 312)
        rs16, _ = N.handsegnet(synth_weights, img, acc=np.float64, f16=True)
        crop = synth.make_batch(110, 2, 256, 256)
        r16 = N.posenet2d(synth_weights, crop, acc=np.float64, f16=True)
        out = {}
        for impl in ('mfma', 'h16_force', 'h16'):
            gpu_engine.set_option('f16_impl', impl)
            n0 = gpu_engine.counter('conv_h16_launches')
            _, small = gpu_engine.handsegnet(img, want_small=True)
            sms = net16.inference_pose2d(crop)
            n1 = gpu_engine.counter('conv_h16_launches')
            out[impl] = (small, sms)
            # h16_force: 13 + 9 3x3 layers, and (round 4, option f16_k7k1) PoseNet2D's ten 7x7 stages + three 1x1 layers with >= 64 couts
            assert (n1 - n0 == 0) if impl == 'mfma' else (n1 - n0 >= (13 + 9 + 13 if impl == 'h16_force' else 1)), (impl, n1 - n0)
            assert np.abs(small - rs16).max() < 2e-3
            for a, b in zip(sms, r16):
                assert np.abs(a - b).max() < 2e-3
        for impl in ('h16_force', 'h16'):
            assert np.abs(out[impl][0] - out['mfma'][0]).max() < 5e-4
            for a, b in zip(out[impl][1], out['mfma'][1]):
                assert np.abs(a - b).max() < 5e-4
        # the 7x7 / 1x1 forms off: those 13 layers go back to the general kernel, same results to accumulation order
        gpu_engine.set_option('f16_impl', 'h16_force')
        gpu_engine.set_option('f16_k7k1', '0')
        n0 = gpu_engine.counter('conv_h16_launches')
        sms_k3 = net16.inference_pose2d(crop)
        n1 = gpu_engine.counter('conv_h16_launches')
        gpu_engine.set_option('f16_k7k1', '1')
        net16.inference_pose2d(crop)
        assert (gpu_engine.counter('conv_h16_launches') - n1) - (n1 - n0) == 13
        for a, b in zip(out['h16_force'][1], sms_k3):
            assert np.abs(a - b).max() < 5e-4
        # conv1_1 computed inside conv1_2's patch stage (default) == conv_first<F16> followed by conv_h16, bit for bit
        gpu_engine.set_option('f16_fuse12', '0')
        _, small_unfused = gpu_engine.handsegnet(img, want_small=True)
        sms_unfused = net16.inference_pose2d(crop)
        gpu_engine.set_option('f16_fuse12', '1')
        assert np.array_equal(out['h16_force'][0], small_unfused)
        for a, b in zip(out['h16_force'][1], sms_unfused):
            assert np.array_equal(a, b)
        # round 6: the fused block's two forms (filter ring, two workgroups per CU / conv1_2's filters resident in registers, one workgroup per
        # CU with the next patch built between the K-steps), each forced, on ragged tiles (232 = 14.5 tiles): the two-launch result bit for bit
        for form in ('ring', 'resident'):
            gpu_engine.set_option('f16_fuse12', form)
            n0 = gpu_engine.counter('conv_h16_first_resident_launches')
            _, small_f = gpu_engine.handsegnet(img, want_small=True)
            sms_f = net16.inference_pose2d(crop)
            assert gpu_engine.counter('conv_h16_first_resident_launches') - n0 == (2 if form == 'resident' else 0), form
            assert np.array_equal(small_f, small_unfused), form
            for a, b in zip(sms_f, sms_unfused):
                assert np.array_equal(a, b), form
    finally:
        gpu_engine.set_option('f16_fuse12', '1')
        gpu_engine.set_option('f16_k7k1', '1')
        gpu_engine.set_option('f16_impl', 'h16')
        gpu_engine.load_weight_dict(synth_weights)
        gpu_engine.finalize_weights(0)


def test_f16_config_c5_shape_properties(gpu_engine, synth_weights):
    """Config 5's per-GPU input size (480 x 640, half-precision trunks) through size-independent properties (the oracle comparison at this size
    and precision is tests/test_gpu_c5_fixture.py): the default path (conv_h16 + fused first block) is deterministic, its fused first block is
    bit-identical to the two-launch form, it agrees with the general half-precision kernel to accumulation order on the
    HandSegNet logits, and the whole pipeline returns finite keypoints with the same crop decisions on both kernels."""
    from hand3d_amd import ColorHandPose3DNetwork
    net16 = ColorHandPose3DNetwork(engine=gpu_engine)
    net16.init_from_dict(synth_weights, dtype='f16')
    try:
        B = 8
        img = synth.make_batch(1000, B, 480, 640)
        hs = synth.hand_sides(B)
        n0 = gpu_engine.counter('conv_h16_launches')
        _, small = gpu_engine.handsegnet(img, want_small=True)
        assert gpu_engine.counter('conv_h16_launches') - n0 >= 12          # the trunk really ran on conv_h16 (fused block = 1 launch)
        _, small2 = gpu_engine.handsegnet(img, want_small=True)
        assert np.array_equal(small, small2)
        gpu_engine.set_option('f16_fuse12', '0')
        _, small_unfused = gpu_engine.handsegnet(img, want_small=True)
        gpu_engine.set_option('f16_fuse12', '1')
        assert np.array_equal(small, small_unfused)
        # ... and that was the filter-resident form (9600 items >= 4 per CU); the ring form gives the same bits
        n0 = gpu_engine.counter('conv_h16_first_resident_launches')
        gpu_engine.handsegnet(img, want_small=True)
        assert gpu_engine.counter('conv_h16_first_resident_launches') - n0 == 1
        gpu_engine.set_option('f16_fuse12', 'ring')
        _, small_ring = gpu_engine.handsegnet(img, want_small=True)
        gpu_engine.set_option('f16_fuse12', '1')
        assert np.array_equal(small, small_ring)
        o_h16 = gpu_engine.infer_full(img, hs)
        gpu_engine.set_option('f16_impl', 'mfma')
        _, small_mfma = gpu_engine.handsegnet(img, want_small=True)
        o_mfma = gpu_engine.infer_full(img, hs)
        gpu_engine.set_option('f16_impl', 'h16')
        assert np.abs(small - small_mfma).max() < 1e-3
        assert np.isfinite(o_h16['coord3d']).all() and np.isfinite(o_mfma['coord3d']).all()
        same = [i for i in range(B) if np.array_equal(o_h16['center'][i], o_mfma['center'][i]) and np.array_equal(o_h16['scale'][i], o_mfma['scale'][i])]
        assert len(same) >= B // 2                 # random-weight logits sit near the threshold: a flipped edge pixel moves the crop
        for i in same:
            assert np.abs(o_h16['coord3d'][i] - o_mfma['coord3d'][i]).max() < 5e-3
    finally:
        gpu_engine.set_option('f16_fuse12', '1')
        gpu_engine.set_option('f16_impl', 'h16')
        gpu_engine.load_weight_dict(synth_weights)
        gpu_engine.finalize_weights(0)


def test_f16_trunks_config_c5(gpu_engine, synth_weights):
    """BASELINE config 5 precision: half-precision HandSegNet / PoseNet2D trunks (v_mfma_f32_32x32x16_f16, f32
    accumulate), float32 heads / mask stage / lifting.  Checked against the oracle with the same rounding points
    (tight) and against the float32 oracle (the config's own looser tolerance: 5e-3 heat-maps)."""
    from hand3d_amd import ColorHandPose3DNetwork
    net16 = ColorHandPose3DNetwork(engine=gpu_engine)
    net16.init_from_dict(synth_weights, dtype='f16')
    try:
        img = synth.make_batch(500, 2, 240, 320)
        large, small = gpu_engine.handsegnet(img, want_small=True)
        rs16, _ = N.handsegnet(synth_weights, img, acc=np.float64, f16=True)
        rs32, _ = N.handsegnet(synth_weights, img, acc=np.float64)
        e16, e32 = np.abs(small - rs16).max(), np.abs(small - rs32).max()
        print("f16 HandSegNet logits: vs f16-rounding oracle %.3e, vs f32 oracle %.3e" % (e16, e32))
        assert e16 < 2e-3 and e32 < 5e-3
        crop = synth.make_batch(100, 1, 256, 256)
        sms = net16.inference_pose2d(crop)
        r16 = N.posenet2d(synth_weights, crop, acc=np.float64, f16=True)
        r32 = N.posenet2d(synth_weights, crop, acc=np.float64)
        for a, b, c in zip(sms, r16, r32):
            print("f16 PoseNet2D heat-map: vs f16-rounding oracle %.3e, vs f32 oracle %.3e" % (np.abs(a - b).max(), np.abs(a - c).max()))
            assert np.abs(a - b).max() < 2e-3 and np.abs(a - c).max() < 5e-3
        # whole path: runs, finite, and close to the float32 engine wherever the mask (hence the crop) agrees
        hs = synth.hand_sides(2)
        o16 = gpu_engine.infer_full(img, hs, want_mask=True)
        gpu_engine.load_weight_dict(synth_weights)
        gpu_engine.finalize_weights(0)
        o32 = gpu_engine.infer_full(img, hs, want_mask=True)
        iou = (np.logical_and(o16['mask'] > 0, o32['mask'] > 0).sum() + 1e-9) / (np.logical_or(o16['mask'] > 0, o32['mask'] > 0).sum() + 1e-9)
        print("f16 vs f32 engine: mask IoU %.4f, |coord3d| diff %.3e" % (iou, np.abs(o16['coord3d'] - o32['coord3d']).max()))
        assert np.isfinite(o16['coord3d']).all() and iou > 0.9   # random-weight logits hover near the threshold: edge pixels flip
        for i in range(2):
            if np.array_equal(o16['center'][i], o32['center'][i]) and np.array_equal(o16['scale'][i], o32['scale'][i]):
                assert np.abs(o16['kpmap'][i] - o32['kpmap'][i]).max() < 5e-3
                assert np.abs(o16['coord3d'][i] - o32['coord3d'][i]).max() < 5e-3
    finally:
        gpu_engine.load_weight_dict(synth_weights)
        gpu_engine.finalize_weights(0)


def test_full_pipeline_arbitrary_image_sizes(net, synth_weights):
    """Any input size from 16 x 16 (VERDICT r2 "missing" 4): the reference resizes the H/8 x W/8 logits back to (s[1], s[2]) whatever they
    are (nets/ColorHandPose3DNetwork.py:165-166) and its VALID 2x2 max-pools floor odd extents.  250 x 330 (even, not multiples of 8; its
    pooled sizes 125 x 165 are odd) and 243 x 325 (odd) against the oracle; then a batch of 8 at 250 x 330 -- filled launches, i.e.
    conv_wino4.hip on ragged 4x4 tiles and the unfused pool after odd layers -- whose first images must agree with the B = 1 runs."""
    from oracle import nets as ON
    names = ['hand_scoremap', 'image_crop', 'scale_crop', 'center', 'keypoints_scoremap', 'keypoint_coord3d']
    tol = [1e-3, 1e-4, 1e-5, 1e-4, 1e-3, 1e-4]
    singles = {}
    for (H, W, seed) in ((250, 330, 41), (243, 325, 42)):
        img = synth.make_batch(seed, 1, H, W)
        hs = synth.hand_sides(1)
        out = net.inference(img, hs, True)
        ref = ON.inference(synth_weights, img, hs, True)
        for n, a, b, t in zip(names, out, ref, tol):
            assert a.shape == np.asarray(b).shape, (n, a.shape, np.asarray(b).shape)
            err = float(np.abs(a - b).max())
            print("%dx%d %-20s %.3e" % (H, W, n, err))
            assert err <= t, (H, W, n, err)
        singles[(H, W)] = (img, out)
    img0, out0 = singles[(250, 330)]
    img8 = np.concatenate([img0, synth.make_batch(43, 7, 250, 330)], 0)
    n0 = net.engine.counter('conv_wino4_launches')
    out8 = net.inference(img8, synth.hand_sides(8), True)
    assert net.engine.counter('conv_wino4_launches') > n0
    for n, a, b, t in zip(names, out8, out0, tol):
        err = float(np.abs(a[:1] - b).max())
        print("250x330 B=8 vs B=1 %-20s %.3e" % (n, err))
        assert err <= t, (n, err)


def test_errors_are_loud(gpu_engine):
    from hand3d_amd import ColorHandPose3DNetwork, Engine
    with pytest.raises(AssertionError):
        ColorHandPose3DNetwork(engine=gpu_engine).init(None, weight_files=['/nonexistent.pickle'])
    e2 = Engine(0)
    with pytest.raises(Exception):
        e2.infer_full(np.zeros((1, 240, 320, 3), np.float32), np.zeros((1, 2), np.float32))   # no weights
    with pytest.raises(AssertionError):
        gpu_engine.handsegnet(np.zeros((1, 12, 100, 3), np.float32))    # H < 16
    e2.close()


def test_stage_timing_and_native_rccl_world1(net, synth_weights):
    """hp3d_get_timing (per-stage GPU ms) and the engine's own RCCL entry points (hp3d_comm_* / hp3d_bcast_weights /
    hp3d_allgather) at world size 1 -- the only size a one-GPU box offers: librccl is dlopen'ed, a communicator is
    built from a fresh id, the weight broadcast is an identity and the all-gather returns the local shard."""
    img = synth.make_batch(77, 2, 240, 320)
    hs = synth.hand_sides(2)
    eng = net.engine
    before = eng.infer_full(img, hs)
    eng.set_profiling(1)
    eng.infer_full(img, hs)
    t = eng.get_timing()
    rows = eng.profile()
    eng.set_profiling(0)
    assert abs(t['total'] - sum(r[2] for r in rows)) < 1e-3 * max(t['total'], 1.0)
    assert t['HandSegNet'] > 0 and t['PoseNet2D'] > 0 and t['lifting'] > 0 and t['mask_crop'] > 0
    assert abs(t['HandSegNet'] + t['mask_crop'] + t['PoseNet2D'] + t['lifting'] - t['total']) < 1e-3 * t['total']
    uid = eng.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    eng.comm_init(0, 1, uid)
    try:
        with pytest.raises(AssertionError):
            eng.comm_init(0, 1, uid)                    # already initialised
        eng.bcast_weights(0)
        after = eng.infer_full(img, hs)
        for k in ('scoremap', 'kpmap', 'coord3d', 'center', 'scale'):
            assert np.array_equal(before[k], after[k]), k
        g = eng.allgather(before['coord3d'], 1)
        assert np.array_equal(g, before['coord3d'])
    finally:
        eng.comm_destroy()
    eng.comm_destroy()                                  # idempotent


def test_micro_batch_chunks_equal_separate_calls(net, synth_weights):
    """hp3d_set_option("micro_batch", N): whole-path calls run as chunks of N images -- bit-identical to making the
    calls chunk by chunk (an unsplit call can differ at rounding level: the small-batch conv plan depends on B), one
    profile per call."""
    img = synth.make_batch(4100, 5, 240, 320)
    hs = synth.hand_sides(5)
    eng = net.engine
    eng.set_option('micro_batch', '0')
    try:
        whole = eng.infer_full(img, hs, want_mask=True)
        sep = [eng.infer_full(img[a:b], hs[a:b], want_mask=True) for a, b in ((0, 2), (2, 4), (4, 5))]
        eng.set_option('micro_batch', '2')
        eng.set_profiling(1)
        parts = eng.infer_full(img, hs, want_mask=True)
        rows = eng.profile()
        eng.set_profiling(0)
    finally:
        eng.set_option('micro_batch', 'auto')
    assert sum(1 for r in rows if r[0] == 'mask_grow') == 3                 # chunks of 2 + 2 + 1 in ONE profile
    for k in ('mask', 'center', 'scale', 'scoremap', 'crop', 'kpmap', 'coord3d'):
        assert np.array_equal(np.concatenate([o[k] for o in sep], 0), parts[k]), k
    assert np.abs(whole['scoremap'] - parts['scoremap']).max() < 2e-5
    iou = (np.logical_and(whole['mask'] > 0, parts['mask'] > 0).sum() + 1.0) / (np.logical_or(whole['mask'] > 0, parts['mask'] > 0).sum() + 1.0)
    assert iou > 0.999
    with pytest.raises(AssertionError):
        eng.set_option('micro_batch', 'many')


def test_full_pipeline_odd_tile_grids_winograd(net, synth_weights):
    """248x328: the pyramid is 124x164 -> 62x82 -> 31x41, so the Winograd layers see odd tile grids (half tiles at the
    right / bottom edge, items that straddle images); 33 images = one chunk of 32 plus a single-image chunk."""
    B, H, W = 33, 248, 328
    img = synth.make_batch(5248, B, H, W)
    hs = synth.hand_sides(B)
    o = net.engine.infer_full(img, hs, want_mask=True)
    for i in (5, B - 1):
        taps = {}
        ref = N.inference(synth_weights, img[i:i + 1], hs[i:i + 1], True, acc=np.float64, taps=taps)
        assert np.abs(o['scoremap'][i:i + 1] - ref[0]).max() < TOL_HEATMAP
        rmask = taps['hand_mask'][0, :, :, 0]
        if not np.array_equal(o['mask'][i], rmask):
            # random weights put some logit pairs within rounding of the decision boundary: a pixel may then fall on
            # the other side (float32 Winograd vs float64-accumulating oracle); anything else is a bug
            diff = np.argwhere(o['mask'][i] != rmask)
            margin = np.abs(ref[0][0, :, :, 1] - ref[0][0, :, :, 0])
            assert len(diff) <= 3 and all(margin[y, x] < 1e-5 for y, x in diff), (len(diff), diff[:5])
            continue
        assert np.array_equal(o['center'][i:i + 1], ref[3]) and np.array_equal(o['scale'][i:i + 1], ref[2])
        assert np.abs(o['kpmap'][i:i + 1] - ref[4]).max() < TOL_HEATMAP
        assert np.abs(o['coord3d'][i:i + 1] - ref[5]).max() < TOL_KP3D


class _DevBuf(object):
    """device memory through the HIP runtime the engine itself links (no torch in this process)."""
    _hip = None

    def __init__(self, array_or_bytes):
        import ctypes as C
        if _DevBuf._hip is None:
            _DevBuf._hip = C.CDLL('libamdhip64.so')
        self.C, self.hip = C, _DevBuf._hip
        self.nbytes = array_or_bytes if isinstance(array_or_bytes, int) else array_or_bytes.nbytes
        self.ptr = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(self.ptr), C.c_size_t(self.nbytes)) == 0
        if not isinstance(array_or_bytes, int):
            self.upload(array_or_bytes)

    def upload(self, a):
        a = np.ascontiguousarray(a)
        assert self.hip.hipMemcpy(self.ptr, a.ctypes.data_as(self.C.c_void_p), self.C.c_size_t(a.nbytes), 1) == 0

    def download(self, shape):
        out = np.empty(shape, np.float32)
        assert self.hip.hipMemcpy(out.ctypes.data_as(self.C.c_void_p), self.ptr, self.C.c_size_t(out.nbytes), 2) == 0
        return out

    def free(self):
        self.hip.hipFree(self.ptr)


def test_hipgraph_replay_matches_plain_launches(net, synth_weights):
    """hp3d_set_option("graph", "1"): from the third identical device-pointer call on the launch sequence is replayed
    as one hipGraph -- same bytes out as the plain launches; new data behind the same pointers is picked up; a changed
    option drops the captured graph."""
    eng = net.engine
    crop = _DevBuf(synth.make_batch(9100, 2, 256, 256))
    outs = [_DevBuf(2 * 32 * 32 * 21 * 4) for _ in range(3)]
    img = _DevBuf(synth.make_batch(9300, 1, 240, 320))
    hs_h = synth.hand_sides(1)
    hs = _DevBuf(hs_h)
    coord = _DevBuf(63 * 4)

    def run():
        rc = eng.lib.hp3d_posenet2d_dev(eng.h, 2, 256, 256, crop.ptr, outs[0].ptr, outs[1].ptr, outs[2].ptr)
        assert rc == 0, eng.lib.hp3d_last_error(eng.h)
        eng.sync()
        return [o.download((2, 32, 32, 21)) for o in outs]

    try:
        plain = run()
        c0, r0 = eng.counter('graph_captures'), eng.counter('graph_replays')
        eng.set_option('graph', '1')
        for g in [run() for _ in range(4)]:                  # warm-up, capture (+ launch), replay, replay
            for a, b in zip(g, plain):
                assert np.array_equal(a, b)
        assert eng.counter('graph_captures') == c0 + 1 and eng.counter('graph_replays') == r0 + 3, \
            "the hipGraph path did not run (capture failed silently?)"
        eng.set_profiling(1)                                  # per-launch events need plain launches: no replay
        run()
        eng.set_profiling(0)
        assert eng.counter('graph_replays') == r0 + 3
        crop.upload(synth.make_batch(9200, 2, 256, 256))      # same pointers, new data
        eng.set_option('graph', '0')
        plain2 = run()
        eng.set_option('graph', '1')
        for _ in range(3):
            g2 = run()
        for a, b in zip(g2, plain2):
            assert np.array_equal(a, b)
        assert not np.array_equal(plain2[2], plain[2])
        ref = eng.infer_full(synth.make_batch(9300, 1, 240, 320), hs_h)['coord3d']
        for _ in range(4):
            eng.infer_full_dev(1, 240, 320, img.ptr.value, hs.ptr.value, coord3d=coord.ptr.value)
            eng.sync()
            assert np.array_equal(coord.download((1, 21, 3)), ref)
    finally:
        eng.set_option('graph', '0')
        for b in [crop, img, hs, coord] + outs:
            b.free()


@pytest.mark.parametrize("case", [(2, 32, 32, 160, 128), (8, 32, 32, 128, 128), (1, 20, 28, 64, 64), (33, 32, 32, 149, 128)],
                         ids=lambda c: "B%d_%dx%d_%d-%d" % c)
def test_conv7x7_on_winograd_kernel_vs_oracle(gpu_engine, case):
    """7x7 filters (PoseNet2D refinement units, nets/ColorHandPose3DNetwork.py:211-214) on conv_wino.hip: nine 3x3
    blocks of the zero-extended 9x9 filter, shifted windows, SAME padding 3; (33, 32x32, 149 -> 128) walks more items
    than CUs with the real layer's channel count."""
    B, H, W, Cin, Cout = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((7, 7, Cin, Cout)) / np.sqrt(49 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    r = T.leaky_relu(T.bias_add(T.conv2d_same(x, w, 1, acc=np.float64), b))
    gpu_engine.set_option('conv_impl', 'winograd')
    try:
        y = gpu_engine.conv2d(x, w, b, 1, True, False)
    finally:
        gpu_engine.set_option('conv_impl', 'mfma')
    assert np.abs(y - r).max() < 5e-5


@pytest.mark.parametrize("case", [(32, 32, 32, 128, 128), (4, 32, 32, 160, 128), (2, 30, 26, 64, 64), (1, 9, 11, 48, 64), (40, 32, 32, 149, 128), (3, 64, 48, 32, 192)],
                         ids=lambda c: "B%d_%dx%d_%d-%d" % c)
def test_conv7x7_as_four_4x4_blocks_f4x4_vs_oracle(gpu_engine, case):
    """conv_wino7.hip (round 5; PoseNet2D's ten 7x7 layers, nets/ColorHandPose3DNetwork.py:206-215): the 7x7 filter as the four 4x4-tap
    blocks of its zero-extended 8x8 form, Winograd F(4x4,4x4) each (49 planes; 169 plane products per 16 outputs after the structurally
    zero planes, 289 in the nine-3x3-block form), the transformed input of a tile block shared by the four blocks.  The bench shape's own
    layer (32 x 32x32 x 128 -> 128: exactly 256 work items), the 160-channel concat buffer, ragged sizes with partial tile blocks, a
    single small image, more items than CUs with the real first layer's 149 channels, three cout blocks on a non-square map.  Against the
    float64 oracle (1e-4 on unit-variance data, the gate of the F(4x4,3x3) kernel's own test) and against the nine-block form on the same input
    (the same order of magnitude: measured 0.5 ... 2.8 times its error, MI355X round 5); deterministic;
    the counter proves which kernel ran."""
    B, H, W, Cin, Cout = case
    rng = np.random.default_rng(sum(case) + 77)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((7, 7, Cin, Cout)) / np.sqrt(49 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    r = T.leaky_relu(T.bias_add(T.conv2d_same(x, w, 1, acc=np.float64), b))
    gpu_engine.set_option('wino7', '1')
    try:
        n0 = gpu_engine.counter('conv_wino7_launches')
        y = gpu_engine.conv2d(x, w, b, 1, True, False)
        assert gpu_engine.counter('conv_wino7_launches') == n0 + 1
        assert np.array_equal(y, gpu_engine.conv2d(x, w, b, 1, True, False)), "not deterministic"
    finally:
        gpu_engine.set_option('wino7', 'auto')
    gpu_engine.set_option('wino4', '1')
    try:
        y9 = gpu_engine.conv2d(x, w, b, 1, True, False)
    finally:
        gpu_engine.set_option('wino4', 'auto')
    e7, e9 = float(np.abs(y - r).max()), float(np.abs(y9 - r).max())
    print("7x7 as four 4x4 blocks %s: F(4x4,4x4) %.2e from the float64 oracle, the nine-block F(4x4,3x3) form %.2e" % (case, e7, e9))
    assert y.shape == r.shape and e7 < 1e-4 and e7 < 3 * e9 + 2e-5


@pytest.mark.parametrize("case", [(1, 32, 32, 128, 128, 0), (1, 32, 32, 160, 128, 0), (2, 32, 32, 160, 128, 3), (1, 30, 41, 149, 128, 0), (4, 32, 32, 128, 128, 4),
                                  (1, 9, 9, 32, 64, 2)], ids=lambda c: "B%d_%dx%d_%d-%d_ks%d" % c)
def test_conv7x7_f4x4_channel_split_vs_oracle(gpu_engine, case):
    """conv_wino7.hip's SPLITK form (round 5): launches that do not fill the chip -- a B = 1 PoseNet2D 7x7 layer is 8 work items on 256 CUs --
    split the 16-channel chunks over workgroups (raw 4x4 sums into [ksplit][B*Ho*Wo][Cout]) and add the slices in conv_splitk_reduce
    (+ bias, leaky-ReLU).  PoseNet2D's own B = 1 layers (128 and the 160-channel concat buffer: 8 and 10 chunks -> automatic split),
    forced uneven splits, the first stage's 149 real channels on a ragged map, one tiny tile block.  Against the float64 oracle and the
    unsplit launch; deterministic; the counter proves the split form ran."""
    B, H, W, Cin, Cout, ks = case
    rng = np.random.default_rng(sum(case) + 79)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((7, 7, Cin, Cout)) / np.sqrt(49 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    r = T.leaky_relu(T.bias_add(T.conv2d_same(x, w, 1, acc=np.float64), b))
    gpu_engine.set_option('wino7', '1')
    try:
        gpu_engine.set_option('wino_splitk', '0')
        y1 = gpu_engine.conv2d(x, w, b, 1, True, False)
        gpu_engine.set_option('wino_splitk', '1')
        gpu_engine.set_option('wino7_ksplit', str(ks) if ks else 'auto')
        n0 = gpu_engine.counter('conv_wino7_split_launches')
        y = gpu_engine.conv2d(x, w, b, 1, True, False)
        assert gpu_engine.counter('conv_wino7_split_launches') == n0 + 1
        assert np.array_equal(y, gpu_engine.conv2d(x, w, b, 1, True, False)), "not deterministic"
    finally:
        gpu_engine.set_option('wino7', 'auto')
        gpu_engine.set_option('wino7_ksplit', 'auto')
        gpu_engine.set_option('wino_splitk', '1')
    e, e1 = float(np.abs(y - r).max()), float(np.abs(y1 - r).max())
    print("7x7 F(4x4,4x4) channel split %s: %.2e from the float64 oracle, unsplit %.2e" % (case, e, e1))
    assert y.shape == r.shape and e < 1e-4 and e < 3 * e1 + 2e-5
    assert float(np.abs(y - y1).max()) < 1e-4


def test_1x1_head_pairs_as_one_launch_vs_two(gpu_engine, synth_weights):
    """conv_pw2.hip (round 5): conv6_1 + conv6_2 of HandSegNet and conv5_1 + conv5_2, conv6_6 + conv6_7, conv7_6 + conv7_7 of PoseNet2D
    (nets/ColorHandPose3DNetwork.py:160-161,202-203,213-214) as ONE launch each, the 512- / 128-channel intermediate in LDS.  B = 24 at
    200 x 264 (25 x 33 maps: a ragged last tile of 64 pixels) and a 256 x 256 crop batch: against the two-launch form of the same engine
    (float32 accumulation in another blocking: 1e-5) and against the float64 oracle on one image; the counter proves which path ran."""
    from hand3d_amd import ColorHandPose3DNetwork
    net = ColorHandPose3DNetwork(engine=gpu_engine)
    net.init_from_dict(synth_weights)
    img = synth.make_batch(4100, 24, 200, 264)
    crop = synth.make_batch(4200, 24, 256, 256)
    outs = {}
    try:
        for mode in ('0', '1'):
            gpu_engine.set_option('pw2', mode)
            n0 = gpu_engine.counter('conv_pw2_launches')
            _, small = gpu_engine.handsegnet(img, want_small=True)
            sms = net.inference_pose2d(crop)
            outs[mode] = (small, sms, gpu_engine.counter('conv_pw2_launches') - n0)
    finally:
        gpu_engine.set_option('pw2', '1')
    assert outs['0'][2] == 0 and outs['1'][2] == 4, (outs['0'][2], outs['1'][2])
    d_seg = float(np.abs(outs['1'][0] - outs['0'][0]).max())
    d_sm = max(float(np.abs(a - b).max()) for a, b in zip(outs['1'][1], outs['0'][1]))
    rs, _ = N.handsegnet(synth_weights, img[:1], acc=np.float64)
    ref = N.posenet2d(synth_weights, crop[:1], acc=np.float64)
    e_seg = float(np.abs(outs['1'][0][:1] - rs).max())
    e_sm = max(float(np.abs(a[:1] - b).max()) for a, b in zip(outs['1'][1], ref))
    print("1x1 head pairs, one launch vs two: logits %.2e, score maps %.2e; vs the float64 oracle: %.2e, %.2e" % (d_seg, d_sm, e_seg, e_sm))
    assert d_seg < 1e-5 and d_sm < 1e-5 and e_seg < 1e-4 and e_sm < 1e-4


def test_device_keypoints_equal_reference_host_functions(net, synth_weights):
    """hp3d_infer_full_kp / hp3d_infer_2d_kp / hp3d_detect_keypoints: detect_keypoints + trafo_coords
    (utils/general.py:331-357) on the device, bit-exact with the host functions applied to the returned 256x256 maps --
    ties and the up-sampling rounding hazard included (tests/test_emu_kernels.py::_near_tie_scoremaps)."""
    from hand3d_amd.utils import general as PG
    from test_emu_kernels import _near_tie_scoremaps
    img = synth.make_batch(700, 3, 240, 320)
    hs = synth.hand_sides(3)
    full = net.inference(img, hs, True)
    c3d, kp_hw, kp_crop, scale, center = net.inference_keypoints(img, hs, True)
    assert np.array_equal(c3d, full[5]) and np.array_equal(scale, full[2]) and np.array_equal(center, full[3])
    assert kp_hw.dtype == np.float64 and kp_crop.dtype == np.float64
    for i in range(3):
        kp = PG.detect_keypoints(full[4][i])
        assert np.array_equal(kp_crop[i], kp)
        assert np.array_equal(kp_hw[i], PG.trafo_coords(kp, full[3][i:i + 1], full[2][i:i + 1], 256))
    kp_hw2, kp_crop2, scale2, center2 = net.inference2d_keypoints(img)
    assert np.array_equal(kp_hw2, kp_hw) and np.array_equal(kp_crop2, kp_crop) and np.array_equal(scale2, scale)
    rng = np.random.default_rng(0)
    for trial in range(4):
        sm = _near_tie_scoremaps(trial, rng)
        up = net.engine.resize_bilinear(sm, 256, 256)
        ref = np.stack([PG.detect_keypoints(up[b]) for b in range(2)])
        assert np.array_equal(net.engine.detect_keypoints(sm), ref), trial
        assert np.array_equal(net.engine.argmax2d(up).astype(np.float64), ref), trial


def test_two_streams_equal_one_stream(net, synth_weights):
    """hp3d_set_option("streams", "2"): the two halves of a batch on two HIP streams (second arena, shared weights) give
    the single-stream results -- identical discrete decisions (mask, centre, scale, arg-max keypoints), floats equal to
    rounding (a half-batch may take the small-batch kernel plan: other summation order) -- for host- and device-pointer
    calls, odd batch sizes included."""
    def same(a, b, k):
        if a is None and b is None:
            return
        if k in ('mask', 'center', 'scale', 'kp_crop', 'kp_hw'):
            assert np.array_equal(a, b), k
        else:
            assert np.abs(a - b).max() < 2e-5, (k, float(np.abs(a - b).max()))
    eng = net.engine
    img = synth.make_batch(8100, 5, 240, 320)
    hs = synth.hand_sides(5)
    outs = ('scoremap', 'crop', 'scale', 'center', 'kpmap', 'coord3d', 'kp_crop', 'kp_hw')
    try:
        eng.set_option('streams', '1')
        one = eng.infer_full(img, hs, want_mask=True, outputs=outs)
        eng.set_option('streams', '2')
        two = eng.infer_full(img, hs, want_mask=True, outputs=outs)
        for k in one:
            same(one[k], two[k], k)
        u8 = np.clip(np.rint((synth.make_batch(8200, 3, 320, 320) + 0.5) * 255.0), 0, 255).astype(np.uint8)
        a = eng.infer_full_u8(u8, hs[:3])
        eng.set_option('streams', '1')
        b = eng.infer_full_u8(u8, hs[:3])
        for k in a:
            same(a[k], b[k], k)
        # device-pointer entry point: the child's stream must see inputs written on the parent's stream and vice versa
        eng.set_option('streams', '2')
        d_img, d_hs, d_c = eng.to_device(img), eng.to_device(hs), eng.dev_alloc(5 * 63 * 4)
        for _ in range(3):
            eng.infer_full_dev(5, 240, 320, int(d_img), int(d_hs), coord3d=int(d_c))
        eng.sync()
        assert np.array_equal(eng.to_host(d_c, (5, 21, 3)), two['coord3d'])
        for buf in (d_img, d_hs, d_c):
            buf.free()
    finally:
        eng.set_option('streams', 'auto')


def test_conv_first_batch_beyond_32bit_offsets(net, synth_weights):
    """conv1_1 at 480x640 with 28 images writes 2.2 GB: conv_first's launcher cuts the batch into image ranges that each stay
    inside the kernel's 32-bit offsets (config 5 runs 128 images per GPU; hp3d_handsegnet does not chunk the batch itself).
    Images on both sides of the cut equal their single-image runs (to rounding: B = 1 takes the small-batch plan)."""
    B = 28
    img = synth.make_batch(8300, 4, 480, 640)
    x = np.concatenate([img] * 7, 0)
    x[27] = synth.make_batch(8399, 1, 480, 640)[0]
    net.engine.set_profiling(1)
    large = net.engine.handsegnet(x)
    kernels = set(k for _, k, _, _, _ in net.engine.profile())
    net.engine.set_profiling(0)
    assert 'conv_first_3x3_c3' in kernels, kernels
    assert large.shape == (B, 480, 640, 2) and np.isfinite(large).all()
    for i in (0, 26, 27):
        li = net.engine.handsegnet(x[i:i + 1])
        assert np.abs(large[i:i + 1] - li).max() < 5e-5, i
    assert np.abs(large[27] - large[3]).max() > 1e-3          # image 27 is a different image: really its own result


def test_size_independent_properties_at_full_size(gpu_engine):
    """Properties that need no oracle, at BASELINE's largest single-image geometry (480x640):
    * the seeded mask growth is idempotent: growing again from the grown mask (as a saturated score map) returns it;
    * the legacy x8 bilinear up-sampling reproduces its source at every 8th pixel and is bounded by it;
    * a crop whose box is the 256x256 window around the centre at scale 1 is the identity on that window;
    * hp3d_detect_keypoints on a map with one planted peak per channel returns 8 x the peak position."""
    e = gpu_engine
    rng = np.random.default_rng(404)
    H, W = 480, 640
    # a random union of rectangles as foreground, strongest where the seed should land
    sm = np.zeros((1, H, W, 2), np.float32)
    sm[..., 0] = 2.0
    for _ in range(12):
        y0, x0 = int(rng.integers(100, 300)), int(rng.integers(150, 400))
        sm[0, y0:y0 + int(rng.integers(20, 90)), x0:x0 + int(rng.integers(20, 120)), 1] = 6.0
    sm[0, 200:210, 300:310, 1] = 9.0
    mask, center, size, scale, seed = e.mask_from_scoremap(sm)
    assert mask.sum() > 0 and mask[0, seed[0, 0], seed[0, 1]] == 1
    sm2 = np.zeros_like(sm)
    sm2[..., 0] = 1.0
    sm2[..., 1] = mask * 5.0
    sm2[0, seed[0, 0], seed[0, 1], 1] = 7.0                 # same seed
    mask2, center2, size2, _, seed2 = e.mask_from_scoremap(sm2)
    assert np.array_equal(mask2, mask) and np.array_equal(center2, center) and np.array_equal(size2, size)
    # up-sampling
    x = rng.standard_normal((2, 60, 80, 21)).astype(np.float32)
    up = e.resize_bilinear(x, 480, 640)
    assert np.array_equal(up[:, ::8, ::8, :], x)
    assert up.max() <= x.max() and up.min() >= x.min()
    # identity crop: scale 1, box = [c - 128, c + 128) on a 480x640 image samples whole pixels
    img = rng.uniform(-0.5, 0.5, (1, H, W, 3)).astype(np.float32)
    c = np.array([[240.0, 320.0]], np.float32)
    crop = e.crop_and_resize(img, c, np.array([1.0], np.float32), 256)
    y1 = np.float32(240 - 128) / np.float32(H) * np.float32(H - 1)          # crop_and_resize maps with (H-1), (W-1)
    assert crop.shape == (1, 256, 256, 3) and np.isfinite(crop).all()
    assert abs(float(crop[0, 0, 0, 0]) - float(img[0, 112, 192, 0])) < 0.6   # the box starts inside pixel (111.77, 191.7)
    # planted peaks
    smk = (rng.standard_normal((3, 32, 32, 21)) * 0.1).astype(np.float32)
    pos = rng.integers(0, 32, (3, 21, 2))
    for b in range(3):
        for ch in range(21):
            smk[b, pos[b, ch, 0], pos[b, ch, 1], ch] = 5.0 + ch
    assert np.array_equal(e.detect_keypoints(smk), (pos * 8).astype(np.int32))

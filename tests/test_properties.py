"""Property tests (hypothesis) on the CPU interpreter of the kernel sources: random geometries for the conv
kernel, random detection maps for the seeded growth, random boxes for crop-and-resize -- always bit-exact or
to rounding against the oracle."""
import numpy as np
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import general as G
from oracle import tf_ops as T

COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


@settings(max_examples=12, **COMMON)
@given(H=st.integers(3, 20), W=st.integers(3, 20), Cin=st.integers(1, 40), Cout=st.integers(1, 70),
       ks=st.sampled_from([(1, 1), (3, 1), (3, 2), (7, 1)]), B=st.integers(1, 2), seed=st.integers(0, 10 ** 6))
def test_conv_random_geometry(emu_engine, H, W, Cin, Cout, ks, B, seed):
    k, s = ks
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    r = T.leaky_relu(T.bias_add(T.conv2d_same(x, w, s, acc=np.float64), b))
    y = emu_engine.conv2d(x, w, b, s, True, False)
    assert y.shape == r.shape and np.abs(y - r).max() < 2e-5
    if k == 3 and s == 1 and H >= 2 and W >= 2:
        assert np.abs(emu_engine.conv2d(x, w, b, 1, True, True) - T.max_pool_2x2(r)).max() < 2e-5


@settings(max_examples=14, **COMMON)
@given(H=st.integers(2, 15), W=st.integers(2, 15), Cin=st.integers(1, 100), Cout=st.sampled_from([64, 128, 192, 256]),
       B=st.integers(1, 3), pool=st.booleans(), act=st.booleans(), seed=st.integers(0, 10 ** 6))
def test_winograd_random_geometry(emu_engine, H, W, Cin, Cout, B, pool, act, seed):
    """conv_wino.hip forced on, both item shapes (32 tiles x 128 couts / 64 tiles x 64 couts): odd sizes (half tiles at
    the right / bottom edge), items that straddle images, several items per workgroup (the interpreter reports 3 CUs),
    channel counts that need zero padding, with and without activation / fused pool."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    r = T.bias_add(T.conv2d_same(x, w, 1, acc=np.float64), b)
    if act:
        r = T.leaky_relu(r)
    if pool:
        r = T.max_pool_2x2(r)
    emu_engine.set_option('conv_impl', 'winograd')
    try:
        y = emu_engine.conv2d(x, w, b, 1, act, pool)
    finally:
        emu_engine.set_option('conv_impl', 'mfma')
    assert y.shape == r.shape and (y.size == 0 or np.abs(y - r).max() < 2e-5)


@settings(max_examples=15, **COMMON)
@given(H=st.integers(24, 72), W=st.integers(32, 96), density=st.floats(0.05, 0.95), seed=st.integers(0, 10 ** 6),
       blocky=st.booleans())
def test_mask_growth_random_maps(emu_engine, H, W, density, seed, blocky):
    """det = random (optionally blocky) binary map; fg peaks at a random pixel.  Mask, seed, bbox, centre,
    size and scale must be identical to the reference algorithm (full max(H,W)//10 passes, no early exit)."""
    rng = np.random.default_rng(seed)
    if blocky:
        small = rng.uniform(size=(H // 6 + 1, W // 6 + 1)) < density
        det = np.kron(small, np.ones((6, 6)))[:H, :W] > 0
    else:
        det = rng.uniform(size=(H, W)) < density
    logit = np.where(det, rng.uniform(0.1, 3.0, size=(H, W)), rng.uniform(-3.0, -0.1, size=(H, W))).astype(np.float32)
    sm = np.zeros((1, H, W, 2), np.float32)
    sm[0, :, :, 1] = logit
    mask, center, size, scale, seed_px = emu_engine.mask_from_scoremap(sm)
    fg, d = G.fg_and_detmap(sm)
    ref_seed = G.find_max_location(fg)
    obj, _ = G.grow_objectmap(d[0], ref_seed[0], early_exit=False)
    rc, _, rs = G.calc_center_bb(obj[None, :, :, None])
    assert np.array_equal(seed_px, ref_seed)
    assert np.array_equal(mask[0], obj)
    assert np.array_equal(center, rc) and np.array_equal(size, rs)
    assert np.array_equal(scale, G.scale_from_crop_size(rs))


@settings(max_examples=20, **COMMON)
@given(cy=st.floats(-20, 80), cx=st.floats(-20, 100), scale=st.floats(0.25, 5.0), seed=st.integers(0, 10 ** 6))
def test_crop_random_boxes(emu_engine, cy, cx, scale, seed):
    rng = np.random.default_rng(seed)
    img = rng.uniform(-.5, .5, (1, 48, 64, 3)).astype(np.float32)
    c = np.array([[cy, cx]], np.float32)
    s = np.array([scale], np.float32)
    assert np.array_equal(emu_engine.crop_and_resize(img, c, s, 32), G.crop_image_from_xy(img, c, 32, s))


@settings(max_examples=10, **COMMON)
@given(B=st.integers(1, 40), Cin=st.integers(1, 300), Cout=st.integers(1, 130), seed=st.integers(0, 10 ** 6))
def test_fc_random_shapes(emu_engine, B, Cin, Cout, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout)) / np.sqrt(Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    assert np.abs(emu_engine.fc(x, w, b, False) - T.fully_connected(x, w, b, np.float64)).max() < 1e-5


@settings(max_examples=10, **COMMON)
@given(H=st.integers(1, 22), W=st.integers(1, 22), Cin=st.integers(1, 100), Cout=st.sampled_from([64, 128, 192]), B=st.integers(1, 3),
       act=st.booleans(), ks=st.integers(0, 7), seed=st.integers(0, 10 ** 6))
def test_winograd_f4x4_4x4_random_geometry_and_channel_splits(emu_engine, H, W, Cin, Cout, B, act, ks, seed):
    """conv_wino7.hip forced on (round 5: a 7x7 filter as four F(4x4,4x4) blocks): maps smaller than one window, ragged tile blocks, channel
    counts that need zero padding to a 16-channel chunk, several images / items per workgroup -- unsplit, and with the chunks split over
    workgroups (`wino7_ksplit` = 0: the automatic choice; N: forced, clamped to the number of chunks) + the deterministic reduce."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((7, 7, Cin, Cout)) / np.sqrt(49 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    r = T.bias_add(T.conv2d_same(x, w, 1, acc=np.float64), b)
    if act:
        r = T.leaky_relu(r)
    emu_engine.set_option('wino7', '1')
    try:
        emu_engine.set_option('wino_splitk', '0')
        n0 = emu_engine.counter('conv_wino7_launches')
        y1 = emu_engine.conv2d(x, w, b, 1, act, False)
        assert emu_engine.counter('conv_wino7_launches') == n0 + 1
        emu_engine.set_option('wino_splitk', '1')
        emu_engine.set_option('wino7_ksplit', str(ks) if ks else 'auto')
        y = emu_engine.conv2d(x, w, b, 1, act, False)
    finally:
        emu_engine.set_option('wino7', 'auto')
        emu_engine.set_option('wino7_ksplit', 'auto')
        emu_engine.set_option('wino_splitk', '1')
    assert y.shape == r.shape and np.abs(y1 - r).max() < 1e-4 and np.abs(y - r).max() < 1e-4


@settings(max_examples=10, **COMMON)
@given(H=st.integers(16, 60), W=st.integers(16, 70), B=st.integers(1, 4), seed=st.integers(0, 10 ** 6))
def test_first_layer_kernel_random_geometry(emu_engine, H, W, B, seed):
    """conv_first.hip (conv1_1's shape through the per-op entry point): balanced runs of tiles over ragged tile grids and batches; the row
    walk of rounds 2-4 must give the same bits."""
    rng = np.random.default_rng(seed)
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
    assert np.abs(y - r).max() < 1e-5 and np.array_equal(y, y_rows)

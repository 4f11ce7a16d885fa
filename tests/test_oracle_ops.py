"""Pins the oracle's TF-1.3 op restatements against INDEPENDENT implementations (torch-CPU
conv/pool/linear; scalar-loop re-derivations of resize / crop / dilation) and checks the
reference's documented edge cases.  torch is used here only as a checker.
"""
import numpy as np
import pytest

from oracle import general as G
from oracle import tf_ops as T


def _torch_conv_same(x, w, stride):
    import torch
    import torch.nn.functional as F
    B, H, W, Cin = x.shape
    k = w.shape[0]
    _, pt, pb = T.same_pads(H, k, stride)
    _, pl, pr = T.same_pads(W, k, stride)
    xt = torch.from_numpy(x.astype(np.float64)).permute(0, 3, 1, 2)
    xt = F.pad(xt, (pl, pr, pt, pb))
    wt = torch.from_numpy(w.astype(np.float64)).permute(3, 2, 0, 1)
    return F.conv2d(xt, wt, stride=stride).permute(0, 2, 3, 1).numpy()


@pytest.mark.parametrize("H,W,Cin,Cout,k,s", [(12, 16, 5, 7, 3, 1), (8, 8, 4, 6, 3, 2), (9, 11, 3, 4, 3, 2),
                                               (10, 10, 6, 5, 7, 1), (6, 7, 8, 3, 1, 1)])
def test_conv2d_same_vs_torch(H, W, Cin, Cout, k, s):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, H, W, Cin)).astype(np.float32)
    w = rng.standard_normal((k, k, Cin, Cout)).astype(np.float32)
    assert np.abs(T.conv2d_same(x, w, s, acc=np.float64) - _torch_conv_same(x, w, s)).max() < 1e-5
    assert np.abs(T.conv2d_same(x, w, s, acc=np.float32) - _torch_conv_same(x, w, s)).max() < 1e-4


def test_same_padding_is_asymmetric_for_stride2():
    # App. B.1: s=2,k=3 on even sizes -> 0 before, 1 after; symmetric padding is a different function
    assert T.same_pads(32, 3, 2) == (16, 0, 1)
    assert T.same_pads(15, 3, 2) == (8, 1, 1)
    assert T.same_pads(240, 3, 1) == (240, 1, 1) and T.same_pads(32, 7, 1) == (32, 3, 3)
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(1)
    x = rng.standard_normal((1, 8, 8, 2)).astype(np.float32)
    w = rng.standard_normal((3, 3, 2, 2)).astype(np.float32)
    sym = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w).permute(3, 2, 0, 1), stride=2,
                   padding=1).permute(0, 2, 3, 1).numpy()
    assert np.abs(T.conv2d_same(x, w, 2) - sym).max() > 0.1


def test_pools_and_fc_vs_torch():
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 10, 14, 5)).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    assert np.array_equal(T.max_pool_2x2(x), F.max_pool2d(xt, 2).permute(0, 2, 3, 1).numpy())
    x = rng.standard_normal((1, 32, 48, 3)).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    assert np.abs(T.avg_pool_8x8(x) - F.avg_pool2d(xt, 8).permute(0, 2, 3, 1).numpy()).max() < 1e-6
    a = rng.standard_normal((3, 20)).astype(np.float32)
    w = rng.standard_normal((20, 7)).astype(np.float32)
    b = rng.standard_normal(7).astype(np.float32)
    assert np.abs(T.fully_connected(a, w, b) - (torch.from_numpy(a) @ torch.from_numpy(w) + torch.from_numpy(b)).numpy()).max() < 1e-5
    assert np.array_equal(T.leaky_relu(np.array([-2.0, 0.0, 3.0], np.float32)), np.array([-0.02, 0.0, 3.0], np.float32))


def _resize_scalar(x, oh, ow):
    """resize_bilinear_op.cc (TF 1.3) restated with scalar loops."""
    B, H, W, C = x.shape
    out = np.zeros((B, oh, ow, C), np.float32)
    hs, ws = np.float32(H) / np.float32(oh), np.float32(W) / np.float32(ow)
    for y in range(oh):
        iy = np.float32(y) * hs
        y0 = int(np.floor(iy)); y1 = min(y0 + 1, H - 1); ly = np.float32(iy - y0)
        for xx in range(ow):
            ix = np.float32(xx) * ws
            x0 = int(np.floor(ix)); x1 = min(x0 + 1, W - 1); lx = np.float32(ix - x0)
            top = x[:, y0, x0] + (x[:, y0, x1] - x[:, y0, x0]) * lx
            bot = x[:, y1, x0] + (x[:, y1, x1] - x[:, y1, x0]) * lx
            out[:, y, xx] = top + (bot - top) * ly
    return out


def test_resize_bilinear_legacy():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, 4, 5, 2)).astype(np.float32)
    assert np.array_equal(T.resize_bilinear_legacy(x, 32, 40), _resize_scalar(x, 32, 40))
    assert np.array_equal(T.resize_bilinear_legacy(x, 7, 3), _resize_scalar(x, 7, 3))
    assert np.array_equal(T.resize_bilinear_legacy(x, 4, 5), x)            # equal sizes => identity
    up = T.resize_bilinear_legacy(x, 32, 40)
    assert np.array_equal(up[:, ::8, ::8], x)                                # x8 reproduces source pixels (8a-12)
    assert np.array_equal(up[:, -1, :, :], up[:, -8, :, :])                  # no half-pixel: tail replicates last row


def _crop_scalar(image, boxes, ch, cw):
    """crop_and_resize_op.cc (TF 1.3) restated with scalar loops."""
    B, H, W, C = image.shape
    out = np.zeros((B, ch, cw, C), np.float32)
    f = np.float32
    for b in range(B):
        y1, x1, y2, x2 = [f(v) for v in boxes[b]]
        hs = (y2 - y1) * f(H - 1) / f(ch - 1)
        ws = (x2 - x1) * f(W - 1) / f(cw - 1)
        for y in range(ch):
            in_y = f(y1 * f(H - 1) + f(y) * hs)
            if in_y < 0 or in_y > H - 1:
                continue
            t, bo = int(np.floor(in_y)), int(np.ceil(in_y)); ly = f(in_y - t)
            for x in range(cw):
                in_x = f(x1 * f(W - 1) + f(x) * ws)
                if in_x < 0 or in_x > W - 1:
                    continue
                l, r = int(np.floor(in_x)), int(np.ceil(in_x)); lx = f(in_x - l)
                top = image[b, t, l] + (image[b, t, r] - image[b, t, l]) * lx
                bot = image[b, bo, l] + (image[b, bo, r] - image[b, bo, l]) * lx
                out[b, y, x] = top + (bot - top) * ly
    return out


def test_crop_and_resize_and_boxes():
    rng = np.random.default_rng(4)
    img = rng.uniform(-.5, .5, (3, 24, 32, 3)).astype(np.float32)
    center = np.array([[12, 16], [2, 30], [23.5, 0]], np.float32)
    scale = np.array([2.0, 5.0, 0.25], np.float32)
    boxes = G.crop_boxes(center, 32, scale, 24, 32)
    # box arithmetic of utils/general.py:182-190: normalised by H and W (not H-1, W-1)
    cs = np.float32(32) / scale
    assert np.allclose(boxes[:, 0] * 24, center[:, 0] - np.floor(cs / 2))
    assert np.allclose((boxes[:, 2] - boxes[:, 0]) * 24, cs)
    got = G.crop_image_from_xy(img, center, 32, scale)
    assert np.array_equal(got, _crop_scalar(img, boxes, 32, 32))
    assert (got[2] == 0).mean() > 0.5          # scale 0.25 window mostly outside: extrapolation value 0
    assert np.array_equal(G.scale_from_crop_size(np.array([[0.0], [100.0], [1e6]], np.float32)).ravel(),
                          np.array([5.0, 2.048, 0.25], np.float32))       # size 0 -> inf -> 5.0


def test_dilation_flat_vs_naive_and_grow_semantics():
    rng = np.random.default_rng(5)
    x = (rng.uniform(size=(13, 17)) > 0.9).astype(np.float32)
    filt = np.full((5, 5), 1 / 25.0, np.float32)
    assert np.array_equal(T.dilation2d_flat(x, 5, np.float32(1 / 25.0)), T.dilation2d_naive(x, filt))
    det = (rng.uniform(size=(40, 50)) > 0.4).astype(np.float32)
    seed = tuple(np.argwhere(det > 0)[7])
    a, _ = G.grow_objectmap(det, seed, naive=True, num_passes=3)
    b, _ = G.grow_objectmap(det, seed, naive=False, num_passes=3)
    assert np.array_equal(a, b)
    full, n1 = G.grow_objectmap(det, seed)                       # all max(H,W)//10 passes
    early, n2 = G.grow_objectmap(det, seed, early_exit=True)     # fix-point exit: exactly equivalent (8a-6)
    assert np.array_equal(full, early) and n2 <= n1 == 5
    assert full[seed] == 1 and np.all(full <= det)


def test_softmax_round_argmax_conventions():
    assert np.array_equal(T.round_half_even(np.array([0.5, 1.5, 2.5, 0.50001], np.float32)), [0, 2, 2, 1])
    sm = np.zeros((1, 4, 5, 2), np.float32)          # equal logits: fg = 0.5 -> round -> 0 (App. B.6)
    fg, det = G.fg_and_detmap(sm)
    assert np.all(fg == 0.5) and np.all(det == 0)
    assert G.find_max_location(fg).tolist() == [[0, 0]]           # first index on ties (App. B.7)
    sm[0, 2, 3, 1] = 40.0
    sm[0, 3, 1, 1] = 40.0                                          # both saturate to fg == 1.0
    fg, det = G.fg_and_detmap(sm)
    assert fg[0, 2, 3] == 1.0 and fg[0, 3, 1] == 1.0 and G.find_max_location(fg).tolist() == [[2, 3]]
    with pytest.raises(AssertionError):
        G.find_max_location(np.zeros((5, 4, 9), np.float32))     # reference asserts B < H and B < W (:210)


def test_calc_center_bb_and_fallbacks():
    m = np.zeros((2, 20, 30, 1), np.float32)
    m[0, 3:9, 10:25, 0] = 1
    c, bb, s = G.calc_center_bb(m)
    assert c[0].tolist() == [5.5, 17.0] and s[0, 0] == 14.0       # max - min, not +1
    assert bb[0].tolist() == [[3, 8], [10, 24]]
    assert c[1].tolist() == [160.0, 160.0] and s[1, 0] == 100.0   # empty mask (:311-320)
    old = G.EMPTY_REDUCE
    G.EMPTY_REDUCE = 'fltmax'
    try:
        c2, _, s2 = G.calc_center_bb(m)
    finally:
        G.EMPTY_REDUCE = old
    assert c2[1].tolist() == [0.0, 0.0] and s2[1, 0] == 100.0


def test_rotation_and_flip():
    from oracle import nets as N
    u = np.array([[0.3], [-1.2]], np.float32), np.array([[0.5], [0.1]], np.float32), np.array([[-0.7], [2.0]], np.float32)
    R = N.get_rot_mat(*u)
    assert np.abs(np.einsum('bij,bkj->bik', R, R) - np.eye(3)).max() < 1e-6
    assert np.abs(np.linalg.det(R) - 1).max() < 1e-6
    from scipy.spatial.transform import Rotation
    rv = np.stack([u[0][:, 0], u[1][:, 0], u[2][:, 0]], 1)
    assert np.abs(R - Rotation.from_rotvec(rv).as_matrix()).max() < 1e-6
    c = np.arange(2 * 21 * 3, dtype=np.float32).reshape(2, 21, 3)
    f = N.flip_right_hand(c, np.array([[1, 0], [0, 1]], np.float32))
    assert np.array_equal(f[0], c[0]) and np.array_equal(f[1, :, 2], -c[1, :, 2]) and np.array_equal(f[1, :, :2], c[1, :, :2])


def test_evalutil_and_host_helpers_match_oracle():
    from hand3d_amd.utils import general as HG
    rng = np.random.default_rng(6)
    a, b = G.EvalUtil(), HG.EvalUtil()
    for _ in range(7):
        gt, pr = rng.standard_normal((21, 3)), rng.standard_normal((21, 3))
        vis = rng.uniform(size=21) > 0.2
        a.feed(gt, vis, pr)
        b.feed(gt, vis, pr)
    ma, mb = a.get_measures(0.0, 3.0, 20), b.get_measures(0.0, 3.0, 20)
    for x, y in zip(ma, mb):
        assert np.allclose(x, y)
    assert 0 <= ma[2] <= 1 and ma[0] > 0
    sm = rng.standard_normal((64, 64, 21))
    assert np.array_equal(G.detect_keypoints(sm), HG.detect_keypoints(sm))
    kp = G.detect_keypoints(sm)
    assert np.allclose(HG.trafo_coords(kp, np.array([10., 20.]), 2.0, 256), (kp - 128) / 2.0 + [10., 20.])
    assert abs(HG.calc_auc(np.linspace(0, 1, 5), np.ones(5)) - 1.0) < 1e-12

"""VERDICT r2 item 4c, with numbers: PoseNet2D with EVERY 3x3 layer (and optionally the 7x7 layers, as nine 3x3 blocks) computed by
Winograd F(4x4,3x3) in float32 -- transforms, plane products and their channel sums all in float32, as a kernel would -- against the
float64-accumulating oracle, end to end: heat-map error, arg-max keypoints, and the 3-D keypoints after the (unchanged) lifting
nets.  Same for F(2x2,3x3) (what conv_wino.hip computes) as the yard-stick.  CPU only, a few minutes.

    python scripts/micro/wino_f43_posenet.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hand3d_amd import synth  # noqa: E402
from oracle import general as G  # noqa: E402
from oracle import nets as N  # noqa: E402
from oracle import tf_ops as T  # noqa: E402

F = np.float32
W43 = dict(BT=np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64),
           G=np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64),
           AT=np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64), m=4)
W23 = dict(BT=np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64),
           G=np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64),
           AT=np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64), m=2)


def wino3x3(x, w, tf):
    """x [B,H,W,C] f32 (H, W multiples of m), w [3,3,C,K] f32 -> SAME 3x3 convolution by F(m x m, 3x3), float32 throughout."""
    m = tf['m']
    a = m + 2
    B, H, Wd, C = x.shape
    K = w.shape[3]
    BT, Gm, AT = tf['BT'].astype(F), tf['G'], tf['AT'].astype(F)
    xp = np.zeros((B, H + 2, Wd + 2, C), F)
    xp[:, 1:-1, 1:-1] = x
    ty, tx = H // m, Wd // m
    s = xp.strides
    tiles = np.lib.stride_tricks.as_strided(xp, (B, ty, tx, a, a, C), (s[0], s[1] * m, s[2] * m, s[1], s[2], s[3]))
    V = np.einsum('pg,ntxghc->ntxphc', BT, tiles, optimize=True).astype(F)
    V = np.einsum('qh,ntxphc->ntxpqc', BT, V, optimize=True).astype(F)            # [B,ty,tx,a,a,C]
    U = np.einsum('pr,rsck,qs->pqck', Gm, w.astype(np.float64), Gm).astype(F)     # filter transform done once, rounded to f32
    Mp = np.einsum('ntxpqc,pqck->ntxpqk', V, U, optimize=True).astype(F)          # float32 products and sums (BLAS sgemm)
    Y = np.einsum('ip,ntxpqk->ntxiqk', AT, Mp, optimize=True).astype(F)
    Y = np.einsum('jq,ntxiqk->ntxijk', AT, Y, optimize=True).astype(F)
    return Y.transpose(0, 1, 3, 2, 4, 5).reshape(B, H, Wd, K)


def conv7_as_blocks(x, w, tf):
    """7x7 SAME as nine 3x3 blocks of the filter zero-extended to 9x9 (conv_wino.hip's decomposition), each block by F(m, 3)."""
    B, H, Wd, C = x.shape
    w9 = np.zeros((9, 9, C, w.shape[3]), F)
    w9[:7, :7] = w
    out = np.zeros((B, H, Wd, w.shape[3]), F)
    xp = np.zeros((B, H + 16, Wd + 16, C), F)
    xp[:, 8:-8, 8:-8] = x
    for i in range(3):
        for j in range(3):
            dy, dx = 3 * i - 2, 3 * j - 2          # block (i, j) = a 3x3 convolution of the input shifted by (3i - 2, 3j - 2)
            # (4 extra pixels all round, cropped afterwards: the 3x3 SAME padding must see the shifted input's real neighbours)
            o = wino3x3(np.ascontiguousarray(xp[:, 4 + dy:12 + dy + H, 4 + dx:12 + dx + Wd]), w9[3 * i:3 * i + 3, 3 * j:3 * j + 3], tf)
            out += o[:, 4:4 + H, 4:4 + Wd]
    return out


_Base = N._Ops


class Ops(_Base):
    def __init__(self, weights, scope, tf3, tf7):
        _Base.__init__(self, weights, scope, np.float64)
        self.tf3, self.tf7 = tf3, tf7

    def conv(self, x, name, kernel_size, stride, out_chan):
        w = self.w['%s/%s/weights' % (self.scope, name)]
        b = self.w['%s/%s/biases' % (self.scope, name)]
        m = (self.tf3 or {}).get('m', 1)
        if kernel_size == 3 and stride == 1 and self.tf3 is not None and x.shape[3] >= 64 and x.shape[1] % m == 0:
            return T.bias_add(wino3x3(np.asarray(x, F), w, self.tf3), b)
        if kernel_size == 7 and self.tf7 is not None:
            return T.bias_add(conv7_as_blocks(np.asarray(x, F), w, self.tf7), b)
        return T.bias_add(T.conv2d_same(x, w, stride, acc=np.float64), b)


def posenet(weights, crop, tf3, tf7):
    saved = N._Ops
    N._Ops = lambda w, scope, acc=np.float32, taps=None, f16=False: Ops(w, scope, tf3, tf7)
    try:
        return N.posenet2d(weights, crop)
    finally:
        N._Ops = saved


def main():
    w = synth.make_weights()
    imgs = synth.make_batch(1000, 2, 320, 320)
    hs = synth.hand_sides(2)
    ref = N.inference(w, imgs, hs, True, acc=np.float64)
    crop = ref[1]
    sm_ref = N.posenet2d(w, crop, acc=np.float64)[-1]
    c3_ref = N.pose3d(w, sm_ref, hs, acc=np.float64)[0]
    kp_ref = [G.detect_keypoints(T.resize_bilinear_legacy(sm_ref[i:i + 1], 256, 256)[0]) for i in range(2)]
    print('%-44s %12s %12s %12s %s' % ('PoseNet2D variant (float32)', 'heat-map max', 'heat-map rms', 'coord3d max', 'arg-max keypoints changed'))
    for name, tf3, tf7 in (('F(2x2,3x3) on 3x3 layers (conv_wino.hip)', W23, None), ('F(2x2,3x3) on 3x3 and 7x7 layers', W23, W23),
                           ('F(4x4,3x3) on 3x3 layers', W43, None), ('F(4x4,3x3) on 3x3 and 7x7 layers', W43, W43)):
        sm = posenet(w, crop, tf3, tf7)[-1]
        c3 = N.pose3d(w, sm, hs, acc=np.float64)[0]
        kp = [G.detect_keypoints(T.resize_bilinear_legacy(sm[i:i + 1], 256, 256)[0]) for i in range(2)]
        changed = sum(int((a != b).any(axis=1).sum()) for a, b in zip(kp, kp_ref))
        e = np.abs(sm - sm_ref)
        print('%-44s %12.3e %12.3e %12.3e %d of 42' % (name, e.max(), np.sqrt((e ** 2).mean()), np.abs(c3 - c3_ref).max(), changed))
    print('gates: heat-maps 1e-3, 3-D keypoints 1e-4 (north star); heat-map scale: max |value| %.2f' % np.abs(sm_ref).max())


if __name__ == '__main__':
    main()

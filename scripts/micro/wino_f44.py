"""Round 5: the 7x7 layers as Winograd F(4x4,4x4) over the filter's four 4x4-tap blocks (conv_wino7.hip) -- where the transform matrices
come from and what the float32 rounding costs, before a line of kernel code was written.

1. Cook-Toom construction of F(m, r) over a point set (exact rationals): A^T, G, B^T as conv_wino7.hip / wino7_pack_weights hold them,
   checked against the direct correlation.
2. One 7x7 layer (128 -> 8 channels, 16 x 16 map, unit-variance input, fan-in-scaled filter) in float32 -- transforms, plane products,
   channel sums -- against float64: the nine-3x3-block form of conv_wino4.hip, F(4x4,4x4) over several point sets, the direct form.
3. One 3x3 layer (256 channels) by F(2x2,3x3), F(4x4,3x3) and F(6x6,3x3): the larger tile was priced too (21 % fewer products, twice
   the F(4x4) error -- but its 8x8 input transform per (tile, channel) is shared by only 64 couts: more VALU than it saves MFMAs).
4. End to end: PoseNet2D + lifting with every 3x3 layer on F(4x4,3x3) and the 7x7 layers on either form (wino_f43_posenet.py's machinery).

    python scripts/micro/wino_f44.py            (CPU only, a few minutes; output in profiles/r05_wino7_numerics.md)
"""
import os
import sys
from fractions import Fraction as Fr

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def cook_toom(m, r, pts):
    """F(m, r) over the n - 1 = m + r - 2 finite points `pts` plus infinity.  y = A^T [(G g) * (B^T d)] with y[i] = sum_k g[k] d[i + k].
    Returns (A^T [m, n], G [n, r], B^T [n, n]) as object arrays of Fractions."""
    n = m + r - 1
    assert len(pts) == n - 1
    pts = [Fr(p) for p in pts]
    AT = [[Fr(0)] * n for _ in range(m)]
    for i in range(m):
        for j in range(n - 1):
            AT[i][j] = pts[j] ** i
    AT[m - 1][n - 1] = Fr(1)
    G = [[Fr(0)] * r for _ in range(n)]
    for j in range(n - 1):
        N = Fr(1)
        for k in range(n - 1):
            if k != j:
                N *= pts[j] - pts[k]
        for k in range(r):
            G[j][k] = pts[j] ** k / N
    G[n - 1][r - 1] = Fr(1)

    def polymul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                out[i + j] += x * y
        return out
    M = [Fr(1)]
    for k in range(n - 1):
        M = polymul(M, [-pts[k], Fr(1)])
    BT = [[Fr(0)] * n for _ in range(n)]
    for j in range(n - 1):
        q = [Fr(1)]
        for k in range(n - 1):
            if k != j:
                q = polymul(q, [-pts[k], Fr(1)])
        for i, c in enumerate(q):
            BT[j][i] = c
    for i, c in enumerate(M):
        BT[n - 1][i] = c
    return np.array(AT, dtype=object), np.array(G, dtype=object), np.array(BT, dtype=object)


def tofl(a):
    return np.array([[float(x) for x in row] for row in a], dtype=np.float64)


P44 = [0, 1, -1, 2, -2, Fr(1, 2)]          # conv_wino7.hip's points (+ infinity)


def conv_ref(x, g):
    """x [H,W,C], g [k,k,C,O]: SAME correlation in float64."""
    H, W, C = x.shape
    k = g.shape[0]
    pad = k // 2
    xp = np.zeros((H + 2 * pad, W + 2 * pad, C))
    xp[pad:pad + H, pad:pad + W] = x
    y = np.zeros((H, W, g.shape[3]))
    for u in range(k):
        for v in range(k):
            y += np.tensordot(xp[u:u + H, v:v + W], g[u, v], axes=([2], [0]))
    return y


def wino_blocks(x, g, m, r, pts, dt=np.float32):
    """k x k SAME correlation (k = g.shape[0]) as ceil(k / r)^2 blocks of r x r taps of the zero-extended filter, F(m x m, r x r) each, all
    blocks accumulating into the same planes.  Transforms, products and the channel sums (four channels per addition, like an MFMA
    chain) in `dt`; U = G g G^T in float64, rounded once."""
    AT, G, BT = [tofl(a) for a in cook_toom(m, r, pts)]
    n = m + r - 1
    k = g.shape[0]
    nb = (k + r - 1) // r
    pad = k // 2
    H, W, C = x.shape
    O = g.shape[3]
    ge = np.zeros((nb * r, nb * r, C, O))
    ge[:k, :k] = g
    xp = np.zeros((H + 2 * pad + 2 * n + r * nb, W + 2 * pad + 2 * n + r * nb, C), dt)
    xp[pad:pad + H, pad:pad + W] = x.astype(dt)
    ATd, BTd = AT.astype(dt), BT.astype(dt)
    y = np.zeros((H, W, O), dt)
    for ty in range(0, H, m):
        for tx in range(0, W, m):
            M = np.zeros((n, n, O), dt)
            for i in range(nb):
                for j in range(nb):
                    d = xp[ty + r * i: ty + r * i + n, tx + r * j: tx + r * j + n]
                    V = np.einsum('ab,bcx->acx', BTd, d).astype(dt)
                    V = np.einsum('acx,dc->adx', V, BTd).astype(dt)
                    U = np.einsum('ar,rsco,bs->abco', G, ge[r * i:r * i + r, r * j:r * j + r], G).astype(dt)
                    for c0 in range(0, C, 4):
                        M = (M + np.einsum('abc,abco->abo', V[:, :, c0:c0 + 4], U[:, :, c0:c0 + 4]).astype(dt)).astype(dt)
            Y = np.einsum('ia,abo->ibo', ATd, M).astype(dt)
            Y = np.einsum('ibo,jb->ijo', Y, ATd).astype(dt)
            y[ty:ty + m, tx:tx + m] = Y[:min(m, H - ty), :min(m, W - tx)]
    return y


def one_layer_tables():
    rng = np.random.default_rng(1)
    H = W = 16
    C, O = 128, 8
    x = rng.standard_normal((H, W, C))
    x = np.maximum(x, 0.01 * x)
    g = rng.standard_normal((7, 7, C, O)) * np.sqrt(2.0 / (49 * C))
    ref = conv_ref(x, g)
    err = lambda y: (np.abs(y - ref).max(), np.sqrt(((y - ref) ** 2).mean()))
    print('one 7x7 layer, %d -> %d channels, %d x %d, output rms %.2f; float32 against float64 (max / rms):' % (C, O, H, W, ref.std()))
    print('  %-58s %.3e / %.3e' % ('nine 3x3 blocks, F(4x4,3x3) each (conv_wino4.hip)', *err(wino_blocks(x, g, 4, 3, [0, 1, -1, 2, -2]))))
    for pts in (P44, [0, 1, -1, 2, -2, Fr(-1, 2)], [0, 1, -1, Fr(1, 2), Fr(-1, 2), 2], [0, 1, -1, Fr(1, 2), Fr(-1, 2), Fr(3, 2)], [0, 1, -1, 2, -2, Fr(1, 3)]):
        print('  %-58s %.3e / %.3e' % ('four 4x4 blocks, F(4x4,4x4) over {%s, inf}' % ', '.join(str(p) for p in pts), *err(wino_blocks(x, g, 4, 4, pts))))
    xd, gd = x.astype(np.float32), g.astype(np.float32)
    xp = np.zeros((H + 6, W + 6, C), np.float32)
    xp[3:3 + H, 3:3 + W] = xd
    yd = np.zeros((H, W, O), np.float32)
    for u in range(7):
        for v in range(7):
            for c0 in range(0, C, 4):
                yd = (yd + np.tensordot(xp[u:u + H, v:v + W, c0:c0 + 4], gd[u, v, c0:c0 + 4], axes=([2], [0])).astype(np.float32)).astype(np.float32)
    print('  %-58s %.3e / %.3e' % ('direct form', *err(yd)))
    H = W = 24
    C = 256
    x = rng.standard_normal((H, W, C))
    x = np.maximum(x, 0.01 * x)
    g = rng.standard_normal((3, 3, C, O)) * np.sqrt(2.0 / (9 * C))
    ref = conv_ref(x, g)
    print('one 3x3 layer, %d -> %d channels, %d x %d:' % (C, O, H, W))
    for m, pts in ((2, [0, 1, -1]), (4, [0, 1, -1, 2, -2]), (6, [0, 1, -1, 2, -2, Fr(1, 2), Fr(-1, 2)])):
        print('  %-58s %.3e / %.3e' % ('F(%dx%d,3x3)' % (m, m), *err(wino_blocks(x, g, m, 3, pts))))


def end_to_end():
    import wino_f43_posenet as P
    from hand3d_amd import synth
    from oracle import general as G
    from oracle import nets as N
    from oracle import tf_ops as T
    F = np.float32
    AT, Gm, BT = [tofl(a) for a in cook_toom(4, 4, P44)]

    def conv7_as_four_blocks(x, w):
        """[B,H,W,C] x [7,7,C,K]: F(4x4,4x4) per 4x4-tap block of the 8x8 extension, float32 throughout (vectorised over tiles)."""
        B, H, Wd, C = x.shape
        K = w.shape[3]
        w8 = np.zeros((8, 8, C, K))
        w8[:7, :7] = w
        ty, tx = (H + 3) // 4, (Wd + 3) // 4
        xp = np.zeros((B, 4 * ty + 16, 4 * tx + 16, C), F)
        xp[:, 3:3 + H, 3:3 + Wd] = x
        M = np.zeros((B, ty, tx, 7, 7, K), F)
        s = xp.strides
        for i in range(2):
            for j in range(2):
                tiles = np.lib.stride_tricks.as_strided(xp[:, 4 * i:, 4 * j:], (B, ty, tx, 7, 7, C), (s[0], s[1] * 4, s[2] * 4, s[1], s[2], s[3]))
                V = np.einsum('pg,ntxghc->ntxphc', BT.astype(F), tiles, optimize=True).astype(F)
                V = np.einsum('qh,ntxphc->ntxpqc', BT.astype(F), V, optimize=True).astype(F)
                U = np.einsum('pr,rsck,qs->pqck', Gm, w8[4 * i:4 * i + 4, 4 * j:4 * j + 4], Gm).astype(F)
                M = (M + np.einsum('ntxpqc,pqck->ntxpqk', V, U, optimize=True).astype(F)).astype(F)
        Y = np.einsum('ip,ntxpqk->ntxiqk', AT.astype(F), M, optimize=True).astype(F)
        Y = np.einsum('jq,ntxiqk->ntxijk', AT.astype(F), Y, optimize=True).astype(F)
        return Y.transpose(0, 1, 3, 2, 4, 5).reshape(B, 4 * ty, 4 * tx, K)[:, :H, :Wd]

    class Ops44(P.Ops):
        def conv(self, x, name, kernel_size, stride, out_chan):
            if kernel_size == 7:
                w = self.w['%s/%s/weights' % (self.scope, name)]
                b = self.w['%s/%s/biases' % (self.scope, name)]
                return T.bias_add(conv7_as_four_blocks(np.asarray(x, F), w), b)
            return P.Ops.conv(self, x, name, kernel_size, stride, out_chan)

    w = synth.make_weights()
    imgs = synth.make_batch(1000, 2, 320, 320)
    hs = synth.hand_sides(2)
    ref = N.inference(w, imgs, hs, True, acc=np.float64)
    crop = ref[1]
    sm_ref = N.posenet2d(w, crop, acc=np.float64)[-1]
    c3_ref = N.pose3d(w, sm_ref, hs, acc=np.float64)[0]
    kp_ref = [G.detect_keypoints(T.resize_bilinear_legacy(sm_ref[i:i + 1], 256, 256)[0]) for i in range(2)]
    print('PoseNet2D + lifting in float32 against the float64 oracle, two 320x320 frames (3x3 layers on F(4x4,3x3) in both rows):')
    print('  %-50s %12s %12s %12s %s' % ('7x7 layers', 'heat-map max', 'heat-map rms', 'coord3d max', 'arg-max keypoints changed'))
    for name, cls in (('nine 3x3 blocks, F(4x4,3x3) (conv_wino4.hip)', None), ('four 4x4 blocks, F(4x4,4x4) (conv_wino7.hip)', Ops44)):
        saved = N._Ops
        if cls is None:
            N._Ops = lambda ww, scope, acc=np.float32, taps=None, f16=False: P.Ops(ww, scope, P.W43, P.W43)
        else:
            N._Ops = lambda ww, scope, acc=np.float32, taps=None, f16=False: cls(ww, scope, P.W43, None)
        try:
            sm = N.posenet2d(w, crop)[-1]
        finally:
            N._Ops = saved
        c3 = N.pose3d(w, sm, hs, acc=np.float64)[0]
        kp = [G.detect_keypoints(T.resize_bilinear_legacy(sm[i:i + 1], 256, 256)[0]) for i in range(2)]
        changed = sum(int((a != b).any(axis=1).sum()) for a, b in zip(kp, kp_ref))
        e = np.abs(sm - sm_ref)
        print('  %-50s %12.3e %12.3e %12.3e %d of 42' % (name, e.max(), np.sqrt((e ** 2).mean()), np.abs(c3 - c3_ref).max(), changed))
    print('  gates: heat-maps 1e-3, 3-D keypoints 1e-4 (north star); heat-map scale: max |value| %.2f' % np.abs(sm_ref).max())


def main():
    AT, G, BT = cook_toom(4, 4, P44)
    for name, Mx in (('A^T', AT), ('G', G), ('B^T', BT)):
        print('F(4,4) over {0, 1, -1, 2, -2, 1/2, inf}: %s =' % name)
        for row in Mx:
            print('   [' + ', '.join('%5s' % str(v) for v in row) + ']')
    rng = np.random.default_rng(0)
    g, d = rng.standard_normal(4), rng.standard_normal(7)
    y = tofl(AT) @ ((tofl(G) @ g) * (tofl(BT) @ d))
    print('check against the direct correlation (float64): %.1e' % np.abs(y - np.array([sum(g[k] * d[i + k] for k in range(4)) for i in range(4)])).max())
    one_layer_tables()
    end_to_end()


if __name__ == '__main__':
    main()

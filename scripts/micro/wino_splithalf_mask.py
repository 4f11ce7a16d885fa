"""Round 5, VERDICT r4 item 7, second criterion: would split-operand products move the hand mask more than today's kernel does?

HandSegNet's score map feeds a THRESHOLD (the hand mask): round 4 measured, on the GPU and against the float64 oracle over 256 synthetic
320 x 320 images, a det pixel flipped on 49 images with the direct float32 kernel, 20 with F(2x2,3x3), 41 with F(4x4,3x3)
(`profiles/r04_mask_flip_vs_oracle.md`); VERDICT's bar for a split-operand kernel: no more than today's.

This script runs HandSegNet on the CPU with every 3x3 layer by Winograd F(4x4,3x3) exactly as `wino_splithalf.py` emulates it -- float32
transforms, the plane products either in float32 (today's kernel; not bit-identical to the GPU's summation order, statistically the same
thing) or from three bfloat16 pieces per operand and the six products of weight >= 2^-16, or two float16 pieces / three products with the
per-plane scaling -- and counts the images whose det map (round-half-even of the softmax foreground, utils/general.py:240-242) differs from
the float64-accumulating oracle's in at least one pixel, and the largest oracle margin |l1 - l0| at such a pixel.

    python scripts/micro/wino_splithalf_mask.py [n_images=64] [height=320] [width=320]      (~10 s per image on 16 threads)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import wino_splithalf as WS                                     # noqa: E402
from hand3d_amd import synth                                    # noqa: E402
from oracle import general as G, nets as N, tf_ops as T         # noqa: E402

F32 = np.float32


def wino_conv(x, w, mode, scale):
    """x [H, W, Cin] float32, w [3, 3, Cin, Cout]: F(4x4,3x3) as the kernel computes it, products per `mode` (wino_splithalf.plane_products)."""
    H, W, Cin = x.shape
    Cout = w.shape[3]
    U = np.einsum('ai,ijco,bj->abco', WS.G, w.astype(np.float64), WS.G).reshape(36, Cin, Cout)
    su = np.ones(36)
    if scale:
        su = 2.0 ** (3 - np.floor(np.log2(np.abs(U).reshape(36, -1).max(1))))
    U = (U * su[:, None, None]).astype(F32)
    xp = np.zeros((H + 2, W + 2, Cin), F32)
    xp[1:-1, 1:-1] = x
    ty, tx = H // 4, W // 4
    # windows [ty, tx, 6, 6, Cin] as a strided view, transforms in float32
    s0, s1, s2 = xp.strides
    d = np.lib.stride_tricks.as_strided(xp, (ty, tx, 6, 6, Cin), (4 * s0, 4 * s1, s0, s1, s2))
    BT = WS.BT.astype(F32)
    V = np.einsum('ai,yxijc->yxajc', BT, d, optimize=True).astype(F32)
    V = np.einsum('yxajc,bj->abyxc', V, BT, optimize=True).astype(F32).reshape(36, ty * tx, Cin)
    sv = np.ones(36)
    if scale:
        sv = 2.0 ** (3 - np.floor(np.log2(np.maximum(np.abs(V).reshape(36, -1).max(1), 1e-30))))
        V = (V * sv[:, None, None].astype(F32)).astype(F32)
    M = WS.plane_products(U, V, mode)
    M = (M / (su * sv)[:, None, None].astype(F32)).astype(F32).reshape(6, 6, ty * tx, Cout)
    AT = WS.AT.astype(F32)
    Y = np.einsum('ia,abto->ibto', AT, M, optimize=True).astype(F32)
    Y = np.einsum('ibto,jb->tijo', Y, AT, optimize=True).astype(F32)
    return Y.reshape(ty, tx, 4, 4, Cout).transpose(0, 2, 1, 3, 4).reshape(H, W, Cout)


class WinoOps(N._Ops):
    """oracle/nets.py's layer helpers with the 3x3 / stride-1 convolutions on the Winograd emulation"""
    def __init__(self, weights, mode, scale):
        super().__init__(weights, 'HandSegNet', F32, None, False)
        self.mode, self.scale = mode, scale

    def conv(self, x, name, kernel_size, stride, out_chan):
        w = self.w['%s/%s/weights' % (self.scope, name)]
        b = self.w['%s/%s/biases' % (self.scope, name)]
        if kernel_size == 3 and stride == 1 and x.shape[3] % 16 == 0 and x.shape[1] % 4 == 0 and x.shape[2] % 4 == 0:
            y = np.stack([wino_conv(x[i], w, self.mode, self.scale) for i in range(x.shape[0])])
            return T.bias_add(y, b)
        return super().conv(x, name, kernel_size, stride, out_chan)


def handsegnet_logits(ops, image):
    """nets/ColorHandPose3DNetwork.py:131-168 (as oracle/nets.py:handsegnet walks it) -> full-size logits [B, H, W, 2]"""
    x = np.asarray(image, dtype=F32)
    for block_id, (n, c, pool) in enumerate(zip([2, 2, 4, 4], [64, 128, 256, 512], [True, True, True, False]), 1):
        for layer_id in range(n):
            x = ops.conv_relu(x, 'conv%d_%d' % (block_id, layer_id + 1), 3, 1, c)
        if pool:
            x = ops.max_pool(x, 'pool%d' % block_id)
    x = ops.conv_relu(x, 'conv5_1', 3, 1, 512)
    enc = ops.conv_relu(x, 'conv5_2', 3, 1, 128)
    x = ops.conv_relu(enc, 'conv6_1', 1, 1, 512)
    sm = ops.conv_lin(x, 'conv6_2', 1, 1, 2)
    return T.resize_bilinear_legacy(sm, image.shape[1], image.shape[2])


MODES = [('f32', False, 'float32 products (today)'), (('bf16', 3, 6), False, 'bf16 x3, 6 products'), (('f16', 2, 3), True, 'f16 x2, 3 products, per-plane scaling')]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 320
    weights = synth.make_weights()
    stats = {m[2]: dict(images=0, pixels=0, margin=0.0, err=0.0) for m in MODES}
    t0 = time.time()
    for i in range(n):
        img = synth.make_batch(5000 + i, 1, H, W)
        ref = N.handsegnet(weights, img, acc=np.float64)[1][0]
        _, det_ref = G.fg_and_detmap(ref)
        margin = np.abs(ref[..., 1].astype(np.float64) - ref[..., 0].astype(np.float64))
        for mode, scale, name in MODES:
            lg = handsegnet_logits(WinoOps(weights, mode, scale), img)
            _, det = G.fg_and_detmap(lg)
            diff = det != det_ref
            s = stats[name]
            s['images'] += int(diff.any())
            s['pixels'] += int(diff.sum())
            if diff.any():
                s['margin'] = max(s['margin'], float(margin[diff].max()))
            s['err'] = max(s['err'], float(np.abs(lg - ref).max()))
        if (i + 1) % 8 == 0 or i + 1 == n:
            print('# %d images, %.0f s: ' % (i + 1, time.time() - t0) + '; '.join('%s %d' % (k, v['images']) for k, v in stats.items()), flush=True)
    print()
    print('| 3x3 layers of HandSegNet by F(4x4,3x3) with | images with a det pixel != float64 oracle | det pixels != oracle (of %d) | largest oracle margin at such a pixel | worst score-map error |' % (n * H * W))
    print('|---|---|---|---|---|')
    for mode, scale, name in MODES:
        s = stats[name]
        print('| %s | %d / %d | %d | %.2e | %.2e |' % (name, s['images'], n, s['pixels'], s['margin'], s['err']))


if __name__ == '__main__':
    main()

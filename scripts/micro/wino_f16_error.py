"""Numerical experiment (CPU, NumPy): would a half-precision Winograd F(2x2,3x3) trunk pass config 5's parity gate?

V = B^T d B from the half activations (float32 arithmetic, rounded to half), U = G g G^T from the float32 weights (rounded to
half once), products accumulated in float32 (what v_mfma_f32_32x32x16_f16 does), A^T M A + bias + leaky-ReLU in float32, halves
out.  Compared on the synthetic-weight HandSegNet / PoseNet2D against the f16-rounding oracle (the 2e-3 gate of
tests/test_gpu_parity.py::test_f16_trunks_config_c5) and against the float32 oracle (5e-3).

  python scripts/micro/wino_f16_error.py
"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import nets as N, tf_ops as T          # noqa: E402
from hand3d_amd import synth                        # noqa: E402

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float32)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)
h = lambda a: np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def wino16(x, w):
    B, H, W, C = x.shape
    Hp, Wp = (H + 1) // 2 * 2, (W + 1) // 2 * 2
    xp = np.zeros((B, Hp + 2, Wp + 2, C), np.float32)
    xp[:, 1:1 + H, 1:1 + W] = x
    ty, tx = Hp // 2, Wp // 2
    d = np.empty((B, ty, tx, 4, 4, C), np.float32)
    for i in range(4):
        for j in range(4):
            d[:, :, :, i, j] = xp[:, i:i + 2 * ty:2, j:j + 2 * tx:2]
    V = h(np.einsum('ik,btxklc,jl->btxijc', BT, d, BT))
    U = h(np.einsum('ik,klco,jl->ijco', G, w.astype(np.float32), G))
    M = np.einsum('btxijc,ijco->btxijo', V, U, optimize=True).astype(np.float32)
    Y = np.einsum('ik,btxklo,jl->btxijo', AT, M, AT)
    out = Y.transpose(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return out[:, :H, :W]


class WinoOps(N._Ops):
    def conv(self, x, name, kernel_size, stride, out_chan):
        w = self.w['%s/%s/weights' % (self.scope, name)]
        b = self.w['%s/%s/biases' % (self.scope, name)]
        if kernel_size == 3 and stride == 1 and x.shape[3] >= 64 and self.f16:
            return T.bias_add(wino16(h(x), w), b)
        return super().conv(x, name, kernel_size, stride, out_chan)


def main():
    weights = synth.make_weights()
    img = synth.make_batch(7, 1, 64, 96)
    rs16, _ = N.handsegnet(weights, img, acc=np.float64, f16=True)
    rs32, _ = N.handsegnet(weights, img, acc=np.float64)
    orig = N._Ops
    N._Ops = WinoOps
    try:
        rw, _ = N.handsegnet(weights, img, acc=np.float64, f16=True)
        crop = synth.make_batch(9, 1, 64, 64)
        pw = N.posenet2d(weights, crop, acc=np.float64, f16=True)
    finally:
        N._Ops = orig
    p16 = N.posenet2d(weights, crop, acc=np.float64, f16=True)
    p32 = N.posenet2d(weights, crop, acc=np.float64)
    print('HandSegNet logits: wino16 vs f16 oracle %.3e (gate 2e-3), vs f32 oracle %.3e (gate 5e-3); direct f16 vs f32 %.3e; range %.2f'
          % (np.abs(rw - rs16).max(), np.abs(rw - rs32).max(), np.abs(rs16 - rs32).max(), np.abs(rs32).max()))
    for a, b, c in zip(pw, p16, p32):
        print('PoseNet2D heat-map: wino16 vs f16 oracle %.3e, vs f32 oracle %.3e; direct f16 vs f32 %.3e; range %.3f'
              % (np.abs(a - b).max(), np.abs(a - c).max(), np.abs(b - c).max(), np.abs(c).max()))


if __name__ == '__main__':
    main()

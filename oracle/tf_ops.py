"""TensorFlow-1.3 op semantics restated in NumPy (oracle; test infrastructure only).

Every function names the reference call site it stands in for and the TF kernel
behaviour it restates.  All tensors are float32 NHWC unless
stated.  `acc` selects the accumulation dtype of the contractions: np.float32
(what TF does) or np.float64 (a tighter yard-stick for the GPU parity tests).
"""
import numpy as np

F32 = np.float32

# How conv2d_same / fully_connected contract in float32: 'numpy' = tap-by-tap GEMMs on NumPy's BLAS (default; the form
# every parity test uses), 'torch' = torch.nn.functional.conv2d on the CPU (oneDNN, all cores) with the same explicit TF
# SAME padding -- the "CPU restatement baseline" of SURVEY.md 8d that bench.py times (cpu_baseline, kind "port").
CONV_BACKEND = 'numpy'


# --------------------------------------------------------------------------- conv
def same_pads(in_size, k, stride):
    """TF 'SAME' padding (App. B.1): out=ceil(in/stride); total=max((out-1)*s+k-in,0);
    before=total//2, the remainder goes AFTER (asymmetric for s=2,k=3 on even sizes)."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    before = total // 2
    return out, before, total - before


def conv2d_same(x, w, stride=1, acc=np.float32):
    """tf.nn.conv2d(x, w, [1,s,s,1], 'SAME') -- utils/general.py:46.
    x [B,H,W,Cin], w HWIO [k,k,Cin,Cout]; cross-correlation, zero padding."""
    x = np.asarray(x)
    w = np.asarray(w)
    B, H, W, Cin = x.shape
    k, k2, Cin2, Cout = w.shape
    assert k == k2 and Cin2 == Cin
    Ho, pt, pb = same_pads(H, k, stride)
    Wo, pl, pr = same_pads(W, k, stride)
    if CONV_BACKEND == 'torch' and acc == np.float32:
        import torch
        import torch.nn.functional as Fn
        with torch.no_grad():
            xt = Fn.pad(torch.from_numpy(np.ascontiguousarray(x, dtype=F32)).permute(0, 3, 1, 2), (pl, pr, pt, pb))
            wt = torch.from_numpy(np.ascontiguousarray(w, dtype=F32)).permute(3, 2, 0, 1).contiguous()
            y = Fn.conv2d(xt, wt, stride=stride)
            return np.ascontiguousarray(y.permute(0, 2, 3, 1).numpy())
    xp = np.zeros((B, H + pt + pb, W + pl + pr, Cin), dtype=acc)
    xp[:, pt:pt + H, pl:pl + W, :] = x
    wm = w.astype(acc)
    out = np.zeros((B, Ho, Wo, Cout), dtype=acc)
    # tap-by-tap GEMM: out += X_shifted[B*Ho*Wo, Cin] @ W[r,s][Cin, Cout]
    for r in range(k):
        for s in range(k):
            xs = xp[:, r:r + (Ho - 1) * stride + 1:stride, s:s + (Wo - 1) * stride + 1:stride, :]
            out += (xs.reshape(-1, Cin) @ wm[r, s]).reshape(B, Ho, Wo, Cout)
    return out.astype(F32)


def bias_add(x, b):
    """tf.nn.bias_add -- utils/general.py:51."""
    return (x + np.asarray(b, dtype=F32)).astype(F32)


def leaky_relu(x, slope=0.01):
    """tf.maximum(t, 0.01*t) -- utils/general.py:30-33 (also after FC: :132-136)."""
    x = np.asarray(x, dtype=F32)
    return np.maximum(x, F32(slope) * x)


def max_pool_2x2(x):
    """tf.nn.max_pool ksize 2, stride 2, VALID -- utils/general.py:61-65 (floor(in/2))."""
    B, H, W, C = x.shape
    Ho, Wo = H // 2, W // 2
    v = x[:, :Ho * 2, :Wo * 2, :].reshape(B, Ho, 2, Wo, 2, C)
    return v.max(axis=(2, 4))


def avg_pool_8x8(x):
    """tf.nn.avg_pool ksize 8, stride 8, SAME -- nets/PosePriorNetwork.py:61.
    SAME pads only when in%8 != 0; padded cells are excluded from the mean (TF divides
    by the number of valid cells)."""
    B, H, W, C = x.shape
    Ho, pt, _ = same_pads(H, 8, 8)
    Wo, pl, _ = same_pads(W, 8, 8)
    out = np.zeros((B, Ho, Wo, C), dtype=F32)
    for oy in range(Ho):
        y0, y1 = max(oy * 8 - pt, 0), min(oy * 8 - pt + 8, H)
        for ox in range(Wo):
            x0, x1 = max(ox * 8 - pl, 0), min(ox * 8 - pl + 8, W)
            win = x[:, y0:y1, x0:x1, :].astype(np.float32)
            out[:, oy, ox, :] = win.sum(axis=(1, 2), dtype=np.float32) / F32((y1 - y0) * (x1 - x0))
    return out


def fully_connected(x, w, b, acc=np.float32):
    """tf.matmul(x, W) + b -- utils/general.py:129.  W is [in, out]."""
    return ((x.astype(acc) @ w.astype(acc)).astype(F32) + np.asarray(b, F32)).astype(F32)


# ------------------------------------------------------------------------- resize
def resize_bilinear_legacy(x, out_h, out_w):
    """tf.image.resize_images(x, (h,w)) = ResizeBilinear, align_corners=False, TF 1.3
    (no half-pixel centres) -- nets/ColorHandPose3DNetwork.py:97,128,166.  App. B.3:
    scale=in/out (float32); src=dst*scale; lo=floor(src); hi=min(lo+1,in-1); t=src-lo;
    lerp x in the top and bottom rows first, then y; equal sizes => identity."""
    x = np.asarray(x, dtype=F32)
    B, H, W, C = x.shape
    if (out_h, out_w) == (H, W):
        return x.copy()

    def weights(out_n, in_n):
        scale = F32(in_n) / F32(out_n)
        src = np.arange(out_n, dtype=F32) * scale
        lo = np.floor(src).astype(np.int64)
        hi = np.minimum(lo + 1, in_n - 1)
        t = (src - lo.astype(F32)).astype(F32)
        return lo, hi, t

    ylo, yhi, ty = weights(out_h, H)
    xlo, xhi, tx = weights(out_w, W)
    tx_ = tx[None, None, :, None]
    ty_ = ty[None, :, None, None]
    tl = x[:, ylo][:, :, xlo]
    tr = x[:, ylo][:, :, xhi]
    bl = x[:, yhi][:, :, xlo]
    br = x[:, yhi][:, :, xhi]
    top = tl + (tr - tl) * tx_
    bot = bl + (br - bl) * tx_
    return (top + (bot - top) * ty_).astype(F32)


def crop_and_resize(image, boxes, crop_h, crop_w, extrapolation_value=0.0):
    """tf.image.crop_and_resize(image, boxes, range(B), [crop_h, crop_w]) bilinear --
    utils/general.py:195.  App. B.4 (TF 1.3 crop_and_resize_op.cc):
      in_y = y1*(H-1) + y*(y2-y1)*(H-1)/(crop_h-1); out of [0,H-1] => whole row = extrapolation;
      top=floor, bottom=ceil, lerp=in_y-top; same in x with (W-1);
      val = top + (bottom-top)*y_lerp with top/bottom lerped in x first."""
    image = np.asarray(image, dtype=F32)
    B, H, W, C = image.shape
    out = np.full((B, crop_h, crop_w, C), F32(extrapolation_value), dtype=F32)
    for b in range(B):
        y1, x1, y2, x2 = (F32(v) for v in boxes[b])
        hs = (y2 - y1) * F32(H - 1) / F32(crop_h - 1) if crop_h > 1 else F32(0)
        ws = (x2 - x1) * F32(W - 1) / F32(crop_w - 1) if crop_w > 1 else F32(0)
        ys = np.arange(crop_h, dtype=F32)
        xs = np.arange(crop_w, dtype=F32)
        in_y = (y1 * F32(H - 1) + ys * hs) if crop_h > 1 else np.full(1, F32(0.5) * (y1 + y2) * F32(H - 1), F32)
        in_x = (x1 * F32(W - 1) + xs * ws) if crop_w > 1 else np.full(1, F32(0.5) * (x1 + x2) * F32(W - 1), F32)
        in_y = in_y.astype(F32)
        in_x = in_x.astype(F32)
        vy = (in_y >= 0) & (in_y <= F32(H - 1))
        vx = (in_x >= 0) & (in_x <= F32(W - 1))
        iy = np.where(vy, in_y, 0).astype(F32)
        ix = np.where(vx, in_x, 0).astype(F32)
        ty0 = np.floor(iy).astype(np.int64)
        ty1 = np.ceil(iy).astype(np.int64)
        ly = (iy - ty0.astype(F32)).astype(F32)[:, None, None]
        tx0 = np.floor(ix).astype(np.int64)
        tx1 = np.ceil(ix).astype(np.int64)
        lx = (ix - tx0.astype(F32)).astype(F32)[None, :, None]
        img = image[b]
        tl = img[ty0][:, tx0]
        tr = img[ty0][:, tx1]
        bl = img[ty1][:, tx0]
        br = img[ty1][:, tx1]
        top = tl + (tr - tl) * lx
        bot = bl + (br - bl) * lx
        val = (top + (bot - top) * ly).astype(F32)
        ok = vy[:, None, None] & vx[None, :, None]
        out[b] = np.where(ok, val, F32(extrapolation_value))
    return out


# ---------------------------------------------------------------- softmax & friends
def exp_f32_cr(x):
    """float32 exp, correctly rounded (computed in float64 then rounded).  TF 1.3 uses
    Eigen's float32 polynomial exp (<=1 ulp, not reproducible bit-for-bit outside
    Eigen); the oracle and the HIP kernel both use the correctly rounded value so that
    the arg-max tie pattern of the saturated fg map is well defined on both sides."""
    return np.exp(np.asarray(x, dtype=np.float64)).astype(F32)


def softmax_last(x):
    """tf.nn.softmax over the last axis -- utils/general.py:240: exp(x-max)/sum(exp(x-max))
    in float32."""
    x = np.asarray(x, dtype=F32)
    m = x.max(axis=-1, keepdims=True)
    e = exp_f32_cr(x - m)
    s = e.sum(axis=-1, keepdims=True, dtype=F32)
    return (e / s).astype(F32)


def round_half_even(x):
    """tf.round (App. B.6) -- np.round is round-half-to-even as well."""
    return np.round(np.asarray(x, dtype=F32)).astype(F32)


def argmax_first_flat(x2d):
    """tf.argmax on [B, H*W] (utils/general.py:220-221): first maximal index, row-major."""
    return np.argmax(x2d, axis=1).astype(np.int32)


def dilation2d_flat(x, k, filt_value):
    """tf.nn.dilation2d(x, filter=const(filt_value)[k,k,1], strides 1, rates 1, 'SAME') on
    a single-channel [H,W] map -- utils/general.py:259.  App. B.8:
    out[y,x] = max over the in-image part of the k x k window of (in + filt_value); padded
    positions are ignored.  A flat filter makes the 2-D max separable (exact)."""
    x = np.asarray(x, dtype=F32)
    H, W = x.shape
    r = k // 2
    neg = F32(-np.inf)
    xp = np.full((H, W + 2 * r), neg, dtype=F32)
    xp[:, r:r + W] = x
    hmax = xp[:, 0:W].copy()
    for d in range(1, k):
        np.maximum(hmax, xp[:, d:d + W], out=hmax)
    yp = np.full((H + 2 * r, W), neg, dtype=F32)
    yp[r:r + H, :] = hmax
    vmax = yp[0:H, :].copy()
    for d in range(1, k):
        np.maximum(vmax, yp[d:d + H, :], out=vmax)
    return (vmax + F32(filt_value)).astype(F32)


def dilation2d_naive(x, filt):
    """Scalar-loop statement of TF's dilation2d (general filter) used by the tests to pin
    `dilation2d_flat`; small inputs only."""
    x = np.asarray(x, dtype=F32)
    filt = np.asarray(filt, dtype=F32)
    H, W = x.shape
    kh, kw = filt.shape
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    out = np.full((H, W), -np.inf, dtype=F32)
    for y in range(H):
        for xx in range(W):
            best = F32(-np.inf)
            for dy in range(kh):
                iy = y + dy - pt
                if iy < 0 or iy >= H:
                    continue
                for dx in range(kw):
                    ix = xx + dx - pl
                    if ix < 0 or ix >= W:
                        continue
                    v = F32(x[iy, ix] + filt[dy, dx])
                    if v > best:
                        best = v
            out[y, xx] = best
    return out

"""Round 5, VERDICT r4 item 7 (exploratory, price-first): the Winograd-domain products of conv_wino4.hip on the 16-bit matrix pipe
(16x the f32 rate on MI355X) with the operands split into 16-bit pieces, float32 accumulation, transforms unchanged in float32.

What is emulated (CPU only, numpy): one 3x3 layer by Winograd F(4x4,3x3) exactly as the kernel does it -- U = G g G^T in double rounded
once, V = B^T d B in float32, the 36 plane products summed over the channels in float32, Y = A^T M A in float32 -- where the plane
product  sum_c U[c] V[c]  is formed as

    f32          one float32 product per term (today's kernel: v_mfma_f32_16x16x4_f32)
    f16 x2 (3)   U = Uh + Ul, V = Vh + Vl in float16 (round to nearest), Uh Vh + Uh Vl + Ul Vh     -- VERDICT's proposal, 16/3 = 5.3x
    f16 x2 (4)   ... + Ul Vl                                                                       -- 4x
    bf16 x3 (6)  three bfloat16 pieces each (8 + 8 + 8 bits), the six products of weight >= 2^-16  -- 2.7x
    bf16 x3 (9)  all nine                                                                          -- 1.8x
    bf16 x2 (3)  two pieces, three products (16 bits of operand)                                   -- 5.3x

A product of two 16-bit-format numbers is exact in float32; the matrix core accumulates in float32: emulated as a float32 matmul of the
pieces (numpy sums in float32, in another order than the kernel: the comparison is between the variants, all summed the same way).
Per-plane power-of-two scaling of U (free: at pack time) and of V (one multiply per value) keeps float16 clear of overflow and of
its subnormal range; the script reports the error with and without it, and the largest |V| / smallest |lo| piece it met.

Bar: the error against the float64 oracle on W4_CASES-like layers must be <= today's float32 F(4x4,3x3) kernel.

    python scripts/micro/wino_splithalf.py          (a minute; result recorded in profiles/r05_tuning_notes.md section 10)
"""
import numpy as np

F32, F64 = np.float32, np.float64
BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], F64)
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], F64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], F64)


def to_bf16(x):
    """float32 -> nearest bfloat16, returned as float32"""
    u = np.asarray(x, F32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(F32)


def split(x, fmt, pieces):
    """x (float32) = sum of `pieces` numbers of the 16-bit format, each returned as float32 (remainders formed in float32: exact)."""
    out, r = [], np.asarray(x, F32).copy()
    for _ in range(pieces):
        p = r.astype(np.float16).astype(F32) if fmt == 'f16' else to_bf16(r)
        out.append(p)
        r = (r - p).astype(F32)
    return out


def plane_products(U, V, mode):
    """U [36, Cin, Cout], V [36, T, Cin] float32 -> M [36, T, Cout] float32"""
    mm = lambda a, b: np.matmul(a.astype(F32), b.astype(F32))          # float32 products and sums
    if mode == 'f32':
        return mm(V, U)
    fmt, pieces, nprod = mode
    Us, Vs = split(U, fmt, pieces), split(V, fmt, pieces)
    pairs = sorted(((i, j) for i in range(pieces) for j in range(pieces)), key=lambda ij: (ij[0] + ij[1], ij))[:nprod]
    M = np.zeros((36, V.shape[1], U.shape[2]), F32)
    for i, j in reversed(pairs):                   # small terms first
        M = (M + mm(Vs[j], Us[i])).astype(F32)
    return M


def layer(x, w, mode, scale):
    """x [H, W, Cin] (H, W multiples of 4), w [3, 3, Cin, Cout]; SAME padding.  Returns y [H, W, Cout] float32 and range statistics."""
    H, W, Cin = x.shape
    Cout = w.shape[3]
    U = np.einsum('ai,ijco,bj->abco', G, w.astype(F64), G).reshape(36, Cin, Cout)
    su = np.ones(36)
    if scale:                                      # per-plane power of two: max |U| -> [2^3, 2^4)
        su = 2.0 ** (3 - np.floor(np.log2(np.abs(U).reshape(36, -1).max(1))))
    U = (U * su[:, None, None]).astype(F32)
    xp = np.zeros((H + 2, W + 2, Cin), F32)
    xp[1:-1, 1:-1] = x
    ty, tx = H // 4, W // 4
    d = np.stack([xp[4 * i:4 * i + 6, 4 * j:4 * j + 6] for i in range(ty) for j in range(tx)])          # [T, 6, 6, Cin]
    V = np.einsum('ai,tijc->tajc', BT.astype(F32), d).astype(F32)
    V = np.einsum('tajc,bj->tabc', V, BT.astype(F32)).astype(F32)                                        # float32 transforms
    V = V.reshape(-1, 36, Cin).transpose(1, 0, 2)
    sv = np.ones(36)
    if scale:
        sv = 2.0 ** (3 - np.floor(np.log2(np.abs(V).reshape(36, -1).max(1))))
        V = (V * sv[:, None, None].astype(F32)).astype(F32)
    stats = {}
    if mode != 'f32':
        lo = split(V, mode[0], mode[1])[-1]
        nz = np.abs(lo[lo != 0])
        stats = dict(vmax=float(np.abs(V).max()), lo_min=float(nz.min()) if nz.size else 0.0,
                     lo_subnormal=float((nz < 6.1e-5).mean()) if (nz.size and mode[0] == 'f16') else 0.0)
    M = plane_products(U, V, mode)
    M = (M / (su * sv)[:, None, None].astype(F32)).astype(F32)                                          # exact (powers of two)
    M = M.reshape(6, 6, ty * tx, Cout)
    Y = np.einsum('ia,abto->ibto', AT.astype(F32), M).astype(F32)
    Y = np.einsum('ibto,jb->tijo', Y, AT.astype(F32)).astype(F32)
    y = Y.reshape(ty, tx, 4, 4, Cout).transpose(0, 2, 1, 3, 4).reshape(H, W, Cout)
    return y, stats


def direct64(x, w):
    H, W, Cin = x.shape
    xp = np.zeros((H + 2, W + 2, Cin), F64)
    xp[1:-1, 1:-1] = x
    y = np.zeros((H, W, w.shape[3]), F64)
    for i in range(3):
        for j in range(3):
            y += xp[i:i + H, j:j + W] @ w[i, j].astype(F64)
    return y


MODES = [('f32', 'f32 (today)', 1.0), (('f16', 2, 3), 'f16 x2, 3 products', 16 / 3), (('f16', 2, 4), 'f16 x2, 4 products', 4.0),
         (('bf16', 3, 6), 'bf16 x3, 6 products', 16 / 6), (('bf16', 3, 9), 'bf16 x3, 9 products', 16 / 9), (('bf16', 2, 3), 'bf16 x2, 3 products', 16 / 3)]


def main():
    cases = [(32, 32, 64, 64, 1.0), (32, 32, 256, 256, 1.0), (16, 16, 512, 128, 1.0), (32, 32, 128, 128, 30.0), (32, 32, 128, 128, 0.01)]
    print('| layer (H x W, Cin -> Cout, input scale) | ' + ' | '.join(m[1] for m in MODES) + ' |')
    print('|---|' + '---|' * len(MODES))
    worst = {m[1]: 0.0 for m in MODES}
    notes = []
    for (H, W, Cin, Cout, amp) in cases:
        rng = np.random.default_rng(H + Cin + Cout)
        x = (rng.standard_normal((H, W, Cin)) * amp).astype(F32)
        w = (rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(F32)
        ref = direct64(x, w)
        row = []
        e32 = None
        for mode, name, _ in MODES:
            cells = []
            for scale in ((False,) if mode == 'f32' else (False, True)):
                y, st = layer(x, w, mode, scale)
                e = float(np.abs(y - ref).max()) / amp
                cells.append(e)
                if mode != 'f32' and mode[0] == 'f16' and mode[2] == 3:
                    notes.append('%dx%d %d->%d x%g %s: max |V| %.3g, smallest lo piece %.2e, %.1f %% of the lo pieces subnormal' %
                                 (H, W, Cin, Cout, amp, 'scaled' if scale else 'unscaled', st['vmax'], st['lo_min'], 100 * st['lo_subnormal']))
            if mode == 'f32':
                e32 = cells[0]
                row.append('%.2e' % e32)
            else:
                row.append('%.2e / %.2e (%.1fx)' % (cells[0], cells[1], cells[1] / e32))
                worst[name] = max(worst[name], cells[1] / e32)
        print('| %d x %d, %d -> %d, x%g | ' % (H, W, Cin, Cout, amp) + ' | '.join(row) + ' |')
    print()
    print('(cells: max |y - float64| / input scale, unscaled / with per-plane power-of-two scaling, and the scaled error over today\'s)')
    for m in MODES[1:]:
        print('%-22s matrix-pipe time 1/%.1f of today: worst error %.1fx today\'s' % (m[1], m[2], worst[m[1]]))
    print()
    for n in notes:
        print(n)


if __name__ == '__main__':
    main()

"""The arithmetic of the split convolutions (csrc/conv_wsplit.hip, conv_dsplit.hip), restated in numpy.

No GPU: these tests pin the CLAIMS the kernels rely on -- the bf16 three-way split is exact, the fp16
two-way round-to-nearest split is good to 2^-22 (2^-24 typical) once the tensor is scaled by the exact power of two the
kernels derive from its tracked max |x|, the kept products are exact in fp32, and a long dot product
evaluated that way is as close to fp64 as a plain fp32 fma chain.  The GPU side of the same statements
is tests/test_split_modes_gpu.py.
"""
import numpy as np


def bf16_split3(x, rne=False):
    """h, m, l: 8 significand bits each.  Activations are split by truncation on the device (csrc: ws_split4),
    weights by round-to-nearest-even on the host (split3_host)."""
    x = np.asarray(x, np.float32)
    out, r = [], x.copy()
    for _ in range(3):
        u = r.view(np.uint32).astype(np.uint64)
        if rne:
            u = u + 0x7FFF + ((u >> 16) & 1)
        h = (u & 0xFFFF0000).astype(np.uint32).view(np.float32)
        out.append(h)
        r = (r - h).astype(np.float32)
    return out, r


def f16_split2(x):
    """h, l by round-to-nearest (csrc: ws_split4_h)."""
    x = np.asarray(x, np.float32)
    h = x.astype(np.float16)
    l = (x - h.astype(np.float32)).astype(np.float32).astype(np.float16)
    return h, l


def scale_exp(amax, margin_bits):
    """csrc ws_scale_exp (margin 2: Winograd-transformed inputs, |V| <= 2 max|x|) / ds_scale_exp (margin 1)."""
    b = np.float32(amax).view(np.uint32)
    if b == 0:
        return 0
    e = (14 - margin_bits) - (int(b >> 23) - 127)
    return max(-100, min(100, e))


def test_bf16_three_way_split_is_exact():
    rng = np.random.default_rng(0)
    # normal fp32 values whose low piece (~2^-18 |v|) is still a normal number; below that the fp32
    # subtraction v - h underflows and the split degrades gracefully (nothing in a network lives there)
    x = np.concatenate([rng.standard_normal(100000) * 10.0 ** rng.integers(-25, 30, 100000),
                        [0.0, -0.0, 1.0, -1.0, 3.0e38, 1.0e-25, np.float32(1) + np.float32(2 ** -23),
                         np.float32(2) - np.float32(2 ** -23)]]).astype(np.float32)
    for rne, bm, bl in ((False, 2.0 ** -7, 2.0 ** -14), (True, 2.0 ** -8, 2.0 ** -16)):
        (h, m, l), r = bf16_split3(x, rne)
        assert np.all(r == 0)                                # nothing left after three pieces
        assert np.all(np.abs(m) <= bm * np.abs(x)) and np.all(np.abs(l) <= bl * np.abs(x))
        assert np.array_equal((h.astype(np.float64) + m + l).astype(np.float32), x)
        for p in (h, m, l):                                  # each piece is a bf16 value
            assert np.all((p.view(np.uint32) & np.uint32(0xFFFF)) == 0)


def test_bf16_piece_products_are_exact_in_fp32_and_dropped_terms_are_small():
    rng = np.random.default_rng(1)
    a = rng.standard_normal(50000).astype(np.float32)
    b = rng.standard_normal(50000).astype(np.float32)
    (ah, am, al), _ = bf16_split3(a)             # activations: truncation
    (bh, bm, bl), _ = bf16_split3(b, rne=True)   # weights: round to nearest
    for p, q in ((ah, bh), (ah, bm), (am, bh), (am, bm), (ah, bl), (al, bh)):
        assert np.array_equal((p * q).astype(np.float32).astype(np.float64), p.astype(np.float64) * q)  # 8 x 8 bits
    kept = sum(p.astype(np.float64) * q for p, q in ((ah, bh), (ah, bm), (am, bh), (am, bm), (ah, bl), (al, bh)))
    exact = a.astype(np.float64) * b
    # dropped: am*bl + al*bm + al*bl < (2^-7 2^-16 + 2^-14 2^-8 + 2^-30) |ab| = 2^-21.4 |ab| in the worst case
    assert np.all(np.abs(exact - kept) <= 2.0 ** -21 * np.abs(exact) + 1e-300)
    assert float(np.sqrt(np.mean(((exact - kept) / np.maximum(np.abs(exact), 1e-30)) ** 2))) <= 2.0 ** -24


def test_f16_two_way_split_after_power_of_two_scaling():
    rng = np.random.default_rng(2)
    for mag in (1e-25, 1e-6, 1.0, 3e4, 1e20):
        x = (rng.standard_normal(20000) * mag).astype(np.float32)
        amax = float(np.abs(x).max())
        for margin in (1, 2):
            e = scale_exp(amax, margin)
            xs = (x * np.float32(2.0) ** e).astype(np.float32)          # exact: power of two, no overflow
            assert np.array_equal(xs.astype(np.float64), x.astype(np.float64) * 2.0 ** e)
            assert float(np.abs(xs).max()) * 2 ** (margin - 1) < 2 ** 14 < 65504
            h, l = f16_split2(xs)
            assert np.isfinite(h.astype(np.float32)).all()
            err = np.abs(xs.astype(np.float64) - h.astype(np.float64) - l.astype(np.float64))
            # <= 2^-22 relative in the worst case (value at the bottom, residual at the top of their binades;
            # 2^-24 typical) while the low piece is a normal fp16, 2^-25 absolute (scaled units) below
            assert np.all(err <= np.maximum(2.0 ** -22 * np.abs(xs), 2.0 ** -25))
            assert float(np.sqrt(np.mean((err / np.maximum(np.abs(xs), 1e-30)) ** 2))) <= 2.0 ** -23


def _chain32(a, b, kb):
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k in range(0, a.shape[1], kb):
        acc = (acc + a[:, k:k + kb].astype(np.float64) @ b[k:k + kb].astype(np.float64)).astype(np.float32)
    return acc


def test_long_dot_products_are_fp32_class_in_both_modes():
    """K = 4608 (3x3 x 512 channels): error against fp64 relative to sum |a||b|, fp32 accumulation per MFMA."""
    rng = np.random.default_rng(3)
    K, M, N = 4608, 64, 32
    a = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    b = (rng.standard_normal((K, N)) * 0.02).astype(np.float32)
    truth = a.astype(np.float64) @ b.astype(np.float64)
    s = np.abs(a.astype(np.float64)) @ np.abs(b.astype(np.float64))
    ref = float((np.abs(_chain32(a, b, 4) - truth) / s).max())          # fp32 MFMA 16x16x4 chain

    (ah, am, al), _ = bf16_split3(a)
    (bh, bm, bl), _ = bf16_split3(b, rne=True)
    acc = np.zeros((M, N), np.float32)
    for k in range(0, K, 16):
        for p, q in ((al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)):
            acc = (acc + p[:, k:k + 16].astype(np.float64) @ q[k:k + 16].astype(np.float64)).astype(np.float32)
    e_bf16 = float((np.abs(acc - truth) / s).max())

    ea, eb = scale_exp(float(np.abs(a).max()), 1), scale_exp(float(np.abs(b).max()), 1)
    ah16, al16 = f16_split2(a * np.float32(2.0) ** ea)
    bh16, bl16 = f16_split2(b * np.float32(2.0) ** eb)
    acc = np.zeros((M, N), np.float32)
    for k in range(0, K, 16):
        for p, q in ((al16, bh16), (ah16, bl16), (ah16, bh16)):
            acc = (acc + p[:, k:k + 16].astype(np.float64) @ q[k:k + 16].astype(np.float64)).astype(np.float32)
    e_f16 = float((np.abs(acc.astype(np.float64) * 2.0 ** -(ea + eb) - truth) / s).max())

    assert ref < 1e-6 and e_bf16 < 1e-6 and e_f16 < 1e-6
    assert e_bf16 <= 3 * ref and e_f16 <= 3 * ref            # same class as the plain fp32 chain


# ---------------------------------------------------------------------------------------------------
# Winograd F(4,3) of csrc/conv_w43.hip: points 0, +-a, +-b, inf with a = 5/8, b = 3/2
# ---------------------------------------------------------------------------------------------------
W4_A, W4_B = 0.625, 1.5


def w43_input_transform(d, a=W4_A, b=W4_B, dtype=np.float32):
    """d: (..., 6) -> V: (..., 6), the operation order of produce_point in conv_w43.hip."""
    f = dtype
    a2, b2 = f(a * a), f(b * b)
    a2b2, a2pb2 = f(a * a * b * b), f(a * a + b * b)
    d = [d[..., i].astype(dtype) for i in range(6)]
    e1, o1 = d[4] - b2 * d[2], f(a) * (d[3] - b2 * d[1])
    e2, o2 = d[4] - a2 * d[2], f(b) * (d[3] - a2 * d[1])
    return np.stack([(a2b2 * d[0] - a2pb2 * d[2]) + d[4], e1 + o1, e1 - o1, e2 + o2, e2 - o2,
                     (a2b2 * d[1] - a2pb2 * d[3]) + d[5]], -1).astype(dtype)


def w43_weight_transform(g, a=W4_A, b=W4_B):
    """g: (..., 3) float64 -> U: (..., 6) float64 (prepare_w43: float64 on the host, rounded once afterwards)."""
    a2, b2 = a * a, b * b
    na, nb = 2 * a2 * (a2 - b2), 2 * b2 * (b2 - a2)
    g0, g1, g2 = g[..., 0], g[..., 1], g[..., 2]
    return np.stack([g0 / (a2 * b2), (g0 + a * g1 + a2 * g2) / na, (g0 - a * g1 + a2 * g2) / na,
                     (g0 + b * g1 + b2 * g2) / nb, (g0 - b * g1 + b2 * g2) / nb, g2], -1)


def w43_output_transform(m, a=W4_A, b=W4_B, dtype=np.float32):
    f = dtype
    m = [m[..., i] for i in range(6)]
    s12, d12, s34, d34 = m[1] + m[2], m[1] - m[2], m[3] + m[4], m[3] - m[4]
    return np.stack([(m[0] + s12) + s34, f(a) * d12 + f(b) * d34, f(a * a) * s12 + f(b * b) * s34,
                     (f(a ** 3) * d12 + f(b ** 3) * d34) + m[5]], -1)


def test_winograd_f43_algebra_is_exact_in_float64():
    rng = np.random.default_rng(4)
    d = rng.standard_normal((1000, 6))
    g = rng.standard_normal((1000, 3))
    out = w43_output_transform(w43_input_transform(d, dtype=np.float64) * w43_weight_transform(g), dtype=np.float64)
    ref = np.stack([sum(d[:, i + k] * g[:, k] for k in range(3)) for i in range(4)], -1)
    np.testing.assert_allclose(out, ref, atol=1e-12)
    # every constant the kernel uses is exact in fp32
    for c in (W4_A, W4_B, W4_A ** 2, W4_B ** 2, W4_A ** 3, W4_B ** 3, W4_A ** 2 * W4_B ** 2, W4_A ** 2 + W4_B ** 2):
        assert float(np.float32(c)) == c


def _w43_conv_rows(x, w, a, b):
    """One output row of a 3x3 convolution through F(4,3) with the bf16x3 arithmetic of the kernel: x (3, W+2, Cin)
    zero padded, w (3, 3, Cin, Cout); six split products per 16-channel K-step, fp32 accumulation per MFMA."""
    W, cin, cout = x.shape[1] - 2, x.shape[2], w.shape[3]
    nq = W // 4
    acc = np.zeros((6, nq, cout), np.float32)
    for c0 in range(0, cin, 16):
        for ky in range(3):
            U = np.moveaxis(w43_weight_transform(np.moveaxis(w[ky, :, c0:c0 + 16].astype(np.float64), 0, -1), a, b), -1, 0)
            d = np.stack([x[ky, 4 * q:4 * q + 6, c0:c0 + 16] for q in range(nq)])          # (nq, 6, 16)
            V = np.moveaxis(w43_input_transform(np.moveaxis(d, 1, -1), a, b), -1, 0)          # (6, nq, 16)
            for xi in range(6):
                (ah, am, al), _ = bf16_split3(V[xi])
                (bh, bm, bl), _ = bf16_split3(U[xi].astype(np.float32), rne=True)
                for p, q in ((al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)):
                    acc[xi] = (acc[xi] + p.astype(np.float64) @ q.astype(np.float64)).astype(np.float32)
    return w43_output_transform(np.moveaxis(acc, 0, -1)).transpose(0, 2, 1).reshape(W, cout)


def test_winograd_f43_split_error_is_inside_the_conv_test_bound():
    """The numbers tests/test_conv_gpu.py holds the GPU kernel to (1e-6 max, 1.5e-7 rms of |x| conv |w|), restated
    for the kernel's arithmetic; and the reason for the point set: less round-off than the textbook 0, +-1, +-2."""
    rng = np.random.default_rng(5)
    cin, cout, W = 128, 48, 64
    x = np.maximum(rng.standard_normal((3, W + 2, cin)), 0).astype(np.float32)
    x[:, 0] = x[:, -1] = 0
    w = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    truth = sum(x[ky, kx:kx + W].astype(np.float64) @ w[ky, kx].astype(np.float64) for ky in range(3) for kx in range(3))
    bound = sum(np.abs(x[ky, kx:kx + W].astype(np.float64)) @ np.abs(w[ky, kx].astype(np.float64)) for ky in range(3) for kx in range(3))
    res = {}
    for name, (a, b) in {"kernel": (W4_A, W4_B), "textbook": (1.0, 2.0)}.items():
        r = np.abs(_w43_conv_rows(x, w, a, b) - truth) / bound
        res[name] = (float(r.max()), float(np.sqrt((r ** 2).mean())))
    assert res["kernel"][0] <= 1e-6 and res["kernel"][1] <= 1.5e-7
    assert res["kernel"][1] < res["textbook"][1]


# ---------------------------------------------------------------------------------------------------
# csrc/conv_w43h.hip: F(4,3) in fp16 arithmetic (round 4) -- per-image input scale, per-cout weight scale
# ---------------------------------------------------------------------------------------------------
W4H_TOP = 12  # conv_w43h.hip: scaled inputs lie below 2^13


def w4h_scale_exp(amax):
    """kocr_scale_exp(slot, W4H_TOP): e = 12 - exponent(max |x| of the image), clamped to [-100, 100]; 0 for an all-zero image."""
    b = np.float32(amax).view(np.uint32)
    if b == 0:
        return 0
    return max(-100, min(100, W4H_TOP - (int(b >> 23) - 127)))


def w4h_weight_exp(u_of_cout):
    """prepare_w43h: per output channel, max |U 2^wexp| over the channel's transformed weights lies in [2^14, 2^15)."""
    umax = float(np.abs(u_of_cout).max())
    if umax == 0:
        return 0
    return 15 - int(np.frexp(np.float32(umax))[1])


def _w43h_conv_rows(x, w, pieces):
    """One output row of a 3x3 convolution through F(4,3) as conv_w43h.hip evaluates it: inputs scaled by 2^e, weights by
    2^wexp[o], both split into fp16 pieces (round to nearest), products a_l b_h + a_h b_l + a_h b_h (pieces = 2) or a_h b_h
    (pieces = 1) per 16-channel K-step with fp32 accumulation, the scales undone after the output transform."""
    W, cin, cout = x.shape[1] - 2, x.shape[2], w.shape[3]
    nq = W // 4
    e = w4h_scale_exp(np.abs(x).max())
    Uall = w43_weight_transform(np.moveaxis(w.astype(np.float64), 1, -1))           # (3, cin, cout, 6)
    wexp = np.array([w4h_weight_exp(Uall[:, :, o, :]) for o in range(cout)])
    acc = np.zeros((6, nq, cout), np.float32)
    for c0 in range(0, cin, 16):
        for ky in range(3):
            U = np.moveaxis(Uall[ky, c0:c0 + 16], -1, 0)                              # (6, 16, cout)
            Us = (U.astype(np.float32) * np.float32(2.0) ** wexp).astype(np.float32)  # exact: powers of two
            d = np.stack([x[ky, 4 * q:4 * q + 6, c0:c0 + 16] for q in range(nq)]) * np.float32(2.0) ** e
            V = np.moveaxis(w43_input_transform(np.moveaxis(d, 1, -1)), -1, 0)        # (6, nq, 16); the kernel folds 2^e into
            assert float(np.abs(V).max()) < 65504                                     # the constants: same value, exactly
            for xi in range(6):
                ah, al = f16_split2(V[xi])
                bh, bl = f16_split2(Us[xi])
                prods = ((al, bh), (ah, bl), (ah, bh)) if pieces == 2 else ((ah, bh),)
                for pq, qq in prods:
                    acc[xi] = (acc[xi] + pq.astype(np.float64) @ qq.astype(np.float64)).astype(np.float32)
    out = w43_output_transform(np.moveaxis(acc, 0, -1))                               # (nq, cout, 4)
    out = out * (np.float32(2.0) ** -(e + wexp))[None, :, None]
    return out.transpose(0, 2, 1).reshape(W, cout)


def test_winograd_f43_fp16_modes_error_classes():
    """fp16x2 (KOCR_SPLIT_F16X2): inside the bound the GPU tests hold the kernels to (1e-6 max / 1.5e-7 rms of |x| conv |w|)
    and no worse than the bf16x3 F(4,3) arithmetic; one fp16 piece (KOCR_SPLIT_F16X1, the reduced-precision fast mode):
    inside ITS stated tolerance (1e-3 max / 1e-4 rms) and ~1000x above the fp32-class modes."""
    rng = np.random.default_rng(7)
    cin, cout, W = 128, 40, 64
    x = (np.maximum(rng.standard_normal((3, W + 2, cin)), 0) * 37.0).astype(np.float32)   # max |x| far from a power of two
    x[:, 0] = x[:, -1] = 0
    w = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    w[..., ::5] *= 1e-3                                                                    # channels of very different scale
    truth = sum(x[ky, kx:kx + W].astype(np.float64) @ w[ky, kx].astype(np.float64) for ky in range(3) for kx in range(3))
    bound = sum(np.abs(x[ky, kx:kx + W].astype(np.float64)) @ np.abs(w[ky, kx].astype(np.float64)) for ky in range(3) for kx in range(3))
    r_bf = np.abs(_w43_conv_rows(x, w, W4_A, W4_B) - truth) / bound
    r_h2 = np.abs(_w43h_conv_rows(x, w, 2) - truth) / bound
    r_h1 = np.abs(_w43h_conv_rows(x, w, 1) - truth) / bound
    assert r_h2.max() <= 1e-6 and np.sqrt((r_h2 ** 2).mean()) <= 1.5e-7
    assert r_h2.max() <= 1.5 * r_bf.max()
    assert 5e-6 < r_h1.max() <= 1e-3 and np.sqrt((r_h1 ** 2).mean()) <= 1e-4


def test_fp16_scales_keep_every_operand_inside_fp16():
    """The F(4,3) input transform grows a value by at most 5.28 (row sums of |B^T| with the points 0, +-5/8, +-3/2, inf), so
    inputs below 2^13 stay below 43 300 < 65 504; the per-cout weight exponent puts the channel's largest |U| in [2^14, 2^15)."""
    a, b = W4_A, W4_B
    rows = [a * a * b * b + (a * a + b * b) + 1, 1 + b * b + a * (1 + b * b), 1 + a * a + b * (1 + a * a)]
    assert max(rows) < 5.29 and max(rows) * 2.0 ** 13 < 65504
    for amax in (1e-30, 3e-7, 0.999, 1.0, 37.0, 65504.0, 3e20):
        e = w4h_scale_exp(amax)
        assert float(np.float32(amax)) * 2.0 ** e < 2.0 ** 13 and (e in (-100, 100) or float(np.float32(amax)) * 2.0 ** e >= 2.0 ** 12)
    assert w4h_scale_exp(0.0) == 0
    rng = np.random.default_rng(8)
    for scale in (1e-12, 1.0, 1e9):
        u = rng.standard_normal(200) * scale
        k = w4h_weight_exp(u)
        top = float(np.abs(u).max()) * 2.0 ** k
        assert 2.0 ** 14 <= top < 2.0 ** 15

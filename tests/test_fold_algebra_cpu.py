"""CPU suite: the two rewrites of consecutive LINEAR layers that craft.cpp applies in bf16x3 mode (DESIGN.md section 3,
"Folded linear layers"), restated in float64 with torch and held against the layer-by-layer evaluation the reference
graph prescribes (detection.py:349-353, 380-389, 106-115 -- the oracle's statement of it).  They are identities, so in
float64 both sides agree to ~1e-12; the GPU tests (tests/test_craft_gpu.py) bound what fp32 rounding order makes of them.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _rng_t(rng, *shape, scale=1.0):
    return torch.from_numpy(rng.standard_normal(shape) * scale)


def test_slice5_upconv1_chain_is_one_dilated_conv_plus_a_1x1():
    """slice5.1 (3x3, dilation 6) -> slice5.2 (1x1) -> concat with s4 -> upconv1.conv.0 (1x1) -> BN -> ReLU, with the
    composite weights and constant formed exactly as craft_load does (channel counts scaled down 16x)."""
    rng = np.random.default_rng(0)
    c4, c5, cu = 32, 64, 32  # 512, 1024, 512 in CRAFT
    s4 = _rng_t(rng, 2, c4, 13, 11)  # BN output, no ReLU: signed
    h0 = F.max_pool2d(s4, 3, 1, 1)
    w1, b1 = _rng_t(rng, c5, c4, 3, 3, scale=0.1), _rng_t(rng, c5, scale=0.1)
    w2, b2 = _rng_t(rng, c5, c5, 1, 1, scale=0.1), _rng_t(rng, c5, scale=0.1)
    wu, bu = _rng_t(rng, cu, c5 + c4, 1, 1, scale=0.1), _rng_t(rng, cu, scale=0.1)
    gamma, beta = torch.from_numpy(rng.uniform(0.5, 1.5, cu)), _rng_t(rng, cu, scale=0.1)
    mean, var = _rng_t(rng, cu, scale=0.1), torch.from_numpy(rng.uniform(0.5, 1.5, cu))

    # the reference graph, layer by layer (zero 'same' padding of the dilated conv; biases are not padded)
    s5 = F.conv2d(F.conv2d(h0, w1, b1, padding=6, dilation=6), w2, b2)
    want = F.relu(F.batch_norm(F.conv2d(torch.cat([s5, s4], 1), wu, bu), mean, var, gamma, beta, training=False, eps=1e-5))

    # craft_load: Wu_a = the s5 columns of upconv1.conv.0, composite kernel Wu_a W2 W1, constant Wu_a (W2 b1 + b2)
    wu_a, wu_b = wu[:, :c5, 0, 0], wu[:, c5:]
    p = wu_a @ w2[:, :, 0, 0]
    wc = (p @ w1.reshape(c5, -1)).reshape(cu, c4, 3, 3)
    c0 = p @ b1 + wu_a @ b2
    sc = gamma / torch.sqrt(var + 1e-5)
    pre_a, pre_b = sc, (bu + c0 - mean) * sc + beta
    # craft_run: t = composite dilated conv of h0 (no epilogue); out = relu(pre_a * (conv1x1(s4) + t) + pre_b)
    t = F.conv2d(h0, wc, None, padding=6, dilation=6)
    got = F.relu((F.conv2d(s4, wu_b) + t) * pre_a.view(1, -1, 1, 1) + pre_b.view(1, -1, 1, 1))
    assert got.dtype == torch.float64
    assert float((got - want).abs().max()) < 1e-11
    # 40 % of the products per pixel at CRAFT's sizes
    full = 9 * 512 * 1024 + 1024 * 1024 + 1536 * 512
    folded = 9 * 512 * 512 + 512 * 512
    assert abs(folded / full - 0.4) < 0.01


def test_1x1_conv_commutes_with_the_bilinear_resize():
    """conv1x1(concat(resize(y), skip)) == resize(conv1x1_y(y)) + conv1x1_skip(skip) (+ bias), for an exact 2x level and
    for a level that is not an exact half (32 -> 65), with the resize of the reference (half-pixel centres)."""
    rng = np.random.default_rng(1)
    for (hy, wy), (hs, ws) in (((12, 9), (24, 18)), ((32, 20), (65, 41))):
        cy, cs, co = 24, 40, 16
        y = F.relu(_rng_t(rng, 2, cy, hy, wy))
        skip = F.relu(_rng_t(rng, 2, cs, hs, ws))
        w, b = _rng_t(rng, co, cy + cs, 1, 1, scale=0.2), _rng_t(rng, co, scale=0.1)
        up = lambda x: F.interpolate(x, size=(hs, ws), mode="bilinear", align_corners=False)  # noqa: E731
        want = F.conv2d(torch.cat([up(y), skip], 1), w, b)
        got = up(F.conv2d(y, w[:, :cy])) + F.conv2d(skip, w[:, cy:], b)
        assert float((got - want).abs().max()) < 1e-12


def test_resize_formula_of_the_kernels_is_the_reference_resize():
    """The tap / weight formula shared by resize_bilinear_kernel and conv_ds_kernel's up-sampling epilogue
    (src = (dst + 0.5) * in/out - 0.5; lower = max(floor, 0), upper = min(ceil, in - 1); lerp = src - floor) against
    torch's align_corners=False bilinear interpolation, float64, incl. the clamped borders and a non-half ratio."""
    rng = np.random.default_rng(2)
    for (hi, wi), (ho, wo) in (((5, 7), (10, 14)), ((32, 20), (65, 41)), ((3, 3), (7, 5))):
        x = rng.standard_normal((hi, wi))

        def axis(n_in, n_out):
            src = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
            fl = np.floor(src)
            return np.maximum(fl, 0).astype(int), np.minimum(np.ceil(src), n_in - 1).astype(int), src - fl

        y0, y1, yl = axis(hi, ho)
        x0, x1, xl = axis(wi, wo)
        top = x[y0][:, x0] + (x[y0][:, x1] - x[y0][:, x0]) * xl
        bot = x[y1][:, x0] + (x[y1][:, x1] - x[y1][:, x0]) * xl
        got = top + (bot - top) * yl[:, None]
        want = F.interpolate(torch.from_numpy(x)[None, None], size=(ho, wo), mode="bilinear", align_corners=False)[0, 0].numpy()
        assert np.abs(got - want).max() < 1e-12

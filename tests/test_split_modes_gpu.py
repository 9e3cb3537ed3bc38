"""Both arithmetic modes of the wide convolutions (include/kocr.h KOCR_SPLIT_*) against fp64.

bf16x3 (default): exact 3-way bf16 split, 6 products.  f16x2: round-to-nearest 2-way fp16 split, 3
products, exact power-of-two scaling of inputs (from the tensor's tracked max |x|) and weights.  Stated
bounds, elementwise, S = |x| conv |w| in fp64:

    bf16x3:  |err| <= 1e-6 * S
    f16x2 :  |err| <= 1e-6 * S + 2^-36 * max|x| * (1 conv |w|)

The second term is the fp16 low piece going subnormal for elements more than 2^16 below the tensor's
maximum (their absolute error stays <= 2^-39 max|x|); on tensors of ordinary dynamic range it is far
below the first.  The rest of the GPU suite runs in the context's default mode; this module pins both.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

MODES = ["bf16x3", "f16x2"]


@pytest.fixture(params=MODES)
def mode_ctx(ctx, request):
    old = ctx.get_split_mode()
    ctx.set_split_mode(request.param)
    yield ctx, request.param
    ctx.set_split_mode(old)


def _ref64(x, w, dil):
    xt = torch.from_numpy(x).double().permute(0, 3, 1, 2)
    wt = torch.from_numpy(w).double().permute(3, 2, 0, 1)
    pad = dil * (w.shape[0] // 2)
    want = F.conv2d(xt, wt, None, padding=pad, dilation=dil).permute(0, 2, 3, 1).numpy()
    s = F.conv2d(xt.abs(), wt.abs(), None, padding=pad, dilation=dil).permute(0, 2, 3, 1).numpy()
    ones = F.conv2d(torch.ones_like(xt), wt.abs(), None, padding=pad, dilation=dil).permute(0, 2, 3, 1).numpy()
    return want, s, ones


def _check(got, x, w, dil, mode):
    want, s, ones = _ref64(x, w, dil)
    bound = 1e-6 * s
    if mode == "f16x2":
        bound = bound + 2.0 ** -36 * float(np.abs(x).max()) * ones
    err = np.abs(got.astype(np.float64) - want)
    worst = float((err / np.maximum(bound, 1e-300)).max())
    assert np.isfinite(got).all()
    assert worst <= 1.0, f"{mode}: max err / bound = {worst:.3f}"


CASES = [
    # N, H, W, Cin, Cout, k, dil
    (1, 48, 96, 256, 256, 3, 1),   # Winograd split kernel, 128x128 tiles
    (1, 32, 128, 64, 64, 3, 1),    # 256x64 tiles
    (1, 96, 96, 128, 192, 1, 1),   # direct split kernel, 1x1
    (1, 80, 64, 64, 128, 3, 6),    # direct split kernel, dilated
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_fp32_class_on_ordinary_data(mode_ctx, case):
    ctx, mode = mode_ctx
    n, h, w, cin, cout, k, dil = case
    rng = np.random.default_rng(hash(case) % 2 ** 32)
    x = np.maximum(rng.standard_normal((n, h, w, cin)), 0).astype(np.float32)
    wt = (rng.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / (cin * k * k))).astype(np.float32)
    _check(ctx.conv2d_nhwc(x, wt, dilation=dil), x, wt, dil, mode)


@pytest.mark.parametrize("scale_x,scale_w", [(1e-20, 1.0), (1e20, 1e-3), (3e-7, 5e4), (1.0, 1e-25)])
def test_extreme_magnitudes_are_rescaled_exactly(mode_ctx, scale_x, scale_w):
    """The power-of-two scaling must keep tiny / huge tensors inside the fp16 range without losing bits."""
    ctx, mode = mode_ctx
    rng = np.random.default_rng(11)
    x = (np.maximum(rng.standard_normal((1, 40, 64, 64)), 0) * scale_x).astype(np.float32)
    wt = (rng.standard_normal((3, 3, 64, 96)) * 0.05 * scale_w).astype(np.float32)
    _check(ctx.conv2d_nhwc(x, wt), x, wt, 1, mode)


def test_wide_dynamic_range_inside_one_tensor(mode_ctx):
    """Half of the image 1e7 times weaker than the other half: the stated f16x2 bound has the additive term."""
    ctx, mode = mode_ctx
    rng = np.random.default_rng(12)
    x = np.maximum(rng.standard_normal((1, 64, 128, 64)), 0).astype(np.float32)
    x[:, :, :64, :] *= 1e3
    x[:, :, 64:, :] *= 1e-4
    wt = (rng.standard_normal((3, 3, 64, 128)) * 0.05).astype(np.float32)
    got = ctx.conv2d_nhwc(x, wt)
    _check(got, x, wt, 1, mode)
    # and the weak half is still resolved to ~5 digits relative to ITSELF in f16x2 (7 in bf16x3)
    want, s, _ = _ref64(x, wt, 1)
    rel = np.abs(got[:, :, 70:, :] - want[:, :, 70:, :]) / s[:, :, 70:, :]
    assert float(rel.max()) <= (1e-6 if mode == "bf16x3" else 2e-4)


def test_all_zero_input(mode_ctx):
    ctx, mode = mode_ctx
    x = np.zeros((1, 32, 64, 64), np.float32)
    wt = np.ones((3, 3, 64, 64), np.float32)
    b = np.linspace(-1, 1, 64).astype(np.float32)
    got = ctx.conv2d_nhwc(x, wt, pre_b=b)
    assert np.array_equal(got, np.broadcast_to(b, got.shape))


def test_heatmaps_agree_between_modes(ctx, craft_weights):
    """Whole CRAFT forward (tracked max|x| slots through convs, pools, up-sampling, concat buffers)."""
    from oracle import craft as ocraft

    ctx.load_craft(craft_weights)
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (2, 128, 192, 3), dtype=np.uint8)
    want = ocraft.detector_predict(craft_weights, img)
    old = ctx.get_split_mode()
    try:
        out = {}
        for mode in MODES:
            ctx.set_split_mode(mode)
            out[mode] = ctx.craft_forward(img)
            assert float(np.abs(out[mode] - want).max()) <= 2e-4, mode
        assert float(np.abs(out["bf16x3"] - out["f16x2"]).max()) <= 5e-5
    finally:
        ctx.set_split_mode(old)


def test_crnn_labels_agree_between_modes(ctx, crnn_weights):
    ctx.load_crnn(crnn_weights)
    rng = np.random.default_rng(6)
    crops = rng.random((24, 31, 200), dtype=np.float32)
    old = ctx.get_split_mode()
    try:
        res = {}
        for mode in MODES:
            ctx.set_split_mode(mode)
            res[mode] = ctx.crnn_forward(crops, return_probs=True)
        la, pa = res["bf16x3"]
        lb, pb = res["f16x2"]
        # the CRNN parity bar against the oracle is 2e-4 on the softmax (tests/test_crnn_gpu.py); two
        # fp32-class evaluations of the same graph differ by round-off amplified through the two BiLSTMs
        assert float(np.abs(pa - pb).max()) <= 1e-4
        top2 = np.sort(pa, axis=-1)[..., -2:]
        safe = ((top2[..., 1] - top2[..., 0]) > 1e-3).all(axis=1)
        assert np.array_equal(la[safe], lb[safe])
    finally:
        ctx.set_split_mode(old)

"""CRAFT forward on the GPU (HIP path, through the C-ABI) vs the CPU oracle.

Tolerance (stated, fp32): max |heat_gpu - heat_oracle| <= 5e-5 — the reference's own
cross-framework bar is decimal=4, i.e. 1.5e-4 (tests/test_pytorch_keras.py:49) on
real weights; seeded weights keep activations O(1) so the same scale applies.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HEAT_TOL = 5e-5  # measured ~2e-5 (was 2e-4, 10x looser than measured)


@pytest.fixture(scope="module")
def craft_ctx(ctx, craft_weights):
    ctx.load_craft(craft_weights)
    return ctx


# (1, 64, 512): every pooled layer takes the fused conv+maxpool epilogue (W/8 = 64); the other
# shapes mix fused and unfused levels, odd sizes exercise the floor-pooling fallback
@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 96, 80), (1, 50, 70), (3, 32, 48), (1, 64, 512), (2, 32, 128)])
def test_heatmap_f32_input(craft_ctx, craft_weights, shape):
    from oracle import craft as ocraft

    n, h, w = shape
    rng = np.random.default_rng(7)
    x = rng.standard_normal((n, h, w, 3), dtype=np.float32)
    got = craft_ctx.craft_forward(x)
    want = ocraft.craft_forward(craft_weights, x)
    assert got.shape == want.shape == (n, h // 2, w // 2, 2)
    err = float(np.abs(got - want).max())
    assert err <= HEAT_TOL, f"max abs heat-map error {err}"


def test_heatmap_u8_input_fused_normalisation(craft_ctx, craft_weights):
    """uint8 input: compute_input (detection.py:34-42) fused into the first conv's loader."""
    from oracle import craft as ocraft

    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (2, 64, 96, 3), dtype=np.uint8)
    got = craft_ctx.craft_forward(img)
    want = ocraft.detector_predict(craft_weights, img)
    err = float(np.abs(got - want).max())
    assert err <= HEAT_TOL, f"max abs heat-map error {err}"


def test_micro_batching_is_invisible(craft_ctx):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (5, 32, 32, 3), dtype=np.uint8)
    a = craft_ctx.craft_forward(img, micro_batch=2)
    b = craft_ctx.craft_forward(img, micro_batch=5)
    # bit-identical in both split modes: bf16x3 has no data-dependent scale, and in fp16x2 mode the detector forwards
    # one image at a time whatever micro-batch is asked for, so the scale follows the image alone
    assert np.array_equal(a, b)
    c = craft_ctx.craft_forward(img[3:4])
    assert np.array_equal(c[0], a[3])


def test_forward_before_load_fails_loudly():
    import keras_ocr_amd

    c = keras_ocr_amd.Context(0)
    with pytest.raises(keras_ocr_amd.KocrError):
        c.craft_forward(np.zeros((1, 32, 32, 3), np.uint8))
    c.close()


def test_heatmap_at_baseline_cfg2_size(craft_ctx, craft_weights):
    """One 768x768 image (BASELINE configs[1] resolution) against the oracle."""
    from oracle import craft as ocraft
    from tests import synth

    img = synth.text_page(768, 768, 25, seed=77)[None]
    got = craft_ctx.craft_forward(img)
    want = ocraft.detector_predict(craft_weights, img)
    err = float(np.abs(got - want).max())
    assert err <= HEAT_TOL, f"max abs heat-map error {err}"


@pytest.mark.parametrize("shape", [(1, 520, 392), (2, 96, 80), (1, 1024, 1024)])
def test_folded_linear_layers_equal_the_plain_schedule(craft_ctx, craft_weights, monkeypatch, shape):
    """Two load-time / schedule-level rewrites of consecutive LINEAR layers (craft.cpp, bf16x3 mode) against the plain
    layer-by-layer schedule and against the oracle:
      KOCR_UPFOLD  -- conv1x1(concat(resize(y), skip)) as resize(conv1x1_y(y)) + conv1x1_skip(skip);
      KOCR_LINFOLD -- slice5.1 (3x3 dil 6) -> slice5.2 (1x1) -> upconv1.conv.0 (1x1 over the concat with s4), which have
                      no non-linearity between them, as ONE dilated 3x3 512 -> 512 (float64-composed weights) + a 1x1 on s4.
    520x392 has levels that are not exact halves (65 -> 32: resize ratio 0.492)."""
    from oracle import craft as ocraft
    from tests import synth

    n, h, w = shape
    if min(h, w) >= 200:
        img = np.stack([synth.text_page(h, w, 12, seed=5 + i) for i in range(n)])
    else:
        img = np.random.default_rng(5).integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    want = ocraft.detector_predict(craft_weights, img)
    got = {}
    try:
        for name, (lin, up) in (("folded", (True, True)), ("no_upfold", (True, False)), ("no_linfold", (False, True)),
                                ("plain", (False, False))):
            craft_ctx.set_schedule(fold_linear_chain=lin, fold_upsample=up)
            got[name] = craft_ctx.craft_forward(img)
    finally:
        craft_ctx.set_schedule(True, True)
    errs = {k: float(np.abs(v - want).max()) for k, v in got.items()}
    d = {k: float(np.abs(v - got["plain"]).max()) for k, v in got.items()}
    print(f"{shape}: heat-map error vs oracle {errs}; vs plain schedule {d}")
    assert max(errs.values()) <= HEAT_TOL
    assert max(d.values()) <= 5e-5


def test_heatmap_ragged_page_on_the_fp16_kernels(craft_ctx, craft_weights):
    """A page no pyramid level of which tiles (375 x 500: half of a 750 x 1000 photo at scale 1; levels 187 x 250, 93 x 125,
    46 x 62, 23 x 31 -- tools.resize_image hands the detector any int(W s) x int(H s), tools.py:387-397): since round 5 the
    vertical- / row-reuse fp16 kernels take such images through their ragged grids (masked gather, masked stores, floor
    pooling fused) instead of dropping to the F(2,3) / fp32 kernels.  Same heat-map tolerance; the profiler must show it."""
    import os

    from oracle import craft as ocraft
    from tests import synth

    img = synth.text_page(375, 500, 12, seed=91)[None]
    craft_ctx.profile_enable(True)
    craft_ctx.profile_reset()
    got = craft_ctx.craft_forward(img)
    rows = craft_ctx.profile_report()
    craft_ctx.profile_enable(False)
    want = ocraft.detector_predict(craft_weights, img)
    assert got.shape == want.shape == (1, 187, 250, 2)
    err = float(np.abs(got - want).max())
    print(f"ragged 375x500: heat-map error {err:.2e}; kernels {sorted(k for k in rows if k.startswith('conv'))}")
    assert err <= HEAT_TOL, f"max abs heat-map error {err}"
    if craft_ctx.get_split_mode() != 0 and not any(k.startswith("KOCR_") and k != "KOCR_SPLIT" for k in os.environ):
        assert any(k.startswith("conv_w4hv_256x128") and k.endswith("_rag") for k in rows), sorted(rows)
        assert any(k.startswith("conv_w4hr_256x64") and k.endswith("_rag") for k in rows), sorted(rows)
        assert any(k.startswith("conv_w4hv_256x128_pool") and k.endswith("_rag") for k in rows), sorted(rows)
        assert not any(k.startswith(("conv_ws_", "conv_wino", "conv_w4s_256x128", "conv_w4s_512x64")) and "dil" not in k for k in rows), sorted(rows)

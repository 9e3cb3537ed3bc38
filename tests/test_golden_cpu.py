"""CPU suite, part 1: the oracle (and the product's host logic) against the golden fixtures that
tests/golden/make_golden.py produced by executing the REFERENCE's own code (its PyTorch CRAFT
model, detection.compute_input and the numpy/scipy-only parts of keras_ocr/tools.py)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "reference_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_compute_input_bit_exact(gold):
    from oracle import craft

    for tag in "ab":
        x = np.stack([craft.compute_input(i) for i in gold[f"craft_{tag}_img"]])
        assert np.array_equal(x, gold[f"craft_{tag}_x"])


def test_craft_oracle_matches_reference_torch_model(gold, craft_weights):
    """The reference asserts Keras == PyTorch to 4 decimals (tests/test_pytorch_keras.py:49); the
    oracle (Keras graph restated) must meet the reference's PyTorch statement at that bar."""
    from oracle import craft

    for tag in "ab":
        got = craft.craft_forward(craft_weights, gold[f"craft_{tag}_x"])
        np.testing.assert_almost_equal(got, gold[f"craft_{tag}_heat"], decimal=4)
        assert float(np.abs(got - gold[f"craft_{tag}_heat"]).max()) < 2e-5


def test_rotated_box_ordering_and_size(gold):
    from oracle import tools

    for b, want, wh in zip(gold["rot_in"], gold["rot_out"], gold["rot_wh"]):
        got, _ = tools.get_rotated_box(b, use_min_rect=False)  # the path the golden run took
        assert np.array_equal(got, want)
        assert tools.get_rotated_width_height(got) == tuple(wh)
    # through the min-rotated-rectangle path a true rectangle comes back unchanged (float32)
    for i in (0, 1, 4):
        got, _ = tools.get_rotated_box(gold["rot_in"][i])
        np.testing.assert_allclose(got, gold["rot_out"][i], atol=1e-4)


def test_warpbox_scalar_logic(gold):
    from oracle import tools

    for b, src, dst, dsize in zip(gold["rot_in"], gold["warp_src"], gold["warp_dst"], gold["warp_dsize"]):
        box, _, _, _, ds, d = tools.warp_box_params(b, 31, 200, use_min_rect=False)
        assert np.array_equal(box, src)
        assert np.array_equal(d, dst)
        assert tuple(ds) == tuple(dsize)


def test_resize_rule_oracle_and_product(gold):
    import keras_ocr_amd
    from oracle import tools

    for h, w, ms, mx, sc, dw, dh in gold["resize_rule"]:
        shape = (int(h), int(w), 3)
        ms = int(ms) if float(ms).is_integer() else float(ms)
        for mod in (tools, keras_ocr_amd.tools):
            s = mod.resize_scale(shape, ms, int(mx))
            assert s == sc
            assert (int(shape[1] * s), int(shape[0] * s)) == (int(dw), int(dh))


def test_pipeline_plan_matches_rule(gold):
    import keras_ocr_amd

    pl = keras_ocr_amd.pipeline.Pipeline.__new__(keras_ocr_amd.pipeline.Pipeline)
    pl.scale, pl.max_size = 2, 2048
    rows = [r for r in gold["resize_rule"] if r[2] == 2 and r[3] == 2048]
    shapes = [(int(r[0]), int(r[1]), 3) for r in rows]
    scales, dhs, dws, hmax, wmax = pl._plan(shapes)
    assert scales == [r[4] for r in rows]
    assert dws == [int(r[5]) for r in rows] and dhs == [int(r[6]) for r in rows]
    assert (hmax, wmax) == (max(dhs), max(dws))


def test_fit_rule_product(gold):
    import keras_ocr_amd

    for h, w, sc, dw, dh, fh, fw in gold["fit_rule"]:
        prm = keras_ocr_amd.tools.fit_params((int(h), int(w), 3), 200, 31)
        assert (fh, fw) == (31, 200)
        if dw < 0:
            assert prm is None and sc == 1
        else:
            assert prm == (int(dw), int(dh), sc)


def test_pad_and_adjust_boxes(gold):
    import keras_ocr_amd
    from oracle import tools

    for mod in (tools, keras_ocr_amd.tools):
        assert np.array_equal(mod.pad(gold["pad_in"], width=9, height=8), gold["pad_out"])
        got = mod.adjust_boxes(gold["rot_in"], scale=1 / 2)
        assert got.dtype == gold["adjust_out"].dtype and np.array_equal(got, gold["adjust_out"])
        same = gold["rot_in"]
        assert mod.adjust_boxes(same, scale=1) is same  # identity, not a copy (tools.py:249-250)
    # the reference's asserts compare the target with itself (tools.py:371-372), so an
    # undersized target surfaces as numpy's broadcasting ValueError — mirrored as is
    with pytest.raises(ValueError):
        keras_ocr_amd.tools.pad(gold["pad_in"], width=3, height=8)

"""Perspective-warp crops on the GPU (csrc/warp.hip through kocr_warp_crops) vs the oracle
(oracle/tools.py, restating tools.py:61-117 + recognition.py:507-526).  Integer pixel work:
bit-exact (the crops are uint8 values / 255 in float32)."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _oracle_crops(images, box_groups):
    from oracle import tools as otools

    crops = []
    for im, boxes in zip(images, box_groups):
        gray = otools.rgb2gray_u8(im)
        for box in boxes:
            crops.append(otools.warp_box(gray, box, 31, 200))
    return np.array(crops, dtype="float32") / 255 if crops else np.zeros((0, 31, 200), np.float32)


def test_crops_match_oracle(ctx):
    rng = np.random.default_rng(1)
    images = rng.integers(0, 256, (2, 120, 160, 3), dtype=np.uint8)
    th = 0.4
    c, s = np.cos(th), np.sin(th)
    rot = np.array([[-40, -12], [40, -12], [40, 12], [-40, 12]], np.float32) @ np.array([[c, s], [-s, c]], np.float32)
    box_groups = [
        np.array([[[10, 20], [110, 20], [110, 50], [10, 50]],        # axis aligned
                  rot + np.float32([80, 70]),                        # rotated rectangle
                  [[-8, 5], [40, 5], [40, 30], [-8, 30]]], np.float32),  # partly outside: border 0
        np.array([[[20, 10], [36, 10], [36, 100], [20, 100]],        # tall box: scale by height
                  [[100, 60], [150, 40], [158, 60], [108, 80]]], np.float32),  # general quad
    ]
    got = ctx.warp_crops(images, box_groups)
    want = _oracle_crops(images, box_groups)
    assert got.shape == want.shape == (5, 31, 200)
    assert np.array_equal(got, want), np.abs(got - want).max()


def test_crops_from_detected_boxes(ctx):
    from oracle import postproc

    y = synth.heatmap_batch()[:1]
    boxes = postproc.get_boxes(y)
    img = synth.text_page(240, 320, 12, seed=3)[None]
    got = ctx.warp_crops(img, boxes)
    want = _oracle_crops(img, boxes)
    assert len(got) == len(boxes[0]) > 0
    assert np.array_equal(got, want)


def test_no_boxes_and_zero_size_box(ctx):
    img = np.zeros((1, 32, 32, 3), np.uint8)
    assert ctx.warp_crops(img, [np.array([])]).shape == (0, 31, 200)
    with pytest.raises(ZeroDivisionError):  # tools.py:95 divides by int(w) == 0
        ctx.warp_crops(img, [np.array([[[5, 5], [5.4, 5], [5.4, 9], [5, 9]]], np.float32)])

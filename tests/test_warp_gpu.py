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


def _oracle_warp_box(image, box, target_height=None, target_width=None, margin=0, cval=None, skip_rotate=False):
    """tools.warpBox (tools.py:61-117), full signature, from the oracle's pieces (per channel for RGB)."""
    from oracle import tools as ot

    if cval is None:
        cval = (0, 0, 0) if image.ndim == 3 else 0
    if not skip_rotate:
        box, _ = ot.get_rotated_box(box)
    w, h = ot.get_rotated_width_height(box)
    if target_width is None and target_height is None:
        target_width, target_height = w, h
    scale = min(target_width / w, target_height / h)
    dst = np.array([[margin, margin], [scale * w - margin, margin], [scale * w - margin, scale * h - margin],
                    [margin, scale * h - margin]]).astype("float32")
    M = ot.get_perspective_transform(np.asarray(box, np.float32), dst)
    dsize = (int(scale * w), int(scale * h))
    if image.ndim == 2:
        crop = ot.warp_perspective_u8(image, M, dsize)
    else:
        crop = np.stack([ot.warp_perspective_u8(image[..., c], M, dsize) for c in range(3)], -1)
    shape = (target_height, target_width, 3) if image.ndim == 3 else (target_height, target_width)
    full = (np.zeros(shape) + cval).astype("uint8")
    full[:crop.shape[0], :crop.shape[1]] = crop
    return full, M


def test_warpbox_general_signature(ctx):
    """tools.warpBox with everything the reference accepts: RGB in -> RGB out, default target size = the box's own
    size, margin, cval, skip_rotate, return_transform (tools.py:61-117)."""
    import keras_ocr_amd

    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (90, 140, 3), dtype=np.uint8)
    gray = rng.integers(0, 256, (90, 140), dtype=np.uint8)
    box = np.array([[100, 60], [30, 40], [34, 20], [104, 41]], np.float32)  # unordered start corner
    for image in (rgb, gray):
        for kw in ({}, {"target_height": 31, "target_width": 200}, {"margin": 3, "target_height": 40, "target_width": 120},
                   {"cval": (7, 8, 9) if image.ndim == 3 else 5, "target_height": 64, "target_width": 64},
                   {"skip_rotate": True, "target_height": 31, "target_width": 200}):
            b = box if not kw.get("skip_rotate") else np.array([[30, 40], [104, 41], [100, 60], [34, 20]], np.float32)
            got, M = keras_ocr_amd.tools.warpBox(image, b, return_transform=True, ctx=ctx, **kw)
            want, Mw = _oracle_warp_box(image, b, **kw)
            assert got.shape == want.shape and got.dtype == np.uint8
            assert np.array_equal(got, want), (kw, np.abs(got.astype(int) - want).max())
            assert np.array_equal(M, Mw)  # same float64 operation order on the device as in the oracle
    assert keras_ocr_amd.tools.warpBox(rgb, box, ctx=ctx).shape[2] == 3
    with pytest.raises(AssertionError):
        keras_ocr_amd.tools.warpBox(rgb, box, target_height=31, ctx=ctx)
    with pytest.raises(ZeroDivisionError):
        keras_ocr_amd.tools.warpBox(gray, np.array([[5, 5], [5.4, 5], [5.4, 9], [5, 9]], np.float32), ctx=ctx)


def test_rotated_box_helper_equals_oracle():
    import keras_ocr_amd
    from oracle import tools as ot

    rng = np.random.default_rng(4)
    for _ in range(50):
        c = rng.uniform(20, 80, 2)
        a = rng.uniform(0, np.pi)
        w, h = rng.uniform(5, 40, 2)
        r = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        q = (np.array([[-w, -h], [w, -h], [w, h], [-w, h]]) @ r.T + c + rng.normal(0, 0.3, (4, 2))).astype(np.float32)
        q = np.roll(q, int(rng.integers(0, 4)), 0)
        got, rot = keras_ocr_amd.tools.get_rotated_box(q)
        want, wrot = ot.get_rotated_box(q)
        assert np.allclose(got, want, atol=1e-4)
        assert keras_ocr_amd.tools.get_rotated_width_height(got) == ot.get_rotated_width_height(want)
        assert np.isclose(rot, wrot, atol=1e-6) or (np.isnan(rot) and np.isnan(wrot))


def test_fit_letterbox_and_crop_vs_oracle(ctx):
    """tools.fit (tools.py:402-452) in both modes against cv2.resize's restatement + the reference's paste / window."""
    import keras_ocr_amd
    from oracle import tools as ot

    rng = np.random.default_rng(5)
    for shape in ((40, 180, 3), (100, 100, 3), (20, 400, 3), (62, 400, 3), (31, 200, 3)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        for mode in ("letterbox", "crop"):
            got, scale = keras_ocr_amd.tools.fit(img, width=200, height=31, cval=0, mode=mode, return_scale=True, ctx=ctx)
            prm = keras_ocr_amd.tools.fit_params(shape, 200, 31, mode)
            if prm is None:
                assert got is img and scale == 1
                continue
            rw, rh, sc = prm
            resized = ot.cv_resize_linear_u8(img, (rw, rh))
            if mode == "letterbox":
                want = np.zeros((31, 200, 3), np.uint8)
                want[:resized.shape[0], :resized.shape[1]] = resized[:31, :200]
            else:
                want = resized[:31, :200]
            assert scale == sc and got.shape == want.shape
            assert np.array_equal(got, want), (shape, mode)
    with pytest.raises(NotImplementedError):
        keras_ocr_amd.tools.fit(img, width=10, height=10, mode="stretch", ctx=ctx)

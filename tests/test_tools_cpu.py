"""Host-side ``keras_ocr_amd.tools`` pieces that need no GPU: ``tools.read`` against what the reference's two cv2 calls
do (tools.py:19-38: a PATH goes through cv2.imread -- 8-bit, 3 channels, EXIF orientation applied --, a BUFFER through
cv2.imdecode(IMREAD_UNCHANGED) -- stored depth / channels, no orientation), the float restatement of the image ops the
reference applies to non-uint8 images (cv2.resize / cvtColor / warpPerspective in float), and the visual helpers.
OpenCV itself is absent from this image: the float ops are checked against torch's bilinear interpolation and against
the oracle's fixed-point uint8 warp (oracle/tools.py), not against cv2."""
import io
import os

import numpy as np
import pytest

from keras_ocr_amd import tools


def _exif_jpeg(path_or_buf, arr, orientation):
    from PIL import Image

    im = Image.fromarray(arr)
    exif = Image.Exif()
    exif[0x0112] = orientation
    im.save(path_or_buf, format="JPEG", quality=95, exif=exif.tobytes())


def test_read_applies_exif_orientation_for_paths_only(tmp_path):
    """cv2.imread honours the EXIF orientation (tools.py:36); imdecode(IMREAD_UNCHANGED) does not (tools.py:30-31)."""
    arr = np.zeros((40, 80, 3), np.uint8)
    arr[:, :40] = (250, 30, 30)    # left half red, right half blue: orientation 6 = "rotate 90 degrees clockwise to display"
    arr[:, 40:] = (30, 30, 250)
    path = str(tmp_path / "rot.jpg")
    _exif_jpeg(path, arr, 6)
    from_path = tools.read(path)
    assert from_path.shape == (80, 40, 3) and from_path.dtype == np.uint8          # rotated upright
    assert from_path[10, 20, 0] > 200 and from_path[70, 20, 2] > 200                # red on top, blue below
    with open(path, "rb") as f:
        raw = f.read()
    from_buffer = tools.read(io.BytesIO(raw))
    assert from_buffer.shape == (40, 80, 3)                                          # as stored
    assert from_buffer[20, 10, 0] > 200 and from_buffer[20, 70, 2] > 200
    plain = str(tmp_path / "plain.jpg")
    _exif_jpeg(plain, arr, 1)
    assert tools.read(plain).shape == (40, 80, 3)
    arr2 = np.arange(12, dtype=np.uint8).reshape(2, 2, 3)
    assert tools.read(arr2) is arr2                                                  # ndarray passthrough (tools.py:26-27)
    with pytest.raises(AssertionError, match="Could not find image at path"):
        tools.read(str(tmp_path / "missing.png"))


def test_read_depth_and_alpha_follow_the_two_cv2_calls(tmp_path):
    from PIL import Image

    rgba = np.zeros((6, 7, 4), np.uint8)
    rgba[..., 0], rgba[..., 1], rgba[..., 2], rgba[..., 3] = 10, 20, 30, 128
    p = str(tmp_path / "a.png")
    Image.fromarray(rgba).save(p)
    assert np.array_equal(tools.read(p), rgba[..., :3])                              # imread drops alpha (no compositing)
    with open(p, "rb") as f:
        assert np.array_equal(tools.read(io.BytesIO(f.read())), rgba[..., :3])       # BGR2RGB of 4 channels drops it too
    g16 = (np.arange(42, dtype=np.uint16).reshape(6, 7) * 1500)
    p16 = str(tmp_path / "g16.png")
    Image.fromarray(g16).save(p16)
    got = tools.read(p16)                                                            # imread: 8 bit (high byte), 3 channels
    assert got.dtype == np.uint8 and got.shape == (6, 7, 3) and np.array_equal(got[..., 0], (g16 >> 8).astype(np.uint8))
    with open(p16, "rb") as f, pytest.raises(ValueError, match="gray image read from a buffer"):
        tools.read(io.BytesIO(f.read()))                                             # cvtColor(BGR2RGB) of a 2-D array fails
    pg = str(tmp_path / "g.png")
    Image.fromarray(np.full((5, 5), 77, np.uint8)).save(pg)
    assert np.array_equal(tools.read(pg), np.full((5, 5, 3), 77, np.uint8))          # imread replicates gray


def test_float_resize_is_half_pixel_bilinear():
    """the ORACLE's statement of cv2.resize on a float image (the product runs it on the GPU: tests/test_float_gpu.py)"""
    import torch
    from oracle import tools as otools

    rng = np.random.default_rng(0)
    im = (rng.random((37, 53, 3)) * 255).astype(np.float32)
    for dw, dh in ((106, 74), (80, 55), (53, 37), (71, 60)):
        got = otools.resize_linear_float(im, (dw, dh))
        want = torch.nn.functional.interpolate(torch.from_numpy(im).permute(2, 0, 1)[None], size=(dh, dw), mode="bilinear",
                                               align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
        assert got.shape == (dh, dw, 3) and got.dtype == np.float32
        assert float(np.abs(got - want).max()) <= 2e-3


def test_float_warp_agrees_with_the_fixed_point_warp_to_one_level():
    from oracle import tools as otools

    rng = np.random.default_rng(1)
    gray = rng.integers(0, 256, (90, 140), dtype=np.uint8)
    for box in (np.array([[20, 15], [110, 15], [110, 40], [20, 40]], np.float32),
                np.array([[30, 20], [100, 38], [94, 62], [24, 44]], np.float32)):
        want = otools.warp_box(gray, box, 31, 200).astype(np.float32)
        got = otools.warp_box_float(gray.astype(np.float32), box, 31, 200)
        assert got.shape == (31, 200) and got.dtype == np.float32
        d = np.abs(got - want)
        assert float(d.max()) <= 1.0 + 1e-3 and float((d > 0.51).mean()) <= 0.02     # rounding of the uint8 result only
    assert np.array_equal(otools.rgb2gray_float(np.full((2, 2, 3), 100, np.float32)), np.full((2, 2), 100, np.float32))


def test_draw_helpers_smoke():
    import matplotlib

    matplotlib.use("Agg")
    image = np.full((120, 200, 3), 255, np.uint8)
    preds = [("left", np.array([[10, 10], [60, 10], [60, 30], [10, 30]], np.float32)),
             ("right", np.array([[130, 70], [190, 70], [190, 95], [130, 95]], np.float32))]
    boxed = tools.drawBoxes(image, preds, boxes_format="predictions", thickness=2)
    assert boxed.shape == image.shape and (boxed != image).any() and (image == 255).all()   # input not modified
    ax = tools.drawAnnotations(image, preds)
    texts = sorted(t.get_text() for t in ax.texts)
    assert texts == ["left", "right"]                                                # one margin label per word ...
    sides = {t.get_text(): t.get_position()[0] for t in ax.texts}
    assert sides["left"] < 0 < 1 < sides["right"]                                    # ... on the side its box starts on
    assert len(ax.images) == 1
    import matplotlib.pyplot as plt

    plt.close("all")

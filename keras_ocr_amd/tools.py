"""Host-side mirror of the ``keras_ocr.tools`` functions on the inference path
(reference ``keras_ocr/tools.py``); pixel work runs in libkocr on the GPU."""
import hashlib
import io
import os
import typing
import urllib.parse
import urllib.request

import numpy as np

from . import _lib


def read(filepath_or_buffer: typing.Union[str, io.BytesIO, np.ndarray]):
    """tools.read (tools.py:19-38): ndarray passthrough; files / buffers / URLs are decoded to
    RGB with PIL (OpenCV is not a dependency of this package)."""
    if isinstance(filepath_or_buffer, np.ndarray):
        return filepath_or_buffer
    from PIL import Image  # local import: only needed for file inputs

    if hasattr(filepath_or_buffer, "read"):
        return np.array(Image.open(io.BytesIO(filepath_or_buffer.read())).convert("RGB"))
    if isinstance(filepath_or_buffer, str):
        if urllib.parse.urlparse(filepath_or_buffer).scheme in ("http", "https"):
            return read(urllib.request.urlopen(filepath_or_buffer))  # pylint: disable=consider-using-with
        assert os.path.isfile(filepath_or_buffer), "Could not find image at path: " + filepath_or_buffer
        return np.array(Image.open(filepath_or_buffer).convert("RGB"))
    raise TypeError(f"Unsupported image source: {type(filepath_or_buffer)}")


def resize_scale(shape, max_scale, max_size):
    """The scale rule of tools.resize_image (tools.py:387-392); max() includes the channel dim."""
    if max(shape) * max_scale > max_size:
        return max_size / max(shape)
    return max_scale


def resize_image(image, max_scale, max_size, ctx=None):
    """tools.resize_image (tools.py:378-398) -> (image, scale); cv2.resize runs on the GPU."""
    scale = resize_scale(image.shape, max_scale, max_size)
    ctx = ctx or _lib.default_context()
    out = ctx.resize_pad(image[np.newaxis], (int(image.shape[1] * scale), int(image.shape[0] * scale)))[0]
    return out, scale


def pad(image, width: int, height: int, cval: int = 255):
    """tools.pad (tools.py:356-375)."""
    if len(image.shape) == 3:
        output_shape = (height, width, image.shape[-1])
    else:
        output_shape = (height, width)
    assert height >= output_shape[0], "Input height must be less than output height."
    assert width >= output_shape[1], "Input width must be less than output width."
    padded = np.zeros(output_shape, dtype=image.dtype) + cval
    padded[: image.shape[0], : image.shape[1]] = image
    return padded


def adjust_boxes(boxes, scale=1, boxes_format="boxes"):
    """tools.adjust_boxes (tools.py:232-260)."""
    if scale == 1:
        return boxes
    if boxes_format == "boxes":
        return np.array(boxes) * scale
    if boxes_format == "lines":
        return [[(np.array(box) * scale, character) for box, character in line] for line in boxes]
    if boxes_format == "predictions":
        return [(word, np.array(box) * scale) for word, box in boxes]
    raise NotImplementedError(f"Unsupported boxes format: {boxes_format}")


def warpBox(image, box, target_height=None, target_width=None, ctx=None):  # pylint: disable=invalid-name
    """tools.warpBox (tools.py:61-117) for the recogniser's use: RGB image -> gray crop, uint8."""
    assert target_width is not None and target_height is not None, \
        "keras-ocr_amd warps to a fixed target size (the recogniser's input)."
    ctx = ctx or _lib.default_context()
    if image.ndim == 2:
        image = np.repeat(image[..., None], 3, -1)
    crops = ctx.warp_crops(image[np.newaxis], [np.asarray(box, np.float32)[np.newaxis]], target_height, target_width)
    return np.rint(crops[0] * 255).astype(np.uint8)


def fit_params(shape, width, height, mode="letterbox"):
    """The size rule of tools.fit (tools.py:425-441) -> (resize_width, resize_height, scale) or None
    when the image already has the requested size."""
    x_scale = width / shape[1]
    y_scale = height / shape[0]
    if x_scale == 1 and y_scale == 1:
        return None
    if (x_scale <= y_scale and mode == "letterbox") or (x_scale >= y_scale and mode == "crop"):
        scale = width / shape[1]
        resize_width = width
        resize_height = (width / shape[1]) * shape[0]
    else:
        scale = height / shape[0]
        resize_height = height
        resize_width = scale * shape[1]
    return int(resize_width), int(resize_height), scale


def fit(image, width: int, height: int, cval: int = 255, mode="letterbox", return_scale=False, ctx=None):
    """tools.fit (tools.py:402-452), letterbox mode: cv2.resize + paste on a cval canvas, on the GPU."""
    prm = fit_params(image.shape, width, height, mode)
    if prm is None:
        fitted, scale = image, 1
    else:
        resize_width, resize_height, scale = prm
        if mode != "letterbox":
            raise NotImplementedError(f"Unsupported mode: {mode}")
        ctx = ctx or _lib.default_context()
        # letterbox: one side equals the target, the other is not larger
        fitted = ctx.resize_pad(image[np.newaxis], (resize_width, resize_height), out_hw=(height, width), cval=cval)[0]
    if not return_scale:
        return fitted
    return fitted, scale


def read_and_fit(filepath_or_array, width: int, height: int, cval: int = 255, mode="letterbox"):
    """tools.read_and_fit (tools.py:455-481)."""
    image = read(filepath_or_array) if isinstance(filepath_or_array, str) else filepath_or_array
    return fit(image=image, width=width, height=height, cval=cval, mode=mode)


def sha256sum(filename):
    """tools.sha256sum (tools.py:484-492): hex digest of a file, read in blocks."""
    digest = hashlib.sha256()
    with open(filename, "rb") as f:
        while True:
            block = f.read(1 << 20)
            if not block:
                return digest.hexdigest()
            digest.update(block)


def get_default_cache_dir():
    """tools.get_default_cache_dir (tools.py:495-498): $KERAS_OCR_CACHE_DIR or ~/.keras-ocr."""
    return os.environ.get("KERAS_OCR_CACHE_DIR") or os.path.join(os.path.expanduser("~"), ".keras-ocr")


def download_and_verify(url, sha256=None, cache_dir=None, verbose=True, filename=None):
    """tools.download_and_verify (tools.py:501-530).  Contract: the file lives at ``cache_dir/filename`` (default
    name = last URL path component); it is fetched when missing OR when a hash is given and the cached copy does not
    match; after that a given hash must match (AssertionError otherwise); the path is returned."""
    target = os.path.join(cache_dir or get_default_cache_dir(),
                          filename or os.path.basename(urllib.parse.urlparse(url).path))
    os.makedirs(os.path.dirname(target), exist_ok=True)
    say = print if verbose else (lambda *_: None)
    say("Looking for " + target)
    cached_ok = os.path.isfile(target) and (not sha256 or sha256sum(target) == sha256)
    if not cached_ok:
        say("Downloading " + target)
        urllib.request.urlretrieve(url, target)
    assert sha256 is None or sha256sum(target) == sha256, "Error occurred verifying sha256."
    return target


def drawBoxes(image, boxes, color=(255, 0, 0), thickness=5, boxes_format="boxes"):  # pylint: disable=invalid-name
    """tools.drawBoxes (tools.py:189-229) on PIL instead of cv2.polylines (visual helper)."""
    from PIL import Image, ImageDraw  # pylint: disable=import-outside-toplevel

    if len(boxes) == 0:
        return image
    if boxes_format == "lines":
        boxes = [box for line in boxes for box, _ in line]
    if boxes_format == "predictions":
        boxes = [box for _, box in boxes]
    canvas = Image.fromarray(np.ascontiguousarray(image))
    draw = ImageDraw.Draw(canvas)
    for box in boxes:
        pts = [tuple(int(v) for v in p) for p in np.asarray(box)]
        draw.line(pts + [pts[0]], fill=tuple(color), width=int(thickness), joint="curve")
    return np.asarray(canvas)


def drawAnnotations(image, predictions, ax=None):  # pylint: disable=invalid-name
    """tools.drawAnnotations (tools.py:150-186): boxes + arrowed text labels on a matplotlib axis."""
    import matplotlib.pyplot as plt  # pylint: disable=import-outside-toplevel

    if ax is None:
        _, ax = plt.subplots()
    ax.imshow(drawBoxes(image=image, boxes=predictions, boxes_format="predictions"))
    predictions = sorted(predictions, key=lambda p: p[1][:, 1].min())
    left, right = [], []
    for word, box in predictions:
        (left if box[:, 0].min() < image.shape[1] / 2 else right).append((word, box))
    ax.set_yticks([])
    ax.set_xticks([])
    for side, group in zip(["left", "right"], [left, right]):
        for index, (text, box) in enumerate(group):
            y = 1 - (index / len(group))
            xy = box[0] / np.array([image.shape[1], image.shape[0]])
            xy[1] = 1 - xy[1]
            ax.annotate(text=text, xy=xy, xytext=(-0.05 if side == "left" else 1.05, y), xycoords="axes fraction",
                        arrowprops={"arrowstyle": "->", "color": "r"}, color="r", fontsize=14,
                        horizontalalignment="right" if side == "left" else "left")
    return ax

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


def _decoded_to_array(img, as_imread):
    """PIL image -> what cv2 hands back for it (before the BGR->RGB swap of tools.py:38).  ``as_imread``: cv2.imread's
    default IMREAD_COLOR (always 8-bit, 3 channels: 16-bit samples keep their high byte, alpha is dropped, gray is
    replicated); otherwise cv2.imdecode(IMREAD_UNCHANGED) (depth and channel count as stored)."""
    mode = img.mode
    if mode in ("I;16", "I;16B", "I;16L", "I"):  # 16-bit gray (PIL reads 16-bit PNG / TIFF gray as one of these)
        arr = np.asarray(img).astype(np.uint16)
        if as_imread:
            g = (arr >> 8).astype(np.uint8)
            return np.stack([g, g, g], -1)
        return arr  # 2-D uint16: the reference's cvtColor(BGR2RGB) rejects it, see read()
    if mode in ("1", "L"):
        g = np.asarray(img.convert("L"))
        return np.stack([g, g, g], -1) if as_imread else g
    if mode == "LA":
        if as_imread:
            g = np.asarray(img.convert("L"))
            return np.stack([g, g, g], -1)
        return np.asarray(img)  # 2 channels: rejected like the reference
    if mode == "P":
        img = img.convert("RGBA" if "transparency" in img.info else "RGB")
        mode = img.mode
    if mode in ("RGBA", "RGBa", "RGBX"):
        return np.asarray(img.convert("RGBA"))[..., :3]  # imread drops alpha; BGR2RGB of 4 channels drops it as well
    return np.asarray(img.convert("RGB"))


def read(filepath_or_buffer: typing.Union[str, io.BytesIO, np.ndarray]):
    """tools.read (tools.py:19-38): ndarray passthrough; files / buffers / URLs are decoded to RGB with PIL (OpenCV is
    not a dependency of this package), following what the reference's two cv2 calls do:

    * a PATH goes through ``cv2.imread`` (tools.py:36): 8-bit, 3 channels, and the EXIF orientation of the file is
      APPLIED (imread's default) -- a JPEG a phone stored rotated comes out upright;
    * a BUFFER / file object / URL goes through ``cv2.imdecode(IMREAD_UNCHANGED)`` (tools.py:30-31): stored depth and
      channels, EXIF orientation NOT applied; an alpha channel is dropped by the BGR->RGB conversion that follows, a
      gray or gray+alpha image makes that conversion fail (cv2.error in the reference, ValueError here)."""
    if isinstance(filepath_or_buffer, np.ndarray):
        return filepath_or_buffer
    from PIL import Image, ImageOps  # local import: only needed for file inputs

    if hasattr(filepath_or_buffer, "read"):
        image = _decoded_to_array(Image.open(io.BytesIO(filepath_or_buffer.read())), as_imread=False)
        if image.ndim != 3 or image.shape[2] not in (3, 4):
            raise ValueError("tools.read: a gray image read from a buffer cannot be converted BGR->RGB "
                             "(the reference's cv2.cvtColor raises here, tools.py:38); pass a path or an array")
        return image[..., :3]
    if isinstance(filepath_or_buffer, str):
        if urllib.parse.urlparse(filepath_or_buffer).scheme in ("http", "https"):
            return read(urllib.request.urlopen(filepath_or_buffer))  # pylint: disable=consider-using-with
        assert os.path.isfile(filepath_or_buffer), "Could not find image at path: " + filepath_or_buffer
        return _decoded_to_array(ImageOps.exif_transpose(Image.open(filepath_or_buffer)), as_imread=True)
    raise TypeError(f"Unsupported image source: {type(filepath_or_buffer)}")


# ---------------------------------------------------------------------------------------------------------------------
# float images (any dtype but uint8): the reference's cv2 calls work in the image's own type -- cv2.resize interpolates in
# float (tools.py:394), cvtColor / warpPerspective likewise (recognition.py:510, tools.py:107).  Since round 5 both run on the
# GPU (kocr_resize_pad_f32, kocr_warp_crops_f32: csrc/imgproc.hip, warp.hip); their numpy statement, against which the
# kernels are tested bit for bit, lives in oracle/tools.py (resize_linear_float, rgb2gray_float, warp_box_float).  OpenCV is
# absent from this image: the float path is NOT checked against cv2 itself.
# ---------------------------------------------------------------------------------------------------------------------
def resize_scale(shape, max_scale, max_size):
    """The scale rule of tools.resize_image (tools.py:387-392); max() includes the channel dim."""
    if max(shape) * max_scale > max_size:
        return max_size / max(shape)
    return max_scale


def resize_image(image, max_scale, max_size, ctx=None):
    """tools.resize_image (tools.py:378-398) -> (image, scale); cv2.resize runs on the GPU."""
    scale = resize_scale(image.shape, max_scale, max_size)
    ctx = ctx or _lib.default_context()
    if np.asarray(image).dtype != np.uint8:  # cv2.resize interpolates a float image in float
        im = np.asarray(image, dtype=np.float32)
        dsize = (int(im.shape[1] * scale), int(im.shape[0] * scale))
        if dsize == (im.shape[1], im.shape[0]):
            return im.copy(), scale
        out = ctx.resize_pad_f32(im[np.newaxis] if im.ndim == 3 else im[np.newaxis, ..., np.newaxis], dsize)[0]
        return (out if im.ndim == 3 else out[..., 0]), scale
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


def get_rotated_width_height(box):
    """tools.get_rotated_width_height (tools.py:41-57): integer mean of opposite side lengths of [tl, tr, br, bl]."""
    p = np.asarray(box, dtype=np.float64)
    side = lambda i, j: float(np.sqrt(((p[i] - p[j]) ** 2).sum()))  # noqa: E731
    return int((side(0, 1) + side(2, 3)) / 2), int((side(0, 3) + side(1, 2)) / 2)


def _min_area_rectangle(points):
    """What shapely's ``MultiPoint(points).minimum_rotated_rectangle`` returns for the 4 box points (tools.py:543-547):
    the smallest rectangle having a side on the convex hull.  Degenerate input -> None (the reference's
    AttributeError fallback to the raw points, :548-550)."""
    pts = sorted({(float(x), float(y)) for x, y in np.asarray(points, dtype=np.float64)})
    if len(pts) < 3:
        return None
    turn = lambda o, a, b: (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])  # noqa: E731
    hull = []
    for seq in (pts, pts[::-1]):
        chain = []
        for q in seq:
            while len(chain) >= 2 and turn(chain[-2], chain[-1], q) <= 0:
                chain.pop()
            chain.append(q)
        hull += chain[:-1]
    if len(hull) < 3:
        return None
    h = np.array(hull)
    best = None
    for i in range(len(h)):
        edge = h[(i + 1) % len(h)] - h[i]
        ux, uy = edge / np.sqrt((edge ** 2).sum())
        u, v = h[:, 0] * ux + h[:, 1] * uy, -h[:, 0] * uy + h[:, 1] * ux
        area = (u.max() - u.min()) * (v.max() - v.min())
        if best is None or area < best[0]:
            best = (area, ux, uy, u.min(), u.max(), v.min(), v.max())
    _, ux, uy, u0, u1, v0, v1 = best
    return np.array([[u * ux - v * uy, u * uy + v * ux] for u, v in ((u0, v0), (u1, v0), (u1, v1), (u0, v1))])


def get_rotated_box(points):
    """tools.get_rotated_box (tools.py:533-581) -> (box float32 [tl, tr, br, bl], rotation in radians)."""
    rect = _min_area_rectangle(points)
    pts = np.asarray(points) if rect is None else rect
    by_x = pts[np.argsort(pts[:, 0], kind="stable")]
    left = by_x[:2][np.argsort(by_x[:2, 1], kind="stable")]
    tl, bl = left
    right = by_x[2:]
    far_first = right[np.argsort(np.sqrt(((right.astype(np.float64) - tl.astype(np.float64)) ** 2).sum(1)), kind="stable")[::-1]]
    br, tr = far_first
    with np.errstate(divide="ignore", invalid="ignore"):
        rotation = np.arctan((tl[0] - bl[0]) / (tl[1] - bl[1]))
    return np.array([tl, tr, br, bl], dtype="float32"), rotation


def warpBox(image, box, target_height=None, target_width=None, margin=0, cval=None, return_transform=False,  # pylint: disable=invalid-name
            skip_rotate=False, ctx=None):
    """tools.warpBox (tools.py:61-117), full signature: warp the quadrilateral ``box`` of ``image`` (HxW or HxWx3
    uint8) into a ``target_height x target_width`` rectangle (default: the box's own integer size), ``margin`` pixels
    inside it, the rest filled with ``cval``.  The homography and the warp run on the GPU (kocr_warp_quads); an RGB
    image is warped channel by channel with the same fixed-point arithmetic cv2.warpPerspective applies per channel."""
    if cval is None:
        cval = (0, 0, 0) if len(image.shape) == 3 else 0
    if not skip_rotate:
        box, _ = get_rotated_box(box)
    box = np.asarray(box, dtype=np.float32)
    w, h = get_rotated_width_height(box)
    assert (target_width is None and target_height is None) or (target_width is not None and target_height is not None), \
        "Either both or neither of target width and height must be provided."
    if target_width is None and target_height is None:
        target_width, target_height = w, h
    scale = min(target_width / w, target_height / h)  # ZeroDivisionError for an empty box, as in the reference
    dst = np.array([[margin, margin], [scale * w - margin, margin], [scale * w - margin, scale * h - margin],
                    [margin, scale * h - margin]]).astype("float32")
    dsize = (int(scale * w), int(scale * h))
    ctx = ctx or _lib.default_context()
    image = np.asarray(image)
    if image.dtype != np.uint8:
        # any other dtype (tools.py:61-117 hands the image to cv2.warpPerspective as it is): the float kernel
        # (kocr_warp_crops_f32: float32 bilinear interpolation, constant-0 border), channel by channel; it derives the rotated
        # box itself, so it serves the default geometry (margin 0, skip_rotate False) -- the one recognize_from_boxes uses
        if margin != 0 or skip_rotate or return_transform:
            raise NotImplementedError("warpBox on a non-uint8 image: margin != 0, skip_rotate and return_transform are "
                                      "implemented for uint8 images only")
        img = image.astype(np.float32)
        planes = [img[..., None]] if img.ndim == 2 else [img[..., c:c + 1] for c in range(img.shape[2])]
        crops = np.concatenate([ctx.warp_crops_f32(p[np.newaxis], [box[np.newaxis]], int(target_height), int(target_width))
                                for p in planes])
        target_shape = (target_height, target_width, 3) if len(image.shape) == 3 else (target_height, target_width)
        full = (np.zeros(target_shape) + cval).astype("uint8")  # tools.py:109-113: a uint8 canvas whatever the image's dtype
        ch, cw = min(dsize[1], target_height), min(dsize[0], target_width)
        with np.errstate(invalid="ignore"):
            if image.ndim == 2:
                full[:ch, :cw] = crops[0, :ch, :cw]
            else:
                full[:ch, :cw] = np.moveaxis(crops[:, :ch, :cw], 0, -1)
        return full
    # channel c as a gray RGB image: OpenCV's RGB->gray of (v, v, v) is v, so the crop is channel c's warp, bit for bit
    planes = [image] if image.ndim == 2 else [image[..., c] for c in range(image.shape[2])]
    stack = np.stack([np.repeat(p[..., None], 3, -1) for p in planes])
    n = len(planes)
    crops, tf = ctx.warp_quads(stack, [box] * n, [dst] * n, np.arange(n), [dsize] * n, int(target_height),
                               int(target_width), return_transforms=True)
    crops = np.rint(crops * 255).astype(np.uint8)
    target_shape = (target_height, target_width, 3) if len(image.shape) == 3 else (target_height, target_width)
    full = (np.zeros(target_shape) + cval).astype("uint8")
    ch, cw = min(dsize[1], target_height), min(dsize[0], target_width)
    if image.ndim == 2:
        full[:ch, :cw] = crops[0, :ch, :cw]
    else:
        full[:ch, :cw] = np.moveaxis(crops[:, :ch, :cw], 0, -1)
    if return_transform:
        return full, tf[0]
    return full


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
    """tools.fit (tools.py:402-452), modes "letterbox" and "crop": cv2.resize on the GPU + paste / window."""
    prm = fit_params(image.shape, width, height, mode)
    if prm is None:
        fitted, scale = image, 1
    else:
        resize_width, resize_height, scale = prm
        ctx = ctx or _lib.default_context()
        if mode == "letterbox":
            # one side equals the target, the other is not larger: resized image pasted top-left on a cval canvas
            fitted = ctx.resize_pad(image[np.newaxis], (resize_width, resize_height), out_hw=(height, width), cval=cval)[0]
        elif mode == "crop":
            # one side equals the target, the other is not smaller: the top-left height x width window of the resize
            fitted = ctx.resize_pad(image[np.newaxis], (resize_width, resize_height))[0][:height, :width]
        else:
            raise NotImplementedError(f"Unsupported mode: {mode}")
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
    """tools.drawAnnotations (tools.py:150-186): the image with its boxes, every word written in the left or right
    margin (whichever side its box starts on), top to bottom, with an arrow to the box's first corner."""
    import matplotlib.pyplot as plt  # pylint: disable=import-outside-toplevel

    if ax is None:
        _, ax = plt.subplots()
    ax.imshow(drawBoxes(image=image, boxes=predictions, boxes_format="predictions"))
    ax.set_xticks([])
    ax.set_yticks([])
    height, width = image.shape[:2]
    columns = {"left": [], "right": []}
    for word, box in sorted(predictions, key=lambda wb: wb[1][:, 1].min()):
        columns["left" if box[:, 0].min() < width / 2 else "right"].append((word, box))
    label_x = {"left": -0.05, "right": 1.05}
    align = {"left": "right", "right": "left"}
    for side, entries in columns.items():
        for rank, (word, box) in enumerate(entries):
            anchor = (box[0, 0] / width, 1 - box[0, 1] / height)  # axes fraction, y upwards
            ax.annotate(text=word, xy=anchor, xytext=(label_x[side], 1 - rank / len(entries)), xycoords="axes fraction",
                        arrowprops={"arrowstyle": "->", "color": "r"}, color="r", fontsize=14,
                        horizontalalignment=align[side])
    return ax

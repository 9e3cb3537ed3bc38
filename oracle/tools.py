"""Oracle: image / geometry helpers of the hot path (numpy, CPU).  TEST INFRASTRUCTURE ONLY.

Restates the ``keras_ocr.tools`` functions ``Pipeline.recognize`` uses (reference
``keras_ocr/tools.py``): resize_image :378-398, pad :356-375, warpBox :61-117,
get_rotated_box :533-581, get_rotated_width_height :41-57, adjust_boxes :232-260, plus the
``cv2.cvtColor(RGB2GRAY)`` call of recognize_from_boxes (``recognition.py:507-510``).

In-repo logic (scale rule, padding, point ordering, width/height, scale, paste) follows the
reference line by line and is pinned by ``tests/golden`` fixtures generated from the
reference's own functions.  The OpenCV/shapely calls underneath are [3P]; the
libraries are installed in neither interpreter of the image, so each restatement is **cross-checked by an independent
implementation** instead (skimage ProjectiveTransform for the homography, numpy.linalg.inv + scipy map_coordinates on
1/32-px coordinates for the warp -- bit for bit --, skimage float bilinear resize within 1 LSB, Pillow's ITU-R 601
gray within 1 LSB, a brute-force float64 rectangle search for shapely: tests/test_thirdparty_crosscheck_cpu.py).
They are restated from their published algorithms:

  cv2.resize(INTER_LINEAR, u8)   half-pixel mapping, 11-bit fixed-point coefficients,
                                 horizontal pass in int32, vertical pass
                                 ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2
  cv2.cvtColor(RGB2GRAY, u8)     (R*9798 + G*19235 + B*3735 + 2^14) >> 15
  cv2.getPerspectiveTransform    8x8 linear system in float64, LU with partial pivoting
  cv2.warpPerspective(u8)        M^-1 (3x3 adjugate, float64), per-pixel float64 map,
                                 coordinates rounded (half-to-even) to 1/32 px, bilinear with
                                 15-bit weights, BORDER_CONSTANT 0
  shapely minimum_rotated_rectangle   min-area enclosing rectangle over hull edges (float64)
"""
import math

import numpy as np


# ---------------------------------------------------------------------------------------
# tools.resize_image / pad / adjust_boxes
# ---------------------------------------------------------------------------------------
def resize_scale(shape, max_scale, max_size):
    """tools.py:387-392 — note max() runs over the full shape, channel dim included."""
    if max(shape) * max_scale > max_size:
        return max_size / max(shape)
    return max_scale


def _resize_axis_tables(src, dst, horizontal):
    """OpenCV resize INTER_LINEAR coefficient tables for one axis (u8 fixed-point path).

    Horizontal: source index clamped with the fraction forced to 0 (xofs/ialpha set-up);
    vertical: fraction kept, the two source rows are clipped individually."""
    scale = src / dst  # double, 1/inv_scale
    i0 = np.zeros(dst, dtype=np.int64)
    i1 = np.zeros(dst, dtype=np.int64)
    c0 = np.zeros(dst, dtype=np.int64)
    c1 = np.zeros(dst, dtype=np.int64)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)  # (float)((dx+0.5)*scale_x - 0.5)
        s = int(math.floor(f))
        f = np.float32(f - np.float32(s))
        if horizontal:
            if s < 0:
                f, s = np.float32(0), 0
            if s >= src - 1:
                f, s = np.float32(0), src - 1
        # saturate_cast<short>(c * INTER_RESIZE_COEF_SCALE): round half to even
        c1[d] = int(np.rint(np.float32(f) * np.float32(2048)))
        c0[d] = int(np.rint((np.float32(1) - np.float32(f)) * np.float32(2048)))
        i0[d] = min(max(s, 0), src - 1)
        i1[d] = min(max(s + 1, 0), src - 1)
    return i0, i1, c0, c1


def cv_resize_linear_u8(image, dsize):
    """cv2.resize(image, dsize=(W', H')) for uint8 HxWxC, INTER_LINEAR (generic C path)."""
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = image.shape[:2]
    img = image.astype(np.int64)
    if img.ndim == 2:
        img = img[..., None]
    xi0, xi1, xa0, xa1 = _resize_axis_tables(sw, dw, True)
    yi0, yi1, yb0, yb1 = _resize_axis_tables(sh, dh, False)
    # horizontal pass: int rows scaled by 2^11
    rows = img[:, xi0, :] * xa0[None, :, None] + img[:, xi1, :] * xa1[None, :, None]
    s0 = rows[yi0]
    s1 = rows[yi1]
    out = (((yb0[:, None, None] * (s0 >> 4)) >> 16) + ((yb1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    if image.ndim == 2:
        out = out[..., 0]
    return out


def resize_image(image, max_scale, max_size):
    """tools.py:378-398."""
    scale = resize_scale(image.shape, max_scale, max_size)
    return cv_resize_linear_u8(image, (int(image.shape[1] * scale), int(image.shape[0] * scale))), scale


def pad(image, width, height, cval=255):
    """tools.py:356-375."""
    if len(image.shape) == 3:
        output_shape = (height, width, image.shape[-1])
    else:
        output_shape = (height, width)
    assert height >= output_shape[0], "Input height must be less than output height."
    assert width >= output_shape[1], "Input width must be less than output width."
    padded = np.zeros(output_shape, dtype=image.dtype) + cval
    padded[: image.shape[0], : image.shape[1]] = image
    return padded


def adjust_boxes(boxes, scale=1):
    """tools.py:249-252 (boxes_format='boxes')."""
    if scale == 1:
        return boxes
    return np.array(boxes) * scale


def rgb2gray_u8(image):
    """cv2.cvtColor(image, COLOR_RGB2GRAY) for uint8 (15-bit coefficients)."""
    im = image.astype(np.int64)
    g = (im[..., 0] * 9798 + im[..., 1] * 19235 + im[..., 2] * 3735 + (1 << 14)) >> 15
    return g.astype(np.uint8)


# ---------------------------------------------------------------------------------------
# tools.get_rotated_box / get_rotated_width_height
# ---------------------------------------------------------------------------------------
def min_rotated_rect_f64(points):
    """shapely MultiPoint(points).minimum_rotated_rectangle exterior (4 corners, float64).

    Min-area rectangle over the convex hull's edges; ties -> first edge.  For the rectangles
    getBoxes emits this is the input itself up to float64 round-off."""
    pts = np.asarray(points, dtype=np.float64)
    uniq = sorted(set(map(tuple, pts.tolist())))
    if len(uniq) < 3:
        raise AttributeError("degenerate")

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower = []
    for p in uniq:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    upper = []
    for p in reversed(uniq):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    hull = lower[:-1] + upper[:-1]
    if len(hull) < 3:
        raise AttributeError("degenerate")
    best = None
    for i in range(len(hull)):
        x0, y0 = hull[i]
        x1, y1 = hull[(i + 1) % len(hull)]
        dx, dy = x1 - x0, y1 - y0
        ln = math.sqrt(dx * dx + dy * dy)
        ux, uy = dx / ln, dy / ln
        us = [px * ux + py * uy for px, py in hull]
        vs = [-px * uy + py * ux for px, py in hull]
        umin, umax, vmin, vmax = min(us), max(us), min(vs), max(vs)
        area = (umax - umin) * (vmax - vmin)
        if best is None or area < best[0]:
            best = (area, ux, uy, umin, umax, vmin, vmax)
    _, ux, uy, umin, umax, vmin, vmax = best
    return np.array([[u * ux - v * uy, u * uy + v * ux]
                     for u, v in ((umin, vmin), (umax, vmin), (umax, vmax), (umin, vmax))], dtype=np.float64)


def get_rotated_box(points, use_min_rect=True):
    """tools.py:533-581 — returns (pts float32 [tl,tr,br,bl], rotation).  ``use_min_rect=False``
    takes the reference's AttributeError fallback (tools.py:548-550): the raw points."""
    points = np.asarray(points)
    try:
        if not use_min_rect:
            raise AttributeError("fallback requested")
        pts = min_rotated_rect_f64(points)
    except AttributeError:
        pts = points
    xSorted = pts[np.argsort(pts[:, 0], kind="stable"), :]
    leftMost = xSorted[:2, :]
    rightMost = xSorted[2:, :]
    leftMost = leftMost[np.argsort(leftMost[:, 1], kind="stable"), :]
    (tl, bl) = leftMost
    D = np.sqrt(((tl[np.newaxis].astype(np.float64) - rightMost.astype(np.float64)) ** 2).sum(1))
    (br, tr) = rightMost[np.argsort(D, kind="stable")[::-1], :]
    pts = np.array([tl, tr, br, bl], dtype="float32")
    with np.errstate(divide="ignore", invalid="ignore"):
        rotation = np.arctan((tl[0] - bl[0]) / (tl[1] - bl[1]))
    return pts, rotation


def _dist(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return math.sqrt(float(((a - b) ** 2).sum()))


def get_rotated_width_height(box):
    """tools.py:41-57 (scipy cdist == float64 euclidean)."""
    w = (_dist(box[0], box[1]) + _dist(box[2], box[3])) / 2
    h = (_dist(box[0], box[3]) + _dist(box[1], box[2])) / 2
    return int(w), int(h)


# ---------------------------------------------------------------------------------------
# cv2.getPerspectiveTransform / warpPerspective
# ---------------------------------------------------------------------------------------
def solve8_lu(A, b):
    """Gaussian elimination with partial pivoting in float64, fixed operation order (the
    host code of libkocr performs exactly the same sequence)."""
    n = 8
    A = [[float(A[i][j]) for j in range(n)] for i in range(n)]
    b = [float(v) for v in b]
    for col in range(n):
        piv = col
        for r in range(col + 1, n):
            if abs(A[r][col]) > abs(A[piv][col]):
                piv = r
        if A[piv][col] == 0.0:
            raise ZeroDivisionError("singular perspective system")
        if piv != col:
            A[piv], A[col] = A[col], A[piv]
            b[piv], b[col] = b[col], b[piv]
        for r in range(col + 1, n):
            f = A[r][col] / A[col][col]
            if f != 0.0:
                for c in range(col, n):
                    A[r][c] = A[r][c] - f * A[col][c]
                b[r] = b[r] - f * b[col]
    x = [0.0] * n
    for r in range(n - 1, -1, -1):
        s = b[r]
        for c in range(r + 1, n):
            s = s - A[r][c] * x[c]
        x[r] = s / A[r][r]
    return x


def get_perspective_transform(src, dst):
    """cv2.getPerspectiveTransform(src, dst): src, dst float32 (4,2) -> 3x3 float64."""
    src = np.asarray(src, dtype=np.float32)
    dst = np.asarray(dst, dtype=np.float32)
    A = [[0.0] * 8 for _ in range(8)]
    b = [0.0] * 8
    for i in range(4):
        sx, sy = float(src[i, 0]), float(src[i, 1])
        dx, dy = float(dst[i, 0]), float(dst[i, 1])
        A[i][0] = A[i + 4][3] = sx
        A[i][1] = A[i + 4][4] = sy
        A[i][2] = A[i + 4][5] = 1.0
        A[i][6] = -sx * dx
        A[i][7] = -sy * dx
        A[i + 4][6] = -sx * dy
        A[i + 4][7] = -sy * dy
        b[i] = dx
        b[i + 4] = dy
    x = solve8_lu(A, b)
    return np.array(x + [1.0], dtype=np.float64).reshape(3, 3)


def invert3(M):
    """cv::invert for 3x3 float64: adjugate / determinant, fixed operation order."""
    m = [[float(M[i][j]) for j in range(3)] for i in range(3)]
    d = (m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1])
         - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0])
         + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]))
    if d == 0.0:
        return [[0.0] * 3 for _ in range(3)]
    d = 1.0 / d
    t = [[0.0] * 3 for _ in range(3)]
    t[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) * d
    t[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * d
    t[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * d
    t[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) * d
    t[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * d
    t[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * d
    t[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) * d
    t[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * d
    t[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * d
    return t


def warp_perspective_u8(image, M, dsize, assoc="pixel"):
    """cv2.warpPerspective(image, M, dsize) — 2-D uint8, INTER_LINEAR, BORDER_CONSTANT 0.

    ``assoc``: floating-point association of the coordinate numerators.  "pixel" (the oracle's and the device code's
    order) evaluates ``(Mi0 x + Mi1 y) + Mi2`` per pixel.  "blockwise" is the order OpenCV's WarpPerspectiveInvoker is
    believed to use (imgwarp.cpp; not executable here, no cv2): per block of 64 columns starting at bx,
    ``X0 = (Mi0 bx + Mi1 y) + Mi2`` and per pixel ``X0 + Mi0 x1`` with x = bx + x1 (likewise Y and W).  The two differ
    by one float64 rounding; that can move a 1/32-pixel coordinate across a rounding tie and with it a crop pixel by
    one grey level.  tests/test_oracle_cpu.py::test_warp_association_orders_differ_in_few_pixels counts how often."""
    dw, dh = int(dsize[0]), int(dsize[1])
    H, W = image.shape[:2]
    Mi = np.array(invert3(M), dtype=np.float64)
    out = np.zeros((dh, dw), dtype=np.uint8)
    if dw <= 0 or dh <= 0:
        return out
    xs = np.arange(dw, dtype=np.float64)[None, :]
    ys = np.arange(dh, dtype=np.float64)[:, None]
    if assoc == "pixel":
        X0 = (Mi[0, 0] * xs + Mi[0, 1] * ys) + Mi[0, 2]
        Y0 = (Mi[1, 0] * xs + Mi[1, 1] * ys) + Mi[1, 2]
        W0 = (Mi[2, 0] * xs + Mi[2, 1] * ys) + Mi[2, 2]
    elif assoc == "blockwise":
        bw = min(64, dw)                      # BLOCK_SZ^2 / min(BLOCK_SZ / 2, height) columns per block
        bx = (xs // bw) * bw
        x1 = xs - bx
        X0 = ((Mi[0, 0] * bx + Mi[0, 1] * ys) + Mi[0, 2]) + Mi[0, 0] * x1
        Y0 = ((Mi[1, 0] * bx + Mi[1, 1] * ys) + Mi[1, 2]) + Mi[1, 0] * x1
        W0 = ((Mi[2, 0] * bx + Mi[2, 1] * ys) + Mi[2, 2]) + Mi[2, 0] * x1
    else:
        raise ValueError(assoc)
    with np.errstate(divide="ignore", invalid="ignore"):
        Wi = np.where(W0 != 0, 32.0 / W0, 0.0)
    fX = np.clip(X0 * Wi, -2147483648.0, 2147483647.0)
    fY = np.clip(Y0 * Wi, -2147483648.0, 2147483647.0)
    X = np.rint(fX).astype(np.int64)  # saturate_cast<int>: round half to even
    Y = np.rint(fY).astype(np.int64)
    sx, sy = X >> 5, Y >> 5
    ax, ay = X & 31, Y & 31
    img = image.astype(np.int64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        return np.where(ok, img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0)

    w00 = (32 - ax) * (32 - ay) * 32
    w01 = ax * (32 - ay) * 32
    w10 = (32 - ax) * ay * 32
    w11 = ax * ay * 32
    acc = w00 * tap(sy, sx) + w01 * tap(sy, sx + 1) + w10 * tap(sy + 1, sx) + w11 * tap(sy + 1, sx + 1)
    out[:] = ((acc + (1 << 14)) >> 15).astype(np.uint8)
    return out


def warp_box_params(box, target_height, target_width, use_min_rect=True):
    """The scalar part of tools.warpBox (:86-106): ordered box, (w,h), scale, M, crop dsize."""
    box, _ = get_rotated_box(box, use_min_rect)
    w, h = get_rotated_width_height(box)
    scale = min(target_width / w, target_height / h)  # ZeroDivisionError like the reference
    dst = np.array([[0, 0], [scale * w, 0], [scale * w, scale * h], [0, scale * h]]).astype("float32")
    M = get_perspective_transform(box, dst)
    return box, (w, h), scale, M, (int(scale * w), int(scale * h)), dst


def warp_box(image, box, target_height, target_width):
    """tools.warpBox (:61-117) for a 2-D (gray) image, margin=0, cval=0."""
    _, _, _, M, dsize, _ = warp_box_params(box, target_height, target_width)
    crop = warp_perspective_u8(image, M, dsize)
    full = np.zeros((target_height, target_width), dtype=np.uint8)
    full[: crop.shape[0], : crop.shape[1]] = crop
    return full


# ---------------------------------------------------------------------------------------------------------------------
# Float images (any dtype but uint8): the reference's cv2 calls work in the image's own type -- cv2.resize interpolates in
# float (tools.py:394), cvtColor / warpPerspective likewise (recognition.py:510, tools.py:107).  This is the numpy
# restatement of those three operations (round 4's host path of the product; since round 5 the product runs them on the GPU
# and these are the ORACLE its kernels are compared with, tests/test_float_gpu.py).  NOT pinned to OpenCV, which is absent
# from this image: checked against torch's half-pixel bilinear interpolation and against the fixed-point uint8 warp only
# (tests/test_oracle_cpu.py) -- "parity unpinned" for the float path.
# ---------------------------------------------------------------------------------------------------------------------
def resize_linear_float(image, dsize):
    """cv2.resize(image, dsize=(width, height)) for a float image, INTER_LINEAR: source coordinate
    (d + 0.5) * (src / dst) - 0.5, taps clamped to the image (replicated border), horizontal pass then vertical
    pass, float32 coefficients."""
    src = np.asarray(image, dtype=np.float32)
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = src.shape[:2]
    if (dw, dh) == (sw, sh):
        return src.copy()

    def taps(dst_n, src_n):
        f = (np.arange(dst_n, dtype=np.float64) + 0.5) * (src_n / dst_n) - 0.5
        i0 = np.floor(f).astype(np.int64)
        a = (f - i0).astype(np.float32)
        a[i0 < 0] = 0
        i0 = np.clip(i0, 0, src_n - 1)
        i1 = np.clip(i0 + 1, 0, src_n - 1)
        return i0, i1, a

    x0, x1, ax = taps(dw, sw)
    y0, y1, ay = taps(dh, sh)
    ax = ax.reshape((1, dw) + (1,) * (src.ndim - 2))
    ay = ay.reshape((dh, 1) + (1,) * (src.ndim - 2))
    rows = src[:, x0] * (np.float32(1) - ax) + src[:, x1] * ax
    return (rows[y0] * (np.float32(1) - ay) + rows[y1] * ay).astype(np.float32)


def rgb2gray_float(image):
    """cv2.cvtColor(float image, COLOR_RGB2GRAY): 0.299 R + 0.587 G + 0.114 B in float32."""
    im = np.asarray(image, dtype=np.float32)
    return im[..., 0] * np.float32(0.299) + im[..., 1] * np.float32(0.587) + im[..., 2] * np.float32(0.114)


def warp_box_float(gray, box, target_height=31, target_width=200):
    """tools.warpBox (tools.py:61-117, margin 0, cval 0) of a 2-D float image: get_rotated_box, integer width / height,
    homography box -> [[0,0],[s w,0],[s w,s h],[0,s h]], cv2.warpPerspective with INTER_LINEAR (source coordinates rounded
    to 1/32 pixel as OpenCV's remap does, float weights, constant-0 border), pasted top-left into target_height x
    target_width zeros."""
    box, _ = get_rotated_box(box)
    w, h = get_rotated_width_height(box)
    scale = min(target_width / w, target_height / h)  # ZeroDivisionError for an empty box, as in the reference
    dst = np.array([[0, 0], [scale * w, 0], [scale * w, scale * h], [0, scale * h]], np.float32).astype(np.float64)
    srcq = np.asarray(box, np.float32).astype(np.float64)
    a, b = [], []
    for (x, y), (u, v) in zip(srcq, dst):  # getPerspectiveTransform: 8 x 8 system for M (src -> dst)
        a.append([x, y, 1, 0, 0, 0, -x * u, -y * u])
        a.append([0, 0, 0, x, y, 1, -x * v, -y * v])
        b += [u, v]
    m = np.append(np.linalg.solve(np.array(a), np.array(b)), 1.0).reshape(3, 3)
    mi = np.linalg.inv(m)
    cw, ch = int(scale * w), int(scale * h)
    out = np.zeros((target_height, target_width), np.float32)
    cw, ch = min(cw, target_width), min(ch, target_height)
    if cw <= 0 or ch <= 0:
        return out
    xs, ys = np.meshgrid(np.arange(cw, dtype=np.float64), np.arange(ch, dtype=np.float64))
    den = mi[2, 0] * xs + mi[2, 1] * ys + mi[2, 2]
    den = np.where(den != 0, 1.0 / den, 0.0)
    fx = np.rint((mi[0, 0] * xs + mi[0, 1] * ys + mi[0, 2]) * den * 32).astype(np.int64)
    fy = np.rint((mi[1, 0] * xs + mi[1, 1] * ys + mi[1, 2]) * den * 32).astype(np.int64)
    x0, y0 = fx >> 5, fy >> 5
    ax, ay = ((fx & 31) / 32.0).astype(np.float32), ((fy & 31) / 32.0).astype(np.float32)
    g = np.asarray(gray, np.float32)
    hh, ww = g.shape

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < hh) & (xx >= 0) & (xx < ww)
        return np.where(ok, g[np.clip(yy, 0, hh - 1), np.clip(xx, 0, ww - 1)], np.float32(0))

    one = np.float32(1)
    out[:ch, :cw] = (tap(y0, x0) * ((one - ax) * (one - ay)) + tap(y0, x0 + 1) * (ax * (one - ay))
                     + tap(y0 + 1, x0) * ((one - ax) * ay) + tap(y0 + 1, x0 + 1) * (ax * ay))
    return out

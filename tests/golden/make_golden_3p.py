"""Independent cross-check fixtures for the [3P] halves of the oracle (OpenCV / shapely semantics).

Run with the image's SECOND interpreter, which ships scikit-image 0.18.3, scipy 1.7.1, Pillow 8.4 and
h5py 3.3.0 (none of them importable from the main interpreter):

    /opt/conda/bin/python3.9 tests/golden/make_golden_3p.py

OpenCV, shapely and TensorFlow are installed in NEITHER interpreter, so the reference's third-party
calls cannot be executed here.  What CAN be done is to state each of those operations a second time with
libraries that had no part in writing ``oracle/`` or the HIP kernels, run both on the same seeded inputs,
and commit the outputs of the independent statement as fixtures (``tests/golden/thirdparty_golden.npz``);
``tests/test_thirdparty_crosscheck_cpu.py`` then holds ``oracle/`` to them.  Nothing in this file imports
``oracle`` or ``keras_ocr_amd``.

  operation in the reference                    independent statement used here
  --------------------------------------------  -------------------------------------------------------------
  cv2.connectedComponentsWithStats(conn=4)      skimage.measure.label(connectivity=1) + regionprops
    (detection.py:227-236)                        (label order = raster order of the first pixel, area, bbox)
  cv2.dilate(RECT k x k, anchor k//2)           scipy.ndimage.maximum_filter(size=k, origin 0, zero border)
    (detection.py:259-264)                        = max over src(p + j - k//2), j in [0,k): OpenCV's formula;
                                                  odd k also against skimage.morphology.binary_dilation
  findContours(...)[0] fragment + hull          skimage.measure.label(connectivity=2) fragments,
    (detection.py:267-273)                        scipy.spatial.ConvexHull (Qhull) vertices
  cv2.minAreaRect / boxPoints                   brute-force rotation search in float64 over Qhull's edges
    (detection.py:273)                            (area + corner set)
  shapely minimum_rotated_rectangle             the same search on 4-point inputs (tools.py:543-547)
  cv2.getPerspectiveTransform                   skimage.transform.ProjectiveTransform.estimate (SVD based)
    (tools.py:96-106)
  cv2.warpPerspective INTER_LINEAR u8           numpy.linalg.inv + per-pixel float64 map, coordinates rounded to
    (tools.py:107)                                1/32 px (half to even), scipy.ndimage.map_coordinates(order=1,
                                                  mode='grid-constant'), round half up  -- with 1/32-px coordinates
                                                  the four weights are multiples of 2^-10, so OpenCV's 15-bit
                                                  fixed-point blend is exact and must agree bit for bit
  cv2.resize INTER_LINEAR u8                    skimage.transform.resize(order=1, mode='edge', no anti-aliasing)
    (tools.py:394)                                in float64 (OpenCV's 11-bit coefficients: within 1 LSB)
  cv2.cvtColor(RGB2GRAY) u8                     PIL.Image.convert('L') (ITU-R 601, 16-bit coefficients: within 1 LSB)
    (recognition.py:510)
"""
import os

import numpy as np
import scipy.ndimage as ndi
from scipy.spatial import ConvexHull
from skimage import measure, morphology, transform
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
out = {}
rng = np.random.default_rng(31337)


def smooth_field(h, w, sigma, seed):
    r = np.random.default_rng(seed)
    f = ndi.gaussian_filter(r.standard_normal((h, w)), sigma)
    return (f / np.abs(f).max()).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# 1. connected components, 4-connectivity: label order, area, bbox
# ------------------------------------------------------------------------------------------------
masks = []
tables = []
for i, (h, w, sigma, thr) in enumerate([(96, 128, 2.5, 0.25), (80, 100, 1.5, 0.3), (64, 64, 0.8, 0.4), (120, 90, 4.0, 0.1)]):
    m = smooth_field(h, w, sigma, 100 + i) > thr
    lab = measure.label(m, connectivity=1)
    rows = []
    for p in measure.regionprops(lab):
        ys, xs = np.nonzero(lab == p.label)
        first = int(ys[0]) * w + int(xs[0])  # np.nonzero is in raster order
        y0, x0, y1, x1 = p.bbox
        rows.append([first, int(p.area), x0, y0, x1 - x0, y1 - y0])
    rows.sort()
    pad = np.zeros((128, 128), bool)
    pad[:h, :w] = m
    masks.append(pad)
    t = np.full((400, 6), -1, np.int64)
    t[:len(rows)] = rows
    tables.append(t)
out["ccl_masks"] = np.array(masks)
out["ccl_shapes"] = np.array([(96, 128), (80, 100), (64, 64), (120, 90)])
out["ccl_tables"] = np.array(tables)  # per component, sorted by first pixel: first, area, left, top, width, height

# ------------------------------------------------------------------------------------------------
# 2. dilation with a k x k rectangle, anchor k//2, zero outside the ROI
# ------------------------------------------------------------------------------------------------
rois, dil = [], []
for k in range(1, 10):
    roi = rng.random((23, 31)) < 0.03
    roi[0, 0] = roi[-1, -1] = True  # corners: border handling
    d = ndi.maximum_filter(roi.astype(np.uint8), size=(k, k), mode="constant", cval=0, origin=0).astype(bool)
    if k % 2 == 1:
        assert np.array_equal(d, morphology.binary_dilation(roi, np.ones((k, k), bool))), k
    rois.append(roi)
    dil.append(d)
out["dil_rois"] = np.array(rois)
out["dil_out"] = np.array(dil)  # index k-1

# ------------------------------------------------------------------------------------------------
# 3. fragments (8-connectivity), hull and min-area rectangle of the chosen fragment
# ------------------------------------------------------------------------------------------------
def min_area_rect_bruteforce(pts):
    """float64 rotation search over the hull's edge directions -> (area, 4 corners)."""
    hull = ConvexHull(pts)
    hv = pts[hull.vertices].astype(np.float64)
    best = None
    for i in range(len(hv)):
        e = hv[(i + 1) % len(hv)] - hv[i]
        ang = np.arctan2(e[1], e[0])
        c, s = np.cos(-ang), np.sin(-ang)
        R = np.array([[c, -s], [s, c]])
        q = hv @ R.T
        lo, hi = q.min(0), q.max(0)
        area = (hi[0] - lo[0]) * (hi[1] - lo[1])
        if best is None or area < best[0] - 1e-9:
            corners = np.array([[lo[0], lo[1]], [hi[0], lo[1]], [hi[0], hi[1]], [lo[0], hi[1]]]) @ R
            best = (area, corners)
    return best[0], best[1], hv


frag_masks, frag_pick, hull_v, rect_area, rect_corners = [], [], [], [], []
for i in range(10):
    h, w = 40, 56
    m = smooth_field(h, w, 2.0 + 0.3 * i, 500 + i) > 0.35
    if i % 2 == 0:  # rotated bars: non-trivial rectangles
        yy, xx = np.mgrid[:h, :w]
        a = np.deg2rad(17 * i + 8)
        u = (xx - w / 2) * np.cos(a) + (yy - h / 2) * np.sin(a)
        v = -(xx - w / 2) * np.sin(a) + (yy - h / 2) * np.cos(a)
        m = (np.abs(u) < 17) & (np.abs(v) < 4)
        if i % 4 == 0:
            m |= smooth_field(h, w, 1.5, 900 + i) > 0.55  # extra fragments
    lab = measure.label(m, connectivity=2)
    n = lab.max()
    assert n >= 1
    firsts = []
    for l in range(1, n + 1):
        ys, xs = np.nonzero(lab == l)
        firsts.append((int(ys[0]) * w + int(xs[0]), l))
    pick = max(firsts)[1]  # the fragment whose raster-first pixel comes LAST (findContours lists it first)
    fm = lab == pick
    ys, xs = np.nonzero(fm)
    pts = np.stack([xs, ys], 1).astype(np.float64)
    if len(np.unique(xs)) < 2 or len(np.unique(ys)) < 2 or len(pts) < 3:
        continue
    area, corners, hv = min_area_rect_bruteforce(pts)
    frag_masks.append(m)
    frag_pick.append(fm)
    hv_pad = np.full((64, 2), -1.0)
    hv_pad[:len(hv)] = hv
    hull_v.append(hv_pad)
    rect_area.append(area)
    rect_corners.append(corners)
out["frag_masks"] = np.array(frag_masks)
out["frag_pick"] = np.array(frag_pick)
out["hull_vertices"] = np.array(hull_v)
out["rect_area"] = np.array(rect_area)
out["rect_corners"] = np.array(rect_corners)

# shapely.minimum_rotated_rectangle of 4 points (the boxes getBoxes emits are rectangles already; general
# quadrilaterals exercise the search)
quads = np.array([
    [[10, 20], [110, 20], [110, 50], [10, 50]],
    [[50.5, 10.25], [120.75, 40.5], [108.25, 69.5], [38.0, 39.25]],
    [[30, 90], [34, 10], [60, 12], [56, 92]],
    [[0, 0], [40, 5], [43, 30], [-2, 22]],
    [[5, 5], [60, 8], [70, 40], [2, 30]],
], dtype=np.float64)
qa, qc = [], []
for q in quads:
    a, c, _ = min_area_rect_bruteforce(q)
    qa.append(a)
    qc.append(c)
out["quad_in"] = quads
out["quad_rect_area"] = np.array(qa)
out["quad_rect_corners"] = np.array(qc)

# ------------------------------------------------------------------------------------------------
# 4. perspective transform + warp
# ------------------------------------------------------------------------------------------------
srcs, dsts, Ms = [], [], []
for i in range(6):
    src = np.array([[12, 7], [150, 20], [140, 60], [5, 40]], np.float64) + rng.uniform(-4, 4, (4, 2))
    if i == 0:
        src = np.array([[20, 30], [120, 30], [120, 61], [20, 61]], np.float64)  # axis aligned
    sw = rng.uniform(60, 200)
    sh = rng.uniform(12, 31)
    dst = np.array([[0, 0], [sw, 0], [sw, sh], [0, sh]], np.float64)
    src = src.astype(np.float32).astype(np.float64)
    dst = dst.astype(np.float32).astype(np.float64)
    t = transform.ProjectiveTransform()
    assert t.estimate(src, dst)
    M = t.params / t.params[2, 2]
    srcs.append(src)
    dsts.append(dst)
    Ms.append(M)
out["persp_src"] = np.array(srcs)
out["persp_dst"] = np.array(dsts)
out["persp_M"] = np.array(Ms)

imgs = []
for i in range(3):
    base = ndi.gaussian_filter(np.random.default_rng(40 + i).standard_normal((80, 170)), 1.2 + i)
    base = (base - base.min()) / (base.max() - base.min())
    img = np.clip(base * 255 + np.random.default_rng(50 + i).integers(-20, 20, base.shape), 0, 255).astype(np.uint8)
    imgs.append(img)
out["warp_imgs"] = np.array(imgs)
warp_out, warp_dsize = [], []
for i, M in enumerate(Ms):
    img = imgs[i % 3]
    dw, dh = int(dsts[i][1, 0]), int(dsts[i][2, 1])
    Mi = np.linalg.inv(M)
    xs, ys = np.meshgrid(np.arange(dw, dtype=np.float64), np.arange(dh, dtype=np.float64))
    X = Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]
    Y = Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]
    Wd = Mi[2, 0] * xs + Mi[2, 1] * ys + Mi[2, 2]
    cx = np.rint(X / Wd * 32.0) / 32.0
    cy = np.rint(Y / Wd * 32.0) / 32.0
    v = ndi.map_coordinates(img.astype(np.float64), [cy, cx], order=1, mode="grid-constant", cval=0.0, prefilter=False)
    o = np.zeros((31, 200), np.uint8)
    o[:dh, :dw] = np.floor(v + 0.5).astype(np.uint8)
    warp_out.append(o)
    warp_dsize.append((dw, dh))
out["warp_out"] = np.array(warp_out)
out["warp_dsize"] = np.array(warp_dsize)

# ------------------------------------------------------------------------------------------------
# 5. resize (bilinear, half-pixel centres, edge clamp) and RGB -> gray
# ------------------------------------------------------------------------------------------------
rz_in = np.random.default_rng(60).integers(0, 256, (24, 36, 3)).astype(np.uint8)
rz_in[:, :12] = (ndi.gaussian_filter(np.random.default_rng(61).standard_normal((24, 12, 3)), (2, 2, 0)) * 200 + 128).clip(0, 255)
out["resize_in"] = rz_in
for tag, (dh, dw) in {"x2": (48, 72), "x1p5": (36, 54), "x4_3": (32, 48), "aniso": (31, 200)}.items():
    out["resize_" + tag] = transform.resize(rz_in.astype(np.float64), (dh, dw, 3), order=1, mode="edge", anti_aliasing=False,
                                            preserve_range=True)
gray_in = np.random.default_rng(70).integers(0, 256, (40, 50, 3)).astype(np.uint8)
out["gray_in"] = gray_in
out["gray_out"] = np.asarray(Image.fromarray(gray_in).convert("L"))

np.savez_compressed(os.path.join(HERE, "thirdparty_golden.npz"), **out)
print("wrote thirdparty_golden.npz", {k: v.shape for k, v in out.items()})

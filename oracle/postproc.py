"""Oracle: heat-map -> word boxes (numpy / scipy, CPU).  TEST INFRASTRUCTURE ONLY.

Restates ``keras_ocr.detection.getBoxes`` (reference ``keras_ocr/detection.py:207-287``).
Every ``cv2.*`` call there is a third-party [3P] behaviour that cannot be executed in this
environment (OpenCV is installed in neither interpreter of the image).  Status: **cross-checked by independent
implementation** -- label order / areas / bounding boxes against skimage.measure.label + regionprops, the k x k
rectangle dilation (odd AND even k) against scipy.ndimage.maximum_filter and skimage.morphology, the fragment choice
against skimage 8-connectivity labels, hull vertices against Qhull, the min-area rectangle against a float64
brute-force search (tests/golden/make_golden_3p.py -> tests/test_thirdparty_crosscheck_cpu.py).  The one rule that
rests on reading OpenCV's sources rather than on an executable cross-check is WHICH contour ``findContours`` lists
first when a component was split into several fragments (siblings are inserted at the head of the list, so it is
the fragment found last by the raster scan).  The restatement follows the documented semantics:

  cv2.threshold(THRESH_BINARY)            dst = maxval if src > thresh else 0   (strict >)
  connectedComponentsWithStats(conn=4)    labels in raster order of first pixel
  getStructuringElement(RECT,(k,k))+dilate anchor (k//2,k//2); out(p) = max src(p + j - anchor),
                                          j in [0,k)^2, border ignored, ROI view is isolated
  findContours(RETR_TREE, SIMPLE)[-2][0]  first contour of the list = the LAST top-level outer
                                          contour met by the raster scan (siblings are pushed
                                          at the head of the list), i.e. the 8-connected
                                          fragment whose raster-first pixel comes last
  minAreaRect / boxPoints                 min-area enclosing rectangle with one side collinear
                                          with a hull edge; corners clockwise on screen

Design choice that makes the restatement reproducible bit-for-bit by the HIP kernels: the
candidate rectangles are compared in exact integer arithmetic (hull vertices are integer
pixel coordinates) and the chosen rectangle's corners are formed from exact integer
numerators with one float64 division, then rounded to float32.  OpenCV does the same
geometry in float32 (corners differ by ~1e-4 px) and breaks exact area ties by float noise.
"""
import math

import numpy as np
from scipy import ndimage

_CROSS = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=bool)
_FULL = np.ones((3, 3), dtype=bool)


def dilate_rect(roi, k):
    """cv2.dilate(roi, getStructuringElement(MORPH_RECT, (k, k))) on a boolean ROI."""
    a = k // 2
    h, w = roi.shape
    out = np.zeros_like(roi)
    ys, xs = np.nonzero(roi)
    for y, x in zip(ys, xs):
        # a source pixel at (y,x) reaches outputs p with p + j - a = (y,x), j in [0,k)
        y0, y1 = max(y + a - (k - 1), 0), min(y + a, h - 1)
        x0, x1 = max(x + a - (k - 1), 0), min(x + a, w - 1)
        if y0 <= y1 and x0 <= x1:
            out[y0:y1 + 1, x0:x1 + 1] = True
    return out


def convex_hull_rows(pts):
    """Convex hull (strict, no collinear vertices) of integer points.

    Returns vertices clockwise on screen (y down), starting at the top-most then
    left-most point.  Andrew's monotone chain keyed on (y, x)."""
    pts = sorted(set((int(y), int(x)) for x, y in pts))  # (y, x) order
    P = [(x, y) for y, x in pts]
    if len(P) <= 1:
        return P

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    # left chain (top -> bottom along the left side) and right chain
    lower = []
    for p in P:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) >= 0:
            lower.pop()
        lower.append(p)
    upper = []
    for p in reversed(P):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) >= 0:
            upper.pop()
        upper.append(p)
    hull = lower[:-1] + upper[:-1]  # starts at top-left-most; direction fixed below
    # orientation: make it clockwise on screen (x right, y down) == positive shoelace in
    # image coordinates
    area2 = sum(hull[i][0] * hull[(i + 1) % len(hull)][1] - hull[(i + 1) % len(hull)][0] * hull[i][1]
                for i in range(len(hull)))
    if area2 < 0:
        hull = [hull[0]] + hull[1:][::-1]
    return hull


def min_area_box(hull):
    """Exact min-area rectangle over hull edges -> 4 corners (float32), clockwise on screen.

    hull: integer vertices, clockwise on screen, hull[0] top-most/left-most.  Ties in area
    go to the smallest edge index.  Degenerate (<=2 vertices): OpenCV's n==2 / n==1 cases.
    """
    n = len(hull)
    if n == 1:
        x, y = hull[0]
        return np.array([[x, y]] * 4, dtype=np.float32)
    if n == 2:
        (x0, y0), (x1, y1) = hull
        # RotatedRect of width |p0p1|, height 0 -> boxPoints gives the two end points twice
        return np.array([[x0, y0], [x0, y0], [x1, y1], [x1, y1]], dtype=np.float32)
    best = None
    for i in range(n):
        x0, y0 = hull[i]
        x1, y1 = hull[(i + 1) % n]
        dx, dy = x1 - x0, y1 - y0
        L = dx * dx + dy * dy
        us = [px * dx + py * dy for px, py in hull]
        vs = [-px * dy + py * dx for px, py in hull]
        umin, umax, vmin, vmax = min(us), max(us), min(vs), max(vs)
        num = (umax - umin) * (vmax - vmin)
        if best is None or num * best[1] < best[0] * L:  # num/L < best_num/best_L, exact
            best = (num, L, dx, dy, umin, umax, vmin, vmax)
    _, L, dx, dy, umin, umax, vmin, vmax = best
    corners = []
    for u, v in ((umin, vmin), (umax, vmin), (umax, vmax), (umin, vmax)):
        x = np.float64(u * dx - v * dy) / np.float64(L)
        y = np.float64(u * dy + v * dx) / np.float64(L)
        corners.append([np.float32(x), np.float32(y)])
    return np.array(corners, dtype=np.float32)


def first_contour_fragment(seg):
    """Mask of the fragment that ``findContours(...)[-2][0]`` traces (see module docstring)."""
    lab, n = ndimage.label(seg, structure=_FULL)  # 8-connectivity, raster-order labels
    if n == 0:
        return None
    return lab == n  # the fragment whose raster-first pixel comes last


def get_boxes(y_pred, detection_threshold=0.7, text_threshold=0.4, link_threshold=0.4, size_threshold=10,
              return_debug=False):
    """detection.py:207-287.  y_pred: (N,h,w,2) float32.  Returns list of (n_i,4,2) float32
    arrays (``np.array([])`` for an image without boxes, detection.py:286)."""
    box_groups = []
    debug = []
    f32 = np.float32
    for y_pred_cur in y_pred:
        textmap = np.asarray(y_pred_cur[..., 0], dtype=np.float32)
        linkmap = np.asarray(y_pred_cur[..., 1], dtype=np.float32)
        img_h, img_w = textmap.shape
        text_score = textmap > f32(text_threshold)
        link_score = linkmap > f32(link_threshold)
        labels, n_components = ndimage.label(text_score | link_score, structure=_CROSS)
        both = text_score & link_score
        objs = ndimage.find_objects(labels)
        boxes = []
        dbg = []
        for component_id in range(1, n_components + 1):
            sl = objs[component_id - 1]
            sub = labels[sl] == component_id
            size = int(sub.sum())
            if size < size_threshold:
                continue
            if textmap[sl][sub].max() < f32(detection_threshold):
                continue
            y, x = sl[0].start, sl[1].start
            h, w = sl[0].stop - y, sl[1].stop - x
            niter = int(math.sqrt(size * min(w, h) / (w * h)) * 2)
            sx, sy = max(x - niter, 0), max(y - niter, 0)
            ex, ey = min(x + w + niter + 1, img_w), min(y + h + niter + 1, img_h)
            roi = np.zeros((ey - sy, ex - sx), dtype=bool)
            roi[y - sy:y - sy + h, x - sx:x - sx + w] = sub
            roi &= ~both[sy:ey, sx:ex]
            dil = dilate_rect(roi, 1 + niter)
            frag = first_contour_fragment(dil)
            if frag is None:
                # the reference indexes contours[0] of an empty list here
                raise IndexError("list index out of range")
            fy, fx = np.nonzero(frag)
            fx = fx + sx
            fy = fy + sy
            hull = convex_hull_rows(np.stack([fx, fy], 1))
            box = min_area_box(hull)
            # np.linalg.norm on float32 pairs, written out: sqrt(dx*dx + dy*dy) in float32
            dw, dh = box[0] - box[1], box[1] - box[2]
            w_ = np.sqrt(dw[0] * dw[0] + dw[1] * dw[1])
            h_ = np.sqrt(dh[0] * dh[0] + dh[1] * dh[1])
            box_ratio = max(w_, h_) / (min(w_, h_) + f32(1e-5))
            if abs(f32(1) - box_ratio) <= f32(0.1):
                l, r = fx.min(), fx.max()
                t, b = fy.min(), fy.max()
                box = np.array([[l, t], [r, t], [r, b], [l, b]], dtype=np.float32)
            else:
                box = np.array(np.roll(box, 4 - box.sum(axis=1).argmin(), 0))
            boxes.append(f32(2) * box)
            dbg.append(dict(component=component_id, size=size, niter=niter, hull=hull))
        box_groups.append(np.array(boxes))
        debug.append(dbg)
    if return_debug:
        return box_groups, debug
    return box_groups

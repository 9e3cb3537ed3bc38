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


# ---------------------------------------------------------------------------------------------------------------------
# cv2.minAreaRect + cv2.boxPoints in OpenCV's OWN float32 arithmetic (round 6).  [3P], parity unpinned:
# OpenCV is not installed here and its sources are not under /root/reference; this restates the published algorithm of
# modules/imgproc/src/rotcalipers.cpp (`rotatingCalipers`, mode CALIPERS_MINAREARECT, and `cv::minAreaRect`) and of
# `RotatedRect::points` (modules/core/src/types.cpp), OpenCV 4.x, operation by operation in float32 where OpenCV computes in
# float and in float64 where it computes in double.  It exists to QUANTIFY what min_area_box's exact arithmetic hides: the
# corner deviation of cv2's float32 geometry and how often `int()` in tools.get_rotated_width_height (tools.py:49-57) turns
# it into a crop of a different size (scripts/minarearect_deviation.py -> DESIGN.md section 4).  tests/golden/make_golden_real.py
# is the pin: on a host with cv2 its getBoxes fixture decides between the two.
# ---------------------------------------------------------------------------------------------------------------------
def min_area_box_cv32(hull):
    """hull: the strict convex hull as integer (x, y) vertices in the order ``cv2.convexHull(points, clockwise=False)``
    lists them for ``minAreaRect`` -- ``convex_hull_rows`` order reversed is passed by ``get_boxes_cv32``.  Returns the four
    ``boxPoints`` corners (float32)."""
    f32 = np.float32
    n = len(hull)
    pts = np.asarray(hull, dtype=np.float32).reshape(n, 2)
    if n == 1:
        return np.repeat(pts, 4, 0)
    if n == 2:
        # cv::minAreaRect, n == 2: centre = midpoint, width = |p1 - p0|, height 0, angle = atan2(dy, dx)
        cx, cy = (pts[0, 0] + pts[1, 0]) * f32(0.5), (pts[0, 1] + pts[1, 1]) * f32(0.5)
        dx, dy = np.float64(pts[1, 0] - pts[0, 0]), np.float64(pts[1, 1] - pts[0, 1])
        width, height = f32(math.sqrt(dx * dx + dy * dy)), f32(0)
        angle = f32(math.atan2(dy, dx))
    else:
        vect = np.zeros((n, 2), np.float32)
        inv_len = np.zeros(n, np.float32)
        left = bottom = right = top = 0
        left_x = right_x = pts[0, 0]
        top_y = bottom_y = pts[0, 1]
        pt0 = pts[0]
        for i in range(n):
            if pt0[0] < left_x:
                left_x, left = pt0[0], i
            if pt0[0] > right_x:
                right_x, right = pt0[0], i
            if pt0[1] > top_y:
                top_y, top = pt0[1], i
            if pt0[1] < bottom_y:
                bottom_y, bottom = pt0[1], i
            pt = pts[(i + 1) % n]
            dx, dy = np.float64(pt[0]) - np.float64(pt0[0]), np.float64(pt[1]) - np.float64(pt0[1])  # double dx, dy
            vect[i] = (f32(dx), f32(dy))
            inv_len[i] = f32(1.0 / math.sqrt(dx * dx + dy * dy))
            pt0 = pt
        orientation = f32(0)
        ax, ay = np.float64(vect[n - 1, 0]), np.float64(vect[n - 1, 1])
        for i in range(n):
            bx, by = np.float64(vect[i, 0]), np.float64(vect[i, 1])
            convexity = ax * by - ay * bx
            if convexity != 0:
                orientation = f32(1) if convexity > 0 else f32(-1)
                break
            ax, ay = bx, by
        assert orientation != 0
        base_a, base_b = orientation, f32(0)
        seq = [bottom, right, top, left]
        minarea = f32(np.finfo(np.float32).max)
        best = None
        for _ in range(n):
            dp = [base_a * vect[seq[0], 0] + base_b * vect[seq[0], 1],
                  -base_b * vect[seq[1], 0] + base_a * vect[seq[1], 1],
                  -base_a * vect[seq[2], 0] - base_b * vect[seq[2], 1],
                  base_b * vect[seq[3], 0] - base_a * vect[seq[3], 1]]
            maxcos = dp[0] * inv_len[seq[0]]
            main = 0
            for i in range(1, 4):
                cosalpha = dp[i] * inv_len[seq[i]]
                if cosalpha > maxcos:
                    main, maxcos = i, cosalpha
            pindex = seq[main]
            lead_x, lead_y = vect[pindex, 0] * inv_len[pindex], vect[pindex, 1] * inv_len[pindex]
            base_a, base_b = ((lead_x, lead_y), (lead_y, -lead_x), (-lead_x, -lead_y), (-lead_y, lead_x))[main]
            seq[main] = (seq[main] + 1) % n
            dx, dy = pts[seq[1], 0] - pts[seq[3], 0], pts[seq[1], 1] - pts[seq[3], 1]
            width = dx * base_a + dy * base_b
            dx, dy = pts[seq[2], 0] - pts[seq[0], 0], pts[seq[2], 1] - pts[seq[0], 1]
            height = -dx * base_b + dy * base_a
            area = width * height
            if area <= minarea:
                minarea = area
                best = (seq[3], base_a, width, base_b, height, seq[0])
        li, a1, width, b1, height, bi = best
        a2, b2 = -b1, a1
        c1 = a1 * pts[li, 0] + pts[li, 1] * b1
        c2 = a2 * pts[bi, 0] + pts[bi, 1] * b2
        idet = f32(1) / (a1 * b2 - a2 * b1)
        px, py = (c1 * b2 - c2 * b1) * idet, (a1 * c2 - a2 * c1) * idet
        o1x, o1y, o2x, o2y = a1 * width, b1 * width, a2 * height, b2 * height
        cx, cy = px + (o1x + o2x) * f32(0.5), py + (o1y + o2y) * f32(0.5)
        width = f32(math.sqrt(np.float64(o1x) * np.float64(o1x) + np.float64(o1y) * np.float64(o1y)))
        height = f32(math.sqrt(np.float64(o2x) * np.float64(o2x) + np.float64(o2y) * np.float64(o2y)))
        angle = f32(math.atan2(np.float64(o1y), np.float64(o1x)))
    angle = f32(np.float64(angle) * 180 / math.pi)     # box.angle = (float)(box.angle*180/CV_PI)
    # RotatedRect::points
    ang = np.float64(angle) * math.pi / 180.0
    b = f32(math.cos(ang)) * f32(0.5)
    a = f32(math.sin(ang)) * f32(0.5)
    p0 = (cx - a * height - b * width, cy + b * height - a * width)
    p1 = (cx + a * height - b * width, cy - b * height - a * width)
    p2 = (f32(2) * cx - p0[0], f32(2) * cy - p0[1])
    p3 = (f32(2) * cx - p1[0], f32(2) * cy - p1[1])
    return np.array([p0, p1, p2, p3], dtype=np.float32)


def box_from_hull(hull, fx, fy, cv32=False):
    """detection.py:273-285 on one component's hull: boxPoints(minAreaRect), the diamond rule, the roll, x 2."""
    f32 = np.float32
    # cv2.convexHull(points, clockwise=False) lists the hull in the opposite orientation of convex_hull_rows
    box = min_area_box_cv32([hull[0]] + hull[1:][::-1]) if cv32 else min_area_box(hull)
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
    return f32(2) * box

"""CPU suite, part 3: known-answer checks of the oracle's [3P] restatements (closed-form cases
where the published algorithm's answer is known without running OpenCV/TensorFlow)."""
import numpy as np

from tests import synth


def test_resize_x2_is_quarter_three_quarter_blend():
    """cv2.resize x2: interior samples are exactly (3a+b)/4 rounded as OpenCV's fixed point does."""
    from oracle import tools

    img = np.zeros((1, 4, 3), np.uint8)
    img[0, :, 0] = [0, 100, 200, 40]
    out = tools.cv_resize_linear_u8(img, (8, 2))
    assert out.shape == (2, 8, 3)
    assert list(out[0, :, 0]) == [0, 25, 75, 125, 175, 160, 80, 40]
    assert np.array_equal(out[0], out[1])


def test_gray_coefficients():
    from oracle import tools

    px = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]]], np.uint8)
    assert list(tools.rgb2gray_u8(px)[0]) == [255, 76, 150, 29, 18]


def test_identity_warp_returns_the_gray_image():
    from oracle import tools

    rng = np.random.default_rng(0)
    gray = rng.integers(0, 256, (31, 200), dtype=np.uint8)
    box = np.array([[0, 0], [200, 0], [200, 31], [0, 31]], np.float32)
    assert np.array_equal(tools.warp_box(gray, box, 31, 200), gray)


def test_perspective_transform_maps_the_quad():
    from oracle import tools

    src = np.array([[12, 7], [90, 20], [80, 60], [5, 40]], np.float32)
    dst = np.array([[0, 0], [200, 0], [200, 31], [0, 31]], np.float32)
    M = tools.get_perspective_transform(src, dst)
    p = np.concatenate([src, np.ones((4, 1))], 1) @ M.T
    np.testing.assert_allclose(p[:, :2] / p[:, 2:], dst, atol=1e-9)
    Mi = np.array(tools.invert3(M))
    np.testing.assert_allclose(Mi @ M / (Mi @ M)[2, 2], np.eye(3), atol=1e-9)


def test_get_boxes_branches():
    from oracle import postproc

    y = synth.heatmap_batch()
    boxes, dbg = postproc.get_boxes(y, return_debug=True)
    assert [len(b) for b in boxes] == [4, 0, 2, 2]
    assert boxes[1].shape == (0,)  # np.array([]) (detection.py:286)
    # image 0, 4th component is an isolated character: the diamond branch gives an axis-aligned box
    b = boxes[0][3]
    assert b[0, 1] == b[1, 1] and b[1, 0] == b[2, 0] and b[2, 1] == b[3, 1] and b[3, 0] == b[0, 0]
    # every box is clockwise on screen and starts at its min(x+y) corner unless axis-aligned
    for grp in boxes:
        for q in grp:
            x, yy = q[:, 0], q[:, 1]
            assert (x * np.roll(yy, -1) - np.roll(x, -1) * yy).sum() > 0
            assert (q.sum(1)).argmin() == 0
    # niter >= 2 always for a 4-connected component of >= 10 px (dilation never degenerates)
    assert all(d["niter"] >= 2 for grp in dbg for d in grp)


def test_min_area_box_exact_cases():
    from oracle import postproc

    hull = postproc.convex_hull_rows(np.array([[0, 0], [4, 0], [4, 2], [0, 2], [2, 1]]))
    assert hull == [(0, 0), (4, 0), (4, 2), (0, 2)]
    box = postproc.min_area_box(hull)
    assert np.array_equal(box, np.array([[0, 0], [4, 0], [4, 2], [0, 2]], np.float32))
    # a 45-degree rectangle: area 2*sqrt2 * sqrt2 = 4 beats the axis-aligned 3x3 = 9
    hull = postproc.convex_hull_rows(np.array([[1, 0], [3, 2], [2, 3], [0, 1]]))
    box = postproc.min_area_box(hull)
    assert sorted(map(tuple, box.tolist())) == [(0.0, 1.0), (1.0, 0.0), (2.0, 3.0), (3.0, 2.0)]


def test_dilate_anchor_even_kernel():
    from oracle import postproc

    roi = np.zeros((7, 9), bool)
    roi[3, 4] = True
    out = postproc.dilate_rect(roi, 4)  # anchor 2: reaches 1 left/up, 2 right/down
    ys, xs = np.nonzero(out)
    assert (ys.min(), ys.max(), xs.min(), xs.max()) == (2, 5, 3, 6)


def test_ctc_greedy_rules():
    from oracle import crnn

    T, C = 48, 37
    path = [36, 1, 1, 36, 1, 2, 2, 2, 36, 36, 3] + [36] * 37
    p = np.full((1, T, C), 1e-3, np.float32)
    p[0, np.arange(T), path] = 0.9
    lab = crnn.ctc_greedy_decode(p)
    assert list(lab[0][:4]) == [1, 1, 2, 3] and (lab[0][4:] == -1).all()
    assert crnn.decode_strings(lab) == ["1123"]


def test_stn_identity_quirk():
    """theta = identity samples x = 0.5*(x_t+1)*W: the last column falls on x = W and cancels to 0
    (recognition.py:109, 144-152) — the reference quirk the sampler must keep."""
    import torch
    from oracle import crnn

    x = torch.arange(2 * 3 * 4, dtype=torch.float32).reshape(1, 3, 4, 2) + 1
    out = crnn.stn_transform(x, torch.tensor([[1.0, 0, 0, 0, 1.0, 0]]))
    assert torch.equal(out[0, 0, 0], x[0, 0, 0])
    assert float(out[0, :, -1].abs().max()) == 0.0 and float(out[0, -1].abs().max()) == 0.0


def test_warp_association_orders_differ_in_few_pixels():
    """VERDICT r02 (weak 1): the oracle (and the device code) evaluate the warp's coordinate numerators as
    ``(Mi0 x + Mi1 y) + Mi2`` per pixel; OpenCV is believed to form ``X0 = (Mi0 bx + Mi1 y) + Mi2`` per 64-column block
    and add ``Mi0 x1`` per pixel.  Neither can be executed against cv2 here, so this documents the SIZE of that
    unpinned bit: over rotated / perspective boxes on a noise image the two orders give crops that differ in at most
    a handful of pixels, each by one grey level."""
    from oracle import tools as otools

    rng = np.random.default_rng(77)
    gray = rng.integers(0, 256, (400, 600), dtype=np.uint8)
    n_px = n_diff = 0
    worst = 0
    for k in range(60):
        cx, cy = rng.uniform(120, 480), rng.uniform(100, 300)
        w, h = rng.uniform(40, 180), rng.uniform(12, 40)
        th = rng.uniform(-0.6, 0.6)
        c, s_ = np.cos(th), np.sin(th)
        base = np.array([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]])
        box = (base @ np.array([[c, s_], [-s_, c]]) + [cx, cy] + rng.uniform(-1.5, 1.5, (4, 2))).astype(np.float32)
        _, _, _, M, dsize, _ = otools.warp_box_params(box, 31, 200, use_min_rect=False)
        a = otools.warp_perspective_u8(gray, M, dsize, assoc="pixel")
        b = otools.warp_perspective_u8(gray, M, dsize, assoc="blockwise")
        d = np.abs(a.astype(int) - b.astype(int))
        n_px += a.size
        n_diff += int((d > 0).sum())
        worst = max(worst, int(d.max()) if d.size else 0)
    print(f"warp association: {n_diff} of {n_px} crop pixels differ between the two orders, max |diff| {worst}")
    assert n_px > 100000
    assert n_diff <= 2e-3 * n_px      # measured: a few 1e-5
    assert worst <= 2


def test_ties_the_reference_leaves_to_opencv_are_pinned_in_the_oracle():
    """VERDICT r03 item 6c.  Two places of detection.getBoxes (detection.py:273-285) are decided by float noise inside
    OpenCV when the geometry is exactly symmetric; the oracle (and, bit for bit, the device code: tests/test_postproc_gpu.py)
    makes a DETERMINISTIC choice there.  OpenCV's own choice is unobserved (cv2 is absent from this image): these
    assertions document which answer the restatement gives, so that a real-library run can be compared one day.

    1. np.roll(box, 4 - argmin(x + y)): a 45-degree rectangle has TWO corners with the same x + y (its top and its left
       corner when the long side runs down-right).  numpy's argmin takes the FIRST minimum in boxPoints order; the oracle
       lists the corners clockwise from (umin, vmin) of the chosen hull edge, so the tie goes to whichever of the two that
       order meets first.
    2. minAreaRect: an exactly equal-area tie between two hull edges (a square rotated by 45 degrees inside its axis-aligned
       twin cannot happen on a hull, but a symmetric hexagon gives two edges with the same enclosing rectangle area) goes to
       the FIRST hull edge in clockwise order from the top-most / left-most vertex."""
    from oracle import postproc

    # 1. a 45-degree 2*sqrt2 x sqrt2 rectangle: corners (1,0) (3,2) (2,3) (0,1); x + y = 1 for (1,0) AND (0,1)
    hull = postproc.convex_hull_rows(np.array([[1, 0], [3, 2], [2, 3], [0, 1]]))
    box = postproc.min_area_box(hull)
    sums = box.sum(axis=1)
    assert sorted(sums.tolist()) == [1.0, 1.0, 5.0, 5.0]                       # the tie exists in exact arithmetic
    first = int(sums.argmin())
    rolled = np.roll(box, 4 - first, 0)
    assert tuple(rolled[0]) == (1.0, 0.0), rolled                               # the oracle's answer: the TOP corner starts the box
    x, y = rolled[:, 0], rolled[:, 1]
    assert (x * np.roll(y, -1) - np.roll(x, -1) * y).sum() > 0                  # and the order stays clockwise on screen
    # 2. a hexagon symmetric under the swap of its two slanted edge directions: edges (0,0)->(2,0) ... give rectangles of
    #    equal area for the edge pair (4,1)->(3,3) / (3,3)->... mirrored; the first in hull order wins
    pts = np.array([[1, 0], [3, 0], [4, 2], [3, 4], [1, 4], [0, 2]])
    hull = postproc.convex_hull_rows(pts)
    assert hull[0] == (1, 0)                                                    # top-most, then left-most vertex first
    areas = []
    for i in range(len(hull)):
        (x0, y0), (x1, y1) = hull[i], hull[(i + 1) % len(hull)]
        dx, dy = x1 - x0, y1 - y0
        us = [px * dx + py * dy for px, py in hull]
        vs = [-px * dy + py * dx for px, py in hull]
        areas.append(((max(us) - min(us)) * (max(vs) - min(vs)), dx * dx + dy * dy))
    best = min(a / l for a, l in areas)
    winners = [i for i, (a, l) in enumerate(areas) if a * 1.0 / l == best]
    assert len(winners) >= 2                                                    # a genuine exact tie
    box = postproc.min_area_box(hull)
    i = winners[0]                                                              # the oracle takes the first edge of the tie
    (x0, y0), (x1, y1) = hull[i], hull[(i + 1) % len(hull)]
    dx, dy = x1 - x0, y1 - y0
    # the box has a side parallel to that edge
    side = box[1] - box[0]
    assert abs(float(side[0]) * dy - float(side[1]) * dx) < 1e-5

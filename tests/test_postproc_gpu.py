"""Heat-map -> boxes on the GPU (csrc/postproc.hip through kocr_get_boxes) vs the CPU oracle
(oracle/postproc.py, restating detection.py:207-287).

Bar: identical box counts and order (integer work: thresholding, connected components, areas,
ROI, dilation, fragment choice, hull, exact min-area edge) and bit-identical float32 corner
coordinates (both sides form the corners from exact integer numerators with one IEEE float64
division; the HIP TU is built with -ffp-contract=off)."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _check(got, want):
    assert len(got) == len(want)
    for g, w_ in zip(got, want):
        assert g.shape == w_.shape, (g.shape, w_.shape)
        if len(w_):
            assert np.array_equal(g, w_.astype(np.float32)), np.abs(g - w_).max()


def test_boxes_match_oracle_all_branches(ctx):
    from oracle import postproc

    y = synth.heatmap_batch()
    want = postproc.get_boxes(y)
    got = ctx.get_boxes(y)
    assert [len(b) for b in want] == [len(b) for b in got]
    assert sum(len(b) for b in want) >= 6
    assert want[1].shape == (0,) and got[1].shape == (0,)  # np.array([]) for an empty image
    _check(got, want)


@pytest.mark.parametrize("thr", [(0.7, 0.4, 0.4, 10), (0.5, 0.3, 0.5, 4), (0.9, 0.6, 0.2, 30)])
def test_threshold_arguments(ctx, thr):
    from oracle import postproc

    y = synth.heatmap_batch()
    kw = dict(detection_threshold=thr[0], text_threshold=thr[1], link_threshold=thr[2], size_threshold=thr[3])
    _check(ctx.get_boxes(y, **kw), postproc.get_boxes(y, **kw))


def test_random_smooth_fields(ctx):
    """Blob soup from low-pass filtered noise: irregular components, holes, border contact."""
    from scipy import ndimage
    from oracle import postproc

    rng = np.random.default_rng(42)
    ys = []
    for i in range(3):
        f = ndimage.gaussian_filter(rng.standard_normal((2, 150, 130)), (0, 3.0 + i, 3.0 + i))
        f = f / np.abs(f).max() * 1.6
        ys.append(np.moveaxis(f, 0, -1))
    y = np.stack(ys).astype(np.float32)
    want = postproc.get_boxes(y)
    got = ctx.get_boxes(y, cap=8)  # small cap exercises the capacity-retry path
    assert sum(len(b) for b in want) > 10
    _check(got, want)


def test_full_size_properties(ctx):
    """BASELINE cfg4 heat-map size (768x768): size-independent properties instead of the
    (slow) oracle: boxes inside the map, clockwise orientation, identical on a second run."""
    rng = np.random.default_rng(9)
    words = [(rng.uniform(60, 700), rng.uniform(30, 740), int(rng.integers(2, 8)), 11, 4.5,
              rng.uniform(-0.3, 0.3), 1.0) for _ in range(40)]
    y = synth.word_heatmap(768, 768, words)[None]
    a = ctx.get_boxes(y)
    b = ctx.get_boxes(y)
    assert len(a[0]) >= 20
    assert np.array_equal(a[0], b[0])
    bx = a[0]
    assert bx.min() >= -1e-3 and bx.max() <= 2 * 768
    # shoelace > 0 in image coordinates == clockwise on screen
    x, yy = bx[..., 0], bx[..., 1]
    area2 = (x * np.roll(yy, -1, 1) - np.roll(x, -1, 1) * yy).sum(1)
    assert (area2 > 0).all()


def test_large_noisy_field_vs_oracle(ctx):
    """400x400 blob soup with thousands of sub-threshold specks: stresses the union-find across
    many workgroups / XCDs, the ordered compaction and the canvas packing."""
    from scipy import ndimage
    from oracle import postproc

    rng = np.random.default_rng(123)
    f = ndimage.gaussian_filter(rng.standard_normal((2, 400, 400)), (0, 4.0, 4.0))
    f = f / np.abs(f).max() * 1.5 + rng.normal(0, 0.08, f.shape)
    y = np.moveaxis(f, 0, -1)[None].astype(np.float32)
    want = postproc.get_boxes(y)
    got = ctx.get_boxes(y)
    assert len(want[0]) > 20
    _check(got, want)


def test_detect_entry_point_equals_two_step(ctx, craft_weights):
    """kocr_detect (heat-maps stay in HBM) == kocr_craft_forward + kocr_get_boxes."""
    ctx.load_craft(craft_weights)
    img = synth.text_page(96, 128, 6, seed=8)[None]
    heat = ctx.craft_forward(img)
    # thresholds chosen inside the random-init head's output range so that boxes exist
    t = float(np.quantile(heat[..., 0], 0.9))
    kw = dict(detection_threshold=t, text_threshold=t * 0.9, link_threshold=1e9, size_threshold=4)
    a = ctx.get_boxes(heat, **kw)
    b = ctx.detect(img, **kw)
    assert len(a[0]) > 0 and len(a[0]) == len(b[0]) and np.array_equal(a[0], b[0])


def _hull_shapes():
    """One 700 x 1000 heat-map whose components aim at the hull kernel (K13): a digital diamond of 601 rows (every row
    carries two points that can be hull vertices: more candidates than the kernel's LDS chains hold -> the global-scratch
    path), an ellipse (a hull of many vertices in LDS), a 1-pixel line of 650 rows (reduces to two candidates), a small
    diamond (collinear boundary points), a thin slanted bar and a small square."""
    h, w = 700, 1000
    yy, xx = np.mgrid[0:h, 0:w]
    text = np.zeros((h, w), np.float32)
    text[np.abs(xx - 310) + np.abs(yy - 320) <= 300] = 1.0                      # large diamond
    text[((xx - 780) / 120.0) ** 2 + ((yy - 250) / 200.0) ** 2 <= 1.0] = 1.0    # ellipse
    text[30:680, 950] = 1.0                                                    # vertical line
    text[np.abs(xx - 760) + np.abs(yy - 560) <= 40] = 1.0                       # small diamond
    text[(np.abs((yy - 560) - 0.37 * (xx - 880)) <= 2.0) & (np.abs(xx - 880) <= 45)] = 1.0  # slanted bar
    text[660:666, 640:646] = 1.0                                               # square
    return np.stack([text, np.zeros_like(text)], -1)[None]


def test_hull_kernel_shapes_vs_oracle(ctx):
    from oracle import postproc

    y = _hull_shapes()
    want = postproc.get_boxes(y)
    got = ctx.get_boxes(y)
    assert len(want[0]) == 6
    _check(got, want)

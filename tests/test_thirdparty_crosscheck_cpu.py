"""CPU suite: the [3P] halves of the oracle against INDEPENDENT statements of the same operations.

``tests/golden/thirdparty_golden.npz`` is produced by ``tests/golden/make_golden_3p.py`` under the image's
second interpreter (scikit-image / scipy / Pillow; no code shared with ``oracle/`` or the kernels), the STN
vectors in ``reference_golden.npz`` by executing the reference's own ``recognition._transform``
(``tests/golden/make_golden.py``).  The recurrent / convolutional part of the CRNN is re-stated here with
``torch.nn`` MODULES loaded with Keras-ordered weights (the oracle uses hand-written loops / functional
calls), the CTC rule with itertools.  OpenCV / TensorFlow themselves are installed nowhere in this image;
this is the strongest pin available (DESIGN.md section 4 lists what each line is pinned to).
"""
import itertools
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g3():
    return np.load(os.path.join(HERE, "golden", "thirdparty_golden.npz"))


@pytest.fixture(scope="module")
def gref():
    return np.load(os.path.join(HERE, "golden", "reference_golden.npz"))


# ---------------------------------------------------------------------------------------------------
# getBoxes' OpenCV calls (detection.py:221-273)
# ---------------------------------------------------------------------------------------------------
def test_ccl_label_order_area_bbox_vs_skimage(g3):
    """connectedComponentsWithStats(connectivity=4): the oracle's labelling (scipy.ndimage.label + find_objects,
    oracle/postproc.py) against skimage.measure.label / regionprops: same components, same ORDER (raster order
    of the first pixel = OpenCV's label order = the order of the returned boxes), same area and bbox."""
    from scipy import ndimage
    from oracle import postproc

    for mask, (h, w), table in zip(g3["ccl_masks"], g3["ccl_shapes"], g3["ccl_tables"]):
        m = mask[:h, :w]
        lab, n = ndimage.label(m, structure=postproc._CROSS)  # pylint: disable=protected-access
        want = table[table[:, 0] >= 0]
        assert n == len(want)
        objs = ndimage.find_objects(lab)
        for cid in range(1, n + 1):
            sl = objs[cid - 1]
            sub = lab[sl] == cid
            ys, xs = np.nonzero(lab == cid)
            first = int(ys[0]) * w + int(xs[0])
            row = [first, int(sub.sum()), sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start]
            assert row == list(want[cid - 1]), (cid, row, want[cid - 1])


def test_rect_dilation_even_and_odd_vs_scipy_and_skimage(g3):
    """cv2.dilate with getStructuringElement(MORPH_RECT,(k,k)), anchor k//2 (asymmetric window for even k)."""
    from oracle import postproc

    for k in range(1, 10):
        got = postproc.dilate_rect(g3["dil_rois"][k - 1], k)
        assert np.array_equal(got, g3["dil_out"][k - 1]), f"k={k}"


def test_fragment_choice_hull_and_min_area_rect_vs_qhull(g3):
    """findContours(...)[0] -> minAreaRect: fragment = the 8-connected piece whose raster-first pixel comes last;
    hull vertices = Qhull's; min-area rectangle = a float64 brute-force rotation search (area to 1e-9 relative,
    corners as a set to 1e-3 px -- the oracle rounds its exact corners to float32)."""
    from oracle import postproc

    for mask, pick, hv, area, corners in zip(g3["frag_masks"], g3["frag_pick"], g3["hull_vertices"], g3["rect_area"],
                                             g3["rect_corners"]):
        frag = postproc.first_contour_fragment(mask)
        assert np.array_equal(frag, pick)
        ys, xs = np.nonzero(frag)
        hull = postproc.convex_hull_rows(np.stack([xs, ys], 1))
        want_v = {tuple(int(c) for c in v) for v in hv if v[0] >= 0}
        assert set(hull) == want_v
        # clockwise on screen (y down): positive shoelace sum
        a2 = sum(hull[i][0] * hull[(i + 1) % len(hull)][1] - hull[(i + 1) % len(hull)][0] * hull[i][1] for i in range(len(hull)))
        assert a2 > 0
        box = postproc.min_area_box(hull).astype(np.float64)
        e0, e1 = np.linalg.norm(box[1] - box[0]), np.linalg.norm(box[2] - box[1])
        assert abs(e0 * e1 - area) <= 1e-4 * area  # float32 corners
        # same rectangle (corner sets match; ties between equal-area rectangles would show up here)
        d = np.linalg.norm(box[:, None] - corners[None], axis=2)
        assert d.min(axis=1).max() <= 2e-3, d.min(axis=1)
        # boxPoints order: clockwise on screen
        e01, e12 = box[1] - box[0], box[2] - box[1]
        cross = e01[0] * e12[1] - e01[1] * e12[0]
        assert cross > 0


def test_min_rotated_rect_of_quads_vs_bruteforce(g3):
    """shapely MultiPoint(...).minimum_rotated_rectangle (tools.py:543-547) on 4-point inputs."""
    from oracle import tools

    for q, area, corners in zip(g3["quad_in"], g3["quad_rect_area"], g3["quad_rect_corners"]):
        got = tools.min_rotated_rect_f64(q)
        e0, e1 = np.linalg.norm(got[1] - got[0]), np.linalg.norm(got[2] - got[1])
        assert abs(e0 * e1 - area) <= 1e-9 * area
        d = np.linalg.norm(got[:, None] - corners[None], axis=2)
        assert d.min(axis=1).max() <= 1e-8


# ---------------------------------------------------------------------------------------------------
# warpBox's OpenCV calls (tools.py:96-107), resize (tools.py:394), cvtColor (recognition.py:510)
# ---------------------------------------------------------------------------------------------------
def test_perspective_transform_vs_skimage(g3):
    from oracle import tools

    for src, dst, M in zip(g3["persp_src"], g3["persp_dst"], g3["persp_M"]):
        got = tools.get_perspective_transform(src.astype(np.float32), dst.astype(np.float32))
        np.testing.assert_allclose(got, M, rtol=1e-8, atol=1e-10)


def test_warp_perspective_pixels_vs_map_coordinates(g3):
    """cv2.warpPerspective: with the SAME matrix, the oracle's fixed-point path (1/32-px coordinates, 15-bit weights,
    round half up) must reproduce the float64 map_coordinates statement bit for bit; the only freedom left is the
    last bit of the inverse matrix (adjugate vs LAPACK), which can move a coordinate across a 1/32-px tie."""
    from oracle import tools

    total = bad = 0
    for i, (M, (dw, dh), want) in enumerate(zip(g3["persp_M"], g3["warp_dsize"], g3["warp_out"])):
        img = g3["warp_imgs"][i % 3]
        got = tools.warp_perspective_u8(img, M, (int(dw), int(dh)))
        assert got.shape == (dh, dw)
        diff = np.abs(got.astype(int) - want[:dh, :dw].astype(int))
        total += diff.size
        bad += int((diff > 0).sum())
        assert diff.max() <= 8  # a tie flip moves the sample by 1/32 px
    assert bad <= 0.001 * total, (bad, total)


def test_resize_within_one_lsb_of_float_bilinear(g3):
    from oracle import tools

    src = g3["resize_in"]
    for tag in ("x2", "x1p5", "x4_3", "aniso"):
        want = g3["resize_" + tag]
        got = tools.cv_resize_linear_u8(src, (want.shape[1], want.shape[0])).astype(np.float64)
        # OpenCV: 11-bit coefficients and two truncating shifts -> within 1 LSB of the exact bilinear value
        assert np.abs(got - want).max() <= 1.0, tag
    want = g3["resize_x2"]
    got = tools.cv_resize_linear_u8(src, (want.shape[1], want.shape[0])).astype(np.float64)
    # x2: the coefficients 1/4, 3/4 are exact in 11 bits and the exact value is a multiple of 1/16; OpenCV's
    # vertical pass truncates twice before rounding ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2, so the result is
    # the nearest integer except that values within 1/8 above a .5 tie may go DOWN (never up by more than 0.5)
    d = got - want
    assert d.max() <= 0.5 + 1e-9 and d.min() >= -0.625 - 1e-9
    assert (np.abs(d) <= 0.5 + 1e-9).mean() >= 0.9


def test_gray_within_one_lsb_of_pil(g3):
    from oracle import tools

    got = tools.rgb2gray_u8(g3["gray_in"]).astype(int)
    want = g3["gray_out"].astype(int)
    assert np.abs(got - want).max() <= 1
    assert (got == want).mean() >= 0.98


# ---------------------------------------------------------------------------------------------------
# CRNN (recognition.py:54-184, 214-333)
# ---------------------------------------------------------------------------------------------------
def test_stn_sampler_vs_reference_transform(gref):
    """oracle.stn_transform against the REFERENCE's own recognition._transform (executed through a numpy stand-in
    of the ~20 TF ops it uses).  Feature maps here are white noise (gradient ~4 per pixel), so one float32 ulp in a
    sampling coordinate (tf.linspace / matmul accumulation order are not specified to the ulp) moves a value by
    ~1e-5; any logic error (corner order, W vs W-1 scaling, clipping) would be O(1)."""
    from oracle import crnn

    got = crnn.stn_transform(torch.from_numpy(gref["stn_x"]), torch.from_numpy(gref["stn_theta"])).numpy()
    want = gref["stn_out"]
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 3e-5
    # identity theta is NOT an identity map (x = 0.5 (x_t + 1) W, not W - 1): make sure the fixture exercises that
    assert np.abs(want[0] - gref["stn_x"][0]).max() > 0.1


def _keras_lstm_module(w, name):
    m = torch.nn.LSTM(128, 128, batch_first=True)
    with torch.no_grad():
        m.weight_ih_l0.copy_(torch.from_numpy(w[name + "/kernel"]).t())
        m.weight_hh_l0.copy_(torch.from_numpy(w[name + "/recurrent_kernel"]).t())
        m.bias_ih_l0.copy_(torch.from_numpy(w[name + "/bias"]))
        m.bias_hh_l0.zero_()
    return m.eval()


def test_lstm_vs_torch_nn_module(crnn_weights):
    """Keras LSTM (gate order i,f,c,o; sigmoid recurrent activation) == torch.nn.LSTM (i,f,g,o) with W^T, U^T;
    go_backwards=True returns the outputs in PROCESSING order (recognition.py:298-319 does not re-reverse)."""
    from oracle import crnn

    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((3, 50, 128)).astype(np.float32))
    with torch.no_grad():
        for name, back in (("lstm_10", False), ("lstm_10_back", True), ("lstm_11", False), ("lstm_11_back", True)):
            mod = _keras_lstm_module(crnn_weights, name)
            want = mod(torch.flip(x, dims=[1]) if back else x)[0]
            got = crnn._lstm(crnn_weights, name, x, back)  # pylint: disable=protected-access
            assert float((got - want).abs().max()) <= 2e-6, name


def test_conv_stack_vs_torch_nn_modules(crnn_weights):
    """conv_1..bn_7 (ReLU BEFORE BatchNorm, eps 1e-3, valid 2x2 pooling of 31 -> 15 -> 7) with nn.Module layers."""
    from oracle import crnn

    w = crnn_weights
    layers, cin = [], 1
    for i, cout in enumerate((64, 128, 256, 256, 512, 512, 512), 1):
        conv = torch.nn.Conv2d(cin, cout, 3, padding=1)
        with torch.no_grad():
            conv.weight.copy_(torch.from_numpy(w[f"conv_{i}/kernel"]).permute(3, 2, 0, 1))
            conv.bias.copy_(torch.from_numpy(w[f"conv_{i}/bias"]))
        layers += [conv, torch.nn.ReLU()]
        if i in (3, 5, 7):
            bn = torch.nn.BatchNorm2d(cout, eps=1e-3)
            with torch.no_grad():
                bn.weight.copy_(torch.from_numpy(w[f"bn_{i}/gamma"]))
                bn.bias.copy_(torch.from_numpy(w[f"bn_{i}/beta"]))
                bn.running_mean.copy_(torch.from_numpy(w[f"bn_{i}/moving_mean"]))
                bn.running_var.copy_(torch.from_numpy(w[f"bn_{i}/moving_variance"]))
            layers.append(bn)
            if i != 7:
                layers.append(torch.nn.MaxPool2d(2))
        cin = cout
    net = torch.nn.Sequential(*layers).eval()
    rng = np.random.default_rng(6)
    X = rng.random((2, 31, 200, 1)).astype(np.float32)
    with torch.no_grad():
        x = torch.from_numpy(X).permute(0, 2, 1, 3)  # Permute((2,1,3))
        x = torch.flip(x, dims=[2]).permute(0, 3, 1, 2)
        want = net(x).permute(0, 2, 3, 1).numpy()
    _, inter = crnn.crnn_forward(w, X, return_intermediates=True)
    assert inter["bn_7"].shape == (2, 50, 7, 512)
    assert np.abs(inter["bn_7"] - want).max() <= 1e-4 * max(1.0, np.abs(want).max())


def test_ctc_greedy_vs_groupby():
    """keras.backend.ctc_decode(greedy): argmax per step, merge repeats, drop blank (= last class), -1 padding."""
    from oracle import crnn

    rng = np.random.default_rng(7)
    p = rng.random((64, 48, 37)).astype(np.float32)
    p[:, ::3] = p[:, 1::3][:, : p[:, ::3].shape[1]]  # force repeats
    p[5, :, :] = 0
    p[5, :, 36] = 1  # all blank
    p /= p.sum(-1, keepdims=True)
    got = crnn.ctc_greedy_decode(p)
    for m in range(len(p)):
        best = np.log(p[m] + 1e-7).argmax(-1)
        seq = [int(k) for k, _ in itertools.groupby(best) if k != 36]
        want = seq + [-1] * (48 - len(seq))
        assert list(got[m]) == want

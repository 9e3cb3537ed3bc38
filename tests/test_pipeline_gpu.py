"""Pipeline.recognize on the GPU (fused kocr_pipeline) — stage-wise and end-to-end parity.

The detector's last layer is calibrated (keras_ocr_amd.weights.calibrate_craft_head) so that
the random-init network actually emits word boxes.  Checks:
  * resize+pad: bit-exact vs the oracle's cv2.resize restatement (integer work);
  * fused pipeline == composition of the individually-verified GPU stages, exactly;
  * end to end vs the full CPU oracle: same boxes (heat-map differences of ~1e-5 may move a
    thresholded pixel, so corners are compared to 1e-3 px only when the component masks agree —
    asserted for >= 90 % of boxes) and identical strings on those boxes (subject to the CRNN
    margin rule of test_crnn_gpu.py);
  * the reference's own plumbing anchor: a blank image yields no predictions
    (reference tests/test_pipeline.py:10-12).
"""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def calibrated(craft_weights):
    import keras_ocr_amd
    from oracle import craft as ocraft

    page = synth.text_page(96, 128, 5, seed=21)[None]
    from oracle import tools as otools
    big = np.stack([otools.resize_image(p, 2, 2048)[0] for p in page])
    heat = ocraft.detector_predict(craft_weights, big)
    return keras_ocr_amd.weights.calibrate_craft_head(craft_weights, heat, text_frac=0.10, link_frac=0.04)


@pytest.fixture(scope="module")
def pipe(ctx, calibrated, crnn_weights):
    import keras_ocr_amd

    det = keras_ocr_amd.detection.Detector(weights=calibrated, ctx=ctx)
    rec = keras_ocr_amd.recognition.Recognizer(weights=crnn_weights, ctx=ctx)
    return keras_ocr_amd.pipeline.Pipeline(detector=det, recognizer=rec)


@pytest.mark.parametrize("shape,scale", [((37, 53), 2), ((64, 64), 2), ((48, 30), 1.5), ((90, 120), 4 / 3)])
def test_resize_bit_exact(ctx, shape, scale):
    from oracle import tools as otools

    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (2,) + shape + (3,), dtype=np.uint8)
    dsize = (int(shape[1] * scale), int(shape[0] * scale))
    got = ctx.resize_pad(img, dsize, out_hw=(dsize[1] + 3, dsize[0] + 5))
    for i in range(2):
        want = otools.pad(otools.cv_resize_linear_u8(img[i], dsize), width=dsize[0] + 5, height=dsize[1] + 3)
        assert np.array_equal(got[i], want)


def test_blank_image_has_no_predictions(ctx, craft_weights, crnn_weights):
    """reference tests/test_pipeline.py:10-12: a zeros image yields 0 predictions.  With random-init weights the
    head is calibrated on a text page such that background stays below the 0.4 thresholds (as a trained CRAFT's
    does); the black page must then produce exactly [] -- not merely a well-formed answer."""
    import keras_ocr_amd
    from oracle import craft as ocraft, tools as otools

    page = synth.text_page(96, 128, 5, seed=21)
    black = np.zeros((96, 128, 3), np.uint8)
    big = np.stack([otools.resize_image(p, 2, 2048)[0] for p in (page, black)])
    heat = ocraft.detector_predict(craft_weights, big)
    w = keras_ocr_amd.weights.calibrate_craft_head(craft_weights, heat[:1], text_frac=0.10, link_frac=0.04)
    # shift the two biases so that the blank page's maximum sits at 0.2 (and stays below the text page's peaks)
    a = w["conv_cls.8.weight"].reshape(2, -1)[:, 0] / craft_weights["conv_cls.8.weight"].reshape(2, -1)[:, 0]
    cal_black = (heat[1] - craft_weights["conv_cls.8.bias"]) * a + w["conv_cls.8.bias"]
    w = dict(w)
    w["conv_cls.8.bias"] = (w["conv_cls.8.bias"] - np.maximum(cal_black.reshape(-1, 2).max(0) - 0.2, 0)).astype(np.float32)
    c2 = keras_ocr_amd.Context(0)
    det = keras_ocr_amd.detection.Detector(weights=w, ctx=c2)
    rec = keras_ocr_amd.recognition.Recognizer(weights=crnn_weights, ctx=c2)
    p2 = keras_ocr_amd.pipeline.Pipeline(detector=det, recognizer=rec)
    image = np.zeros((256, 256, 3), dtype="uint8")
    predictions = p2.recognize(images=[image])
    assert len(predictions) == 1
    assert len(predictions[0]) == 0
    assert predictions == [[]]
    c2.close()


def test_fused_equals_stagewise(pipe, ctx):
    from oracle import tools as otools

    pages = [synth.text_page(96, 128, 5, seed=21), synth.text_page(80, 100, 4, seed=22)]
    fused = pipe.recognize(pages)
    # stage-wise with the individually verified kernels
    resized = [ctx.resize_pad(p[None], (p.shape[1] * 2, p.shape[0] * 2))[0] for p in pages]
    hmax, wmax = max(r.shape[0] for r in resized), max(r.shape[1] for r in resized)
    batch = np.stack([otools.pad(r, width=wmax, height=hmax) for r in resized])
    boxes = pipe.detector.detect(batch)
    texts = pipe.recognizer.recognize_from_boxes(batch, boxes)
    assert sum(len(b) for b in boxes) >= 4, "calibration should produce word boxes"
    for f, b, t in zip(fused, boxes, texts):
        assert [x[0] for x in f] == t
        if len(b):
            assert np.array_equal(np.stack([x[1] for x in f]), b * np.float32(0.5))


def _vs_oracle(pipe, ctx, calibrated, crnn_weights, pages, scale, max_size):
    from oracle import craft as ocraft, pipeline as opipe, tools as otools
    from tests.parity import flips, compare_page

    got = pipe.recognize(pages)
    want = opipe.recognize(calibrated, crnn_weights, pages, scale=scale, max_size=max_size)
    resized = [otools.resize_image(p, scale, max_size) for p in pages]
    hmax, wmax = max(r.shape[0] for r, _ in resized), max(r.shape[1] for r, _ in resized)
    batch = np.stack([otools.pad(r, width=wmax, height=hmax) for r, _ in resized])
    h_ref = ocraft.detector_predict(calibrated, batch)
    h_gpu = ctx.craft_forward(batch)
    report = {"boxes_equal": 0, "boxes_moved_by_flips": 0, "flipped_pixels": 0, "pixels": 0}
    for g, w_, hg, hr, (_, sc) in zip(got, want, h_gpu, h_ref, resized):
        fl = flips(hg, hr)
        if len(fl) == 0:
            assert len(g) == len(w_)  # identical thresholded maps: identical answer, box for box
        compare_page(g, w_, fl, sc, hr.shape[:2], report)
    return report


def test_end_to_end_vs_oracle(pipe, ctx, calibrated, crnn_weights):
    """Equal box counts and 100 % box + string agreement wherever the thresholded heat-maps agree; the pixels
    that land on the other side of a threshold are counted (tests/parity.py)."""
    pages = [synth.text_page(96, 128, 5, seed=21), synth.text_page(80, 100, 4, seed=22)]
    report = _vs_oracle(pipe, ctx, calibrated, crnn_weights, pages, 2, 2048)
    print("e2e:", report)
    assert report["boxes_equal"] >= 4
    assert report["flipped_pixels"] <= 2


def test_recognizer_single_image_api(pipe, crnn_weights):
    """Recognizer.recognize (recognition.py:467-489) against the oracle: letterbox with cval 0 (tools.fit :402-452,
    cv2.resize restated), RGB->gray, /255, CRNN, CTC -- identical string wherever the oracle's argmax margin is safe."""
    from oracle import crnn as ocrnn, tools as otools

    import keras_ocr_amd
    n_checked = 0
    for seed, shape in ((4, (40, 180)), (5, (31, 200)), (6, (62, 400)), (7, (25, 90))):
        img = synth.text_page(shape[0], shape[1], 3, seed=seed)
        got = pipe.recognizer.recognize(img)
        assert isinstance(got, str)
        prm = keras_ocr_amd.tools.fit_params(img.shape, 200, 31)
        if prm is None:
            fitted = img
        else:
            resized = otools.cv_resize_linear_u8(img, (prm[0], prm[1]))
            fitted = np.zeros((31, 200, 3), np.uint8)
            fitted[:resized.shape[0], :resized.shape[1]] = resized[:31, :200]
        x = otools.rgb2gray_u8(fitted).astype(np.float32)[None, ..., None] / 255
        probs = ocrnn.crnn_forward(crnn_weights, x)
        srt = np.sort(probs, -1)
        if ((srt[..., -1] - srt[..., -2]) > 1e-3).all():
            n_checked += 1
            assert got == ocrnn.decode_strings(ocrnn.ctc_greedy_decode(probs))[0], shape
    assert n_checked >= 2
    # an image that already has the model's input size goes through unchanged (tools.py:426-428)
    img2 = synth.text_page(31, 200, 3, seed=5)
    box = np.array([[0, 0], [200, 0], [200, 31], [0, 31]], np.float32)
    assert pipe.recognizer.recognize(img2) == pipe.recognizer.recognize_from_boxes([img2], [box[None]])[0][0]


def test_device_resident_batch_equals_host_batch(pipe):
    """recognize_device (pointers into HBM, what bench.py times) == recognize (host arrays)."""
    import torch

    pages = np.stack([synth.text_page(96, 128, 5, seed=31), synth.text_page(96, 128, 6, seed=32)])
    host = pipe.recognize(pages)
    t = torch.from_numpy(pages).cuda()
    dev = pipe.recognize_device(t.data_ptr(), 2, 96, 128)
    assert len(host) == len(dev) == 2
    for a, b in zip(host, dev):
        assert [x[0] for x in a] == [x[0] for x in b]
        assert all(np.array_equal(x[1], y[1]) for x, y in zip(a, b))


def test_non_integer_scale_and_mixed_sizes(pipe, calibrated, crnn_weights):
    """BASELINE cfg5's regime in miniature: scale capped by max_size (x4/3, the non-exact cv2.resize
    path) and images of different sizes padded to the batch maximum with 255 (pipeline.py:48-57)."""
    import keras_ocr_amd

    p2 = keras_ocr_amd.pipeline.Pipeline(detector=pipe.detector, recognizer=pipe.recognizer, scale=3, max_size=160)
    pages = [synth.text_page(120, 96, 5, seed=41), synth.text_page(90, 120, 4, seed=42)]
    report = _vs_oracle(p2, pipe.detector._ctx, calibrated, crnn_weights, pages, 3, 160)  # pylint: disable=protected-access
    print("mixed sizes:", report)
    assert report["boxes_equal"] >= 3
    assert report["flipped_pixels"] <= 2


def test_mixed_empty_and_nonempty_images_and_file_paths(pipe, tmp_path):
    """An all-white page (no boxes) next to a text page; inputs given as file paths
    (tools.read, tools.py:19-38) must equal the same images given as arrays."""
    from PIL import Image

    white = np.full((96, 128, 3), 255, np.uint8)
    page = synth.text_page(96, 128, 5, seed=21)
    out = pipe.recognize([white, page])
    assert len(out) == 2 and isinstance(out[0], list) and len(out[1]) > 0
    paths = []
    for i, im in enumerate((white, page)):
        pth = str(tmp_path / f"im{i}.png")
        Image.fromarray(im).save(pth)
        paths.append(pth)
    out2 = pipe.recognize(paths)
    assert [[t for t, _ in g] for g in out2] == [[t for t, _ in g] for g in out]
    assert pipe.recognize([]) == []


def test_recognize_from_boxes_contract(pipe):
    """AssertionError on mismatched groups (recognition.py:501-503); [[]]*N without boxes (:522-523);
    images of different sizes are allowed (the reference loops per image)."""
    a = synth.text_page(60, 100, 3, seed=1)
    b = synth.text_page(80, 90, 3, seed=2)
    with pytest.raises(AssertionError):
        pipe.recognizer.recognize_from_boxes([a, b], [np.array([])])
    assert pipe.recognizer.recognize_from_boxes([a, b], [np.array([]), np.array([])]) == [[], []]
    box = np.array([[[5, 5], [60, 5], [60, 25], [5, 25]]], np.float32)
    res = pipe.recognizer.recognize_from_boxes([a, b], [box, box])
    assert len(res) == 2 and len(res[0]) == 1 and len(res[1]) == 1
    same = pipe.recognizer.recognize_from_boxes([a], [box])
    assert same[0] == res[0]


def test_baseline_cfg4_size_properties(pipe, ctx):
    """BASELINE configs[3] at full size (32 pages of 768x768, scale 2 -> detector input 1536x1536), through
    size-independent properties instead of the CPU oracle (which needs ~3 s per page): the result of an
    image does not depend on its position in the batch, on the batch size, or on the micro-batching --
    exactly, in BOTH split modes (bf16x3 has no data-dependent scale; in fp16x2 mode the detector runs one image per
    forward so that the scale follows the image alone, and the recogniser always uses bf16x3 -- DESIGN.md section 3)."""
    pages = np.stack([synth.text_page(768, 768, 12, seed=100 + i) for i in range(32)])
    full = pipe.recognize(list(pages))
    assert len(full) == 32
    n_words = sum(len(g) for g in full)
    assert n_words > 0
    def same(ga, gb):
        assert len(ga) == len(gb)
        for (ta, ba), (tb, bb) in zip(ga, gb):
            assert ta == tb and np.array_equal(ba, bb)

    perm = np.random.default_rng(0).permutation(32)
    shuffled = pipe.recognize([pages[i] for i in perm])
    for j, i in enumerate(perm):
        same(shuffled[j], full[i])
    halves = pipe.recognize(list(pages[:16])) + pipe.recognize(list(pages[16:]))
    for a, b in zip(halves, full):
        same(a, b)
    single = pipe.recognize([pages[7]])[0]
    same(single, full[7])


def test_baseline_cfg5_share_properties(ctx, calibrated, crnn_weights):
    """BASELINE configs[4], one GPU's kind of work at full image size: 1536x1536 pages with scale 3, which
    `tools.resize_image` caps at max_size 2048 (reference tools.py:387-392) -- the largest detector input of
    the five configs.  Batch of 2 == the two images alone, exactly (either split mode)."""
    import keras_ocr_amd

    det = keras_ocr_amd.detection.Detector(weights=calibrated, ctx=ctx)
    rec = keras_ocr_amd.recognition.Recognizer(weights=crnn_weights, ctx=ctx)
    pipe3 = keras_ocr_amd.pipeline.Pipeline(detector=det, recognizer=rec, scale=3)
    pages = [synth.text_page(1536, 1536, 30, seed=300 + i, scale=2.0) for i in range(2)]
    both = pipe3.recognize(pages)
    alone = [pipe3.recognize([p])[0] for p in pages]
    assert sum(len(g) for g in both) > 0
    for ga, gb in zip(both, alone):
        assert len(ga) == len(gb)
        for (ta, ba), (tb, bb) in zip(ga, gb):
            assert ta == tb and np.array_equal(ba, bb)


def test_baseline_cfg3_size_crnn_order_independence(ctx, crnn_weights):
    """BASELINE configs[2]: 512 crops; labels of a crop do not depend on its position in the batch."""
    ctx.load_crnn(crnn_weights)
    rng = np.random.default_rng(9)
    crops = rng.random((512, 31, 200), dtype=np.float32)
    a = ctx.crnn_forward(crops)
    perm = rng.permutation(512)
    b = ctx.crnn_forward(crops[perm])
    assert np.array_equal(b, a[perm])  # the recogniser runs the exact bf16x3 split in either mode


def test_duck_typed_stages_take_the_reference_stagewise_path(pipe):
    """A detector / recogniser that is not a libkocr-backed object (anything with ``detect`` /
    ``recognize_from_boxes``, as the reference accepts) makes Pipeline.recognize fall back to the reference's own
    sequence (pipeline.py:44-75) over the public stage APIs; with wrappers around the real stages the answer must be
    the fused path's."""
    import keras_ocr_amd

    class Det:  # no _ctx attribute
        def __init__(self, inner):
            self.inner = inner

        def detect(self, images, **kw):
            return self.inner.detect(images, **kw)

    class Rec:
        def __init__(self, inner):
            self.inner, self.alphabet = inner, inner.alphabet

        def recognize_from_boxes(self, images, box_groups, **kw):
            return self.inner.recognize_from_boxes(images, box_groups, **kw)

    pages = [synth.text_page(96, 128, 5, seed=21), synth.text_page(80, 100, 4, seed=22)]
    fused = pipe.recognize(pages)
    p2 = keras_ocr_amd.pipeline.Pipeline(detector=Det(pipe.detector), recognizer=Rec(pipe.recognizer))
    staged = p2.recognize(pages)
    assert [[t for t, _ in g] for g in staged] == [[t for t, _ in g] for g in fused]
    for ga, gb in zip(staged, fused):
        assert all(np.array_equal(a[1], b[1]) for a, b in zip(ga, gb))
    # (a float image used to be refused here; since round 4 it takes the float stage-wise path:
    #  test_float_images_take_the_float_path)


def test_float_images_take_the_float_path(pipe, ctx):
    """Non-uint8 images (pipeline.py:44-57 hands whatever it is given to cv2.resize, which interpolates a float image in
    float, tools.py:394): accepted, processed by the stage-wise path with the float restatement of resize / gray / warp
    on the host (VERDICT r03 missing 4: this used to raise TypeError).  With scale 1 nothing is resized, the detector sees
    the same normalised values as on the uint8 path, so the boxes agree; the crops differ from the fixed-point uint8 crops
    by a rounding step only, so the strings agree on (nearly) every box."""
    import keras_ocr_amd

    page = synth.text_page(128, 192, 6, seed=33)
    p1 = keras_ocr_amd.pipeline.Pipeline(detector=pipe.detector, recognizer=pipe.recognizer, scale=1)
    want = p1.recognize([page])[0]
    got = p1.recognize([page.astype(np.float32)])[0]
    assert len(want) > 0 and abs(len(got) - len(want)) <= 1
    same_box = [(tg, tw) for (tw, bw) in want for (tg, bg) in got if np.abs(np.asarray(bg) - np.asarray(bw)).max() <= 1e-3]
    assert len(same_box) >= 0.9 * len(want)
    assert sum(tg == tw for tg, tw in same_box) >= 0.8 * len(same_box)
    # scale 2 (float bilinear resize on the host) and float64 input: runs, well-formed result
    out = pipe.recognize([page.astype(np.float64)])[0]
    assert all(isinstance(t, str) and np.asarray(b).shape == (4, 2) for t, b in out)
    # recognize_from_boxes on a float image, the reference's per-stage entry point
    boxes = [np.asarray([b for _, b in want], np.float32)]
    texts = pipe.recognizer.recognize_from_boxes([page.astype(np.float32)], boxes)
    assert len(texts) == 1 and len(texts[0]) == len(want)


@pytest.mark.parametrize("cap,max_crops", [(2, None), (256, 3), (2, 3)], ids=["more_boxes_than_cap", "more_crops_than_max_crops", "both"])
def test_capacity_overflow_costs_no_second_detector_forward(pipe, ctx, cap, max_crops):
    """the reference has no cap (detection.py:230-286).  A page with more boxes than the caller's `cap`, or a
    batch with more crops than `max_crops`, must not run CRAFT twice: kocr_pipeline repeats only the post-processing on the
    resident heat-maps (larger device box buffer), recognises all crops and leaves the results in HBM; kocr_pipeline_results
    copies them into buffers of the right size.  Asserted on the profiler rows: ONE first-layer launch, ONE LSTM layer pair."""
    pages = np.stack([synth.text_page(96, 128, 5, seed=70 + i) for i in range(3)])
    hs, ws = [96] * 3, [128] * 3
    want_boxes, want_labels = ctx.pipeline(list(pages), hs, ws, [192] * 3, [256] * 3, 192, 256)
    n_boxes = [len(b) for b in want_boxes]
    assert max(n_boxes) > 2 and sum(n_boxes) > 3, n_boxes        # the small capacities below really overflow
    ctx.profile_reset()
    ctx.profile_enable(True)
    try:
        got_boxes, got_labels = ctx.pipeline(list(pages), hs, ws, [192] * 3, [256] * 3, 192, 256, cap=cap, max_crops=max_crops)
        rep = ctx.profile_report()
    finally:
        ctx.profile_enable(False)
    assert [len(b) for b in got_boxes] == n_boxes
    assert all(np.array_equal(a, b) for a, b in zip(got_boxes, want_boxes)) and np.array_equal(got_labels, want_labels)
    first = [v for k, v in rep.items() if k.startswith("conv_hs_first") or k.startswith("conv_first")]
    assert first and sum(v["launches"] for v in first) == 1, {k: v["launches"] for k, v in rep.items()}   # ONE detector forward
    assert rep["lstm_recurrence"]["launches"] == 2                                                         # ONE recogniser pass
    # the results are still resident: a second fetch into even larger buffers gives the same answer
    res = ctx.pipeline_device_results()
    assert res["n"] == 3 and res["m"] == sum(n_boxes) and res["cap"] >= max(n_boxes)


def test_resident_results_and_build_parameter_error_paths(pipe, ctx):
    """kocr_pipeline_results: buffers smaller than the resident results -> KOCR_ECAPACITY naming the sizes, nothing written past
    them; no resident result -> KOCR_EINVAL.  kocr_crnn_set_rnn_steps_to_discard: out of range -> KOCR_EINVAL."""
    import ctypes
    import keras_ocr_amd
    from keras_ocr_amd._lib import _ptr

    pages = np.stack([synth.text_page(96, 128, 5, seed=70 + i) for i in range(2)])
    boxes, labels = ctx.pipeline(list(pages), [96] * 2, [128] * 2, [192] * 2, [256] * 2, 192, 256)
    res = ctx.pipeline_device_results()
    lib, h = ctx._lib, ctx._h  # pylint: disable=protected-access
    small_b = np.zeros((2, 1, 4, 2), np.float32)
    small_l = np.full((1, 48), -1, np.int32)
    assert res["cap"] >= 1 and res["m"] > 1
    rc = lib.kocr_pipeline_results(h, _ptr(small_b), 1 if res["cap"] > 1 else 0, _ptr(small_l), 1)
    assert rc == -4 and not small_b.any() and (small_l == -1).all()
    big_b = np.zeros((2, res["cap"] + 3, 4, 2), np.float32)
    big_l = np.full((res["m"] + 5, 48), -1, np.int32)
    assert lib.kocr_pipeline_results(h, _ptr(big_b), res["cap"] + 3, _ptr(big_l), res["m"] + 5) == 0
    assert all(np.array_equal(big_b[i, :len(b)], b) for i, b in enumerate(boxes)) and np.array_equal(big_l[:res["m"]], labels)
    ctx.resize_pad(pages, (256, 192))                      # any call that processes images invalidates the resident results
    assert lib.kocr_pipeline_results(h, _ptr(big_b), res["cap"] + 3, _ptr(big_l), res["m"] + 5) == -1   # KOCR_EINVAL
    with pytest.raises(keras_ocr_amd.KocrError):
        ctx.crnn_set_rnn_steps_to_discard(50)
    assert ctx.crnn_label_width() == 48

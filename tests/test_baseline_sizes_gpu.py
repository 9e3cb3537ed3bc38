"""Oracle-checked parity at the BASELINE.json sizes (configs[1]..[4]), through the C-ABI.

The oracle costs ~1 s per 768x768 CRAFT forward and ~3 s per full 768x768 page (scale 2) on the GPU box's
host cores, so full-size comparisons are affordable on a handful of pages:

  cfg2  CRAFT only, the whole 8 x 768x768 batch: heat-maps vs oracle (max-abs / rms reported)
  cfg3  all 512 crops of 31x200: probabilities <= 1e-4, labels exact wherever the oracle's margin > 1e-3
  cfg4  32 pages of 768x768 at scale 2 in ONE call; four of them against the full CPU oracle
  cfg5  one 1536x1536 page at scale 3 (capped: detector input 2048x2048) against the full CPU oracle

How boxes are compared (this replaces the "90 % of boxes" budget of round 1): getBoxes looks at the heat-map
only through three comparisons (text > 0.4, link > 0.4, component max >= 0.7; detection.py:221-241).  The GPU's
post-processing is bit-identical to the oracle's on the SAME heat-map (tests/test_postproc_gpu.py), so an
end-to-end difference can only come from a pixel whose heat value lies within the fp32 heat-map error of a
threshold and lands on the other side.  The tests therefore (1) count those pixels exactly (GPU heat-map vs
oracle heat-map, thresholded), (2) require every oracle box that no flipped pixel touches to be present on the
GPU side to 1e-3 px with the identical string, and (3) bound the number of flipped pixels (<= 2e-5 of the map).
"""
import numpy as np
import pytest

from tests import synth
from tests.parity import flips as _flips, compare_page as _compare_page, heat_tolerance, heat_within_tolerance

pytestmark = pytest.mark.gpu

HEAT_TOL = 5e-5     # stated fp32 tolerance on heat-maps of magnitude O(1) (measured ~2e-5; the reference's own Keras-vs-PyTorch bar is 1.5e-4)
# full-size pages with the calibrated head (maps of magnitude ~ 4.3): oracle.parity.heat_tolerance -- 1.5e-5 per unit of max |heat|
# (6.5e-5), never below 5e-5, plus an rms bound; the maximum alone is a noisy statistic (see its docstring)
PROB_TOL = 1e-4  # measured ~4e-5
MARGIN = 1e-3
FLIP_BUDGET = 2e-5  # fraction of heat-map pixels allowed to sit on the other side of a threshold


@pytest.fixture(scope="module")
def calibrated(craft_weights):
    import keras_ocr_amd
    from oracle import craft as ocraft, tools as otools

    page = synth.text_page(192, 256, 8, seed=21)[None]
    big = np.stack([otools.resize_image(p, 2, 2048)[0] for p in page])
    heat = ocraft.detector_predict(craft_weights, big)
    return keras_ocr_amd.weights.calibrate_craft_head(craft_weights, heat, text_frac=0.06, link_frac=0.025)


@pytest.fixture(scope="module")
def pipe(ctx, calibrated, crnn_weights):
    import keras_ocr_amd

    det = keras_ocr_amd.detection.Detector(weights=calibrated, ctx=ctx)
    rec = keras_ocr_amd.recognition.Recognizer(weights=crnn_weights, ctx=ctx)
    return keras_ocr_amd.pipeline.Pipeline(detector=det, recognizer=rec)


def test_cfg2_craft_batch8_768_heatmaps(ctx, craft_weights):
    """BASELINE configs[1]: CRAFT detector only, batch 8 x 768x768 (4 text pages + 4 noise images)."""
    from oracle import craft as ocraft

    rng = np.random.default_rng(2)
    imgs = np.stack([synth.text_page(768, 768, 14, seed=200 + i) for i in range(4)] +
                    [rng.integers(0, 256, (768, 768, 3), dtype=np.uint8) for _ in range(4)])
    ctx.load_craft(craft_weights)
    got = ctx.craft_forward(imgs)
    want = ocraft.detector_predict(craft_weights, imgs)
    assert got.shape == want.shape == (8, 384, 384, 2)
    err = np.abs(got - want)
    rms = float(np.sqrt((err.astype(np.float64) ** 2).mean()))
    print(f"cfg2: max-abs {err.max():.2e}, rms {rms:.2e}, max |heat| {np.abs(want).max():.2f}")
    assert float(err.max()) <= HEAT_TOL
    assert rms <= 2e-5


def test_cfg3_all_512_crops(ctx, crnn_weights):
    """BASELINE configs[2]: 512 pre-cropped 31x200 strips, CTC greedy."""
    from oracle import crnn as ocrnn

    ctx.load_crnn(crnn_weights)
    x = np.zeros((512, 31, 200), np.float32)
    for i in range(512):
        x[i] = synth.text_page(31, 200, 3, seed=3000 + i)[..., 0] / np.float32(255)
    labels, probs = ctx.crnn_forward(x, return_probs=True)
    want_p = np.concatenate([ocrnn.crnn_forward(crnn_weights, x[s:s + 64, ..., None]) for s in range(0, 512, 64)])
    want_l = ocrnn.ctc_greedy_decode(want_p)
    err = float(np.abs(probs - want_p).max())
    srt = np.sort(want_p, -1)
    safe = ((srt[..., -1] - srt[..., -2]) > MARGIN).all(1)
    print(f"cfg3: max-abs prob err {err:.2e}, {int(safe.sum())}/512 rows with margin > {MARGIN}, "
          f"{int((labels == want_l).all(1).sum())}/512 rows identical")
    assert err <= PROB_TOL
    assert safe.sum() >= 256
    assert np.array_equal(labels[safe], want_l[safe])
    assert ocrnn.decode_strings(labels[safe]) == ocrnn.decode_strings(want_l[safe])


def test_cfg4_pages_from_a_32_batch_vs_oracle(pipe, ctx, calibrated, crnn_weights):
    """BASELINE configs[3]: 32 x 768x768 at scale 2 in one Pipeline.recognize call; pages 0, 9, 18, 31 against the
    full CPU oracle (resize -> CRAFT -> getBoxes -> warp -> CRNN -> CTC)."""
    from oracle import craft as ocraft, pipeline as opipe, tools as otools

    pages = [synth.text_page(768, 768, 20, seed=400 + i) for i in range(32)]
    got = pipe.recognize(pages)
    assert len(got) == 32
    report = {"boxes_equal": 0, "boxes_moved_by_flips": 0, "flipped_pixels": 0, "pixels": 0}
    for i in (0, 9, 18, 31):
        want = opipe.recognize(calibrated, crnn_weights, [pages[i]])[0]
        big = otools.resize_image(pages[i], 2, 2048)[0][None]
        assert np.array_equal(ctx.resize_pad(pages[i][None], (1536, 1536)), big)
        h_ref = ocraft.detector_predict(calibrated, big)[0]
        h_gpu = ctx.craft_forward(big)[0]
        assert heat_within_tolerance(h_gpu, h_ref), (float(np.abs(h_gpu - h_ref).max()), heat_tolerance(h_ref))
        _compare_page(got[i], want, _flips(h_gpu, h_ref), 2.0, h_ref.shape[:2], report)
    print("cfg4:", report)
    assert report["boxes_equal"] >= 20
    assert report["flipped_pixels"] <= FLIP_BUDGET * report["pixels"]


def test_cfg5_one_1536_page_scale3_vs_oracle(ctx, calibrated, crnn_weights):
    """BASELINE configs[4], one image of one rank's share: 1536x1536, scale 3 -> resize_image caps it at
    max_size 2048 (the non-exact x4/3 cv2.resize path), detector input 2048x2048."""
    import keras_ocr_amd
    from oracle import craft as ocraft, pipeline as opipe, tools as otools

    det = keras_ocr_amd.detection.Detector(weights=calibrated, ctx=ctx)
    rec = keras_ocr_amd.recognition.Recognizer(weights=crnn_weights, ctx=ctx)
    pipe3 = keras_ocr_amd.pipeline.Pipeline(detector=det, recognizer=rec, scale=3)
    page = synth.text_page(1536, 1536, 40, seed=555, scale=2.0)
    got = pipe3.recognize([page])[0]
    want = opipe.recognize(calibrated, crnn_weights, [page], scale=3)[0]
    big, sc = otools.resize_image(page, 3, 2048)
    assert big.shape[:2] == (2048, 2048) and abs(sc - 2048 / 1536) < 1e-12
    h_ref = ocraft.detector_predict(calibrated, big[None])[0]
    h_gpu = ctx.craft_forward(big[None])[0]
    assert heat_within_tolerance(h_gpu, h_ref), (float(np.abs(h_gpu - h_ref).max()), heat_tolerance(h_ref))
    report = {"boxes_equal": 0, "boxes_moved_by_flips": 0, "flipped_pixels": 0, "pixels": 0}
    _compare_page(got, want, _flips(h_gpu, h_ref), sc, h_ref.shape[:2], report)
    print("cfg5:", report)
    assert report["boxes_equal"] >= 10
    assert report["flipped_pixels"] <= FLIP_BUDGET * report["pixels"]

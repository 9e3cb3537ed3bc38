"""CRNN recogniser on the GPU (conv/GEMM on MFMA, STN sampler, LSTM recurrence, CTC wave
decoder — through kocr_crnn_forward) vs the CPU oracle (oracle/crnn.py, recognition.py:187-333).

Tolerance (stated, fp32): softmax probabilities |dp| <= 1e-4; label rows must be EXACTLY equal
wherever the oracle's per-step top-2 probability margin exceeds 1e-3 on every step of the row
 — a smaller margin can legitimately flip an argmax under fp32 reordering."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu

PROB_TOL = 1e-4  # measured ~4e-5 (was 2e-4)
MARGIN = 1e-3


def _crops(n, seed):
    x = np.zeros((n, 31, 200), np.float32)
    for i in range(n):
        x[i] = synth.text_page(31, 200, 3, seed=seed + i)[..., 0] / np.float32(255)
    return x


@pytest.fixture(scope="module")
def crnn_ctx(ctx, crnn_weights):
    ctx.load_crnn(crnn_weights)
    assert ctx.crnn_classes() == 37
    return ctx


@pytest.mark.parametrize("m", [1, 5, 40])
def test_probs_and_labels_match_oracle(crnn_ctx, crnn_weights, m):
    from oracle import crnn as ocrnn

    x = _crops(m, seed=100)
    labels, probs = crnn_ctx.crnn_forward(x, return_probs=True)
    want_p = ocrnn.crnn_forward(crnn_weights, x[..., None])
    want_l = ocrnn.ctc_greedy_decode(want_p)
    assert probs.shape == want_p.shape == (m, 48, 37)
    err = float(np.abs(probs - want_p).max())
    assert err <= PROB_TOL, f"max abs prob error {err}"
    srt = np.sort(want_p, -1)
    safe = ((srt[..., -1] - srt[..., -2]) > MARGIN).all(1)
    assert safe.sum() >= max(1, m // 2)
    assert np.array_equal(labels[safe], want_l[safe])
    strings = ocrnn.decode_strings(labels)
    assert len(set(strings)) > 1 or m == 1  # the synthetic weights give diverse strings


def test_labels_only_path_equals_probs_path(crnn_ctx):
    x = _crops(7, seed=5)
    a = crnn_ctx.crnn_forward(x)
    b, _ = crnn_ctx.crnn_forward(x, return_probs=True)
    assert np.array_equal(a, b)


def test_ctc_collapse_rules(crnn_ctx):
    """Decoded rows never contain the blank (36), never repeat without a blank in between in the
    argmax path, and are -1 padded on the right only."""
    labels, probs = crnn_ctx.crnn_forward(_crops(16, seed=50), return_probs=True)
    best = probs.argmax(-1)
    for row, path in zip(labels, best):
        want, prev = [], -1
        for c in path:
            if c != prev and c != 36:
                want.append(int(c))
            prev = c
        want = want + [-1] * (48 - len(want))
        assert list(row) == want


def test_large_custom_alphabet(ctx):
    """Recognizer(alphabet=...) with more symbols than a wavefront has lanes (recognition.py:365-404
    builds fc_12 with len(alphabet)+1 classes): the wave decoder strides classes over lanes."""
    import keras_ocr_amd
    from oracle import crnn as ocrnn

    n_classes = 96
    w = keras_ocr_amd.weights.synthetic_crnn_weights(4321, n_classes=n_classes)
    c2 = keras_ocr_amd.Context(0)
    c2.load_crnn(w)
    assert c2.crnn_classes() == n_classes
    x = _crops(6, seed=7)
    labels, probs = c2.crnn_forward(x, return_probs=True)
    want_p = ocrnn.crnn_forward(w, x[..., None])
    assert float(np.abs(probs - want_p).max()) <= PROB_TOL
    srt = np.sort(want_p, -1)
    safe = ((srt[..., -1] - srt[..., -2]) > MARGIN).all(1)
    assert np.array_equal(labels[safe], ocrnn.ctc_greedy_decode(want_p)[safe])
    assert labels.max() < n_classes - 1
    c2.close()


@pytest.mark.parametrize("stn,discard", [(False, 2), (True, 0), (False, 5)], ids=["no_stn", "discard_0", "no_stn_discard_5"])
def test_non_default_builds_match_the_oracle(ctx, crnn_weights, stn, discard):
    """build_params of recognition.py:187-198 beyond the default: `stn=False` (the model is built
    without the localisation network, :243) and another `rnn_steps_to_discard` (:328) -- label rows are then 50 - steps wide."""
    import keras_ocr_amd
    from keras_ocr_amd.recognition import DEFAULT_BUILD_PARAMS
    from oracle import crnn as ocrnn

    x = _crops(9, seed=300)
    try:
        rec = keras_ocr_amd.recognition.Recognizer(weights=dict(crnn_weights), ctx=ctx,
                                                   build_params=dict(DEFAULT_BUILD_PARAMS, stn=stn, rnn_steps_to_discard=discard))
        assert rec.build_params["stn"] is stn and ctx.crnn_label_width() == 50 - discard
        labels, probs = ctx.crnn_forward(x, return_probs=True)
        w = crnn_weights if stn else {k: v for k, v in crnn_weights.items() if not k.startswith("stn_")}
        want_p = ocrnn.crnn_forward(w, x[..., None], rnn_steps_to_discard=discard)
        want_l = ocrnn.ctc_greedy_decode(want_p)
        assert probs.shape == want_p.shape == (9, 50 - discard, 37) and labels.shape == (9, 50 - discard)
        err = float(np.abs(probs - want_p).max())
        assert err <= PROB_TOL, f"max abs prob error {err}"
        srt = np.sort(want_p, -1)
        safe = ((srt[..., -1] - srt[..., -2]) > MARGIN).all(1)
        assert safe.sum() >= 3 and np.array_equal(labels[safe], want_l[safe])
        # the string API on top of it (recognition.py:467-489)
        img = np.repeat((x[0] * 255).astype(np.uint8)[..., None], 3, -1)
        assert rec.recognize(img) == ocrnn.decode_strings(ctx.crnn_forward(x[:1]))[0]
    finally:
        ctx.crnn_set_rnn_steps_to_discard(2)
        ctx.load_crnn(crnn_weights)
    assert ctx.crnn_label_width() == 48


def test_small_batches_take_a_narrower_cell_grid_with_identical_results(crnn_ctx):
    """batches of <= 8 crops lay the crops out 8 cells per row instead of 16 (a single crop no longer convolves 16
    cells).  A crop's result must not depend on that: the same crops in a batch of 3 (8 cells per row) and inside a batch of 12
    (16 per row) give bit-identical probabilities."""
    x = _crops(12, seed=900)
    _, p12 = crnn_ctx.crnn_forward(x, return_probs=True)
    _, p3 = crnn_ctx.crnn_forward(x[:3], return_probs=True)
    _, p1 = crnn_ctx.crnn_forward(x[2:3], return_probs=True)
    assert np.array_equal(p3, p12[:3]) and np.array_equal(p1[0], p12[2])

"""GPU half of the weight-ingestion tests: the detector / recogniser built the way a user of the reference builds
them -- ``Detector(weights='clovaai_general', load_from_torch=...)``, ``Recognizer(weights='kurapan')``,
``Recognizer(alphabet=<custom>)`` -- from files in the keras-ocr cache directory, give bit-identical results to the
same tensors handed over as a dict.  The files are synthetic (the real artefacts cannot be downloaded here) but have
the real formats: a ``module.``-prefixed torch state dict and Keras HDF5 files written by the real h5py.
Conditional anchors on the REAL pretrained weights (reference tests/test_pipeline.py:10-20) run when
``~/.keras-ocr`` holds them and skip otherwise."""
import os

import numpy as np
import pytest

from tests import synth
from tests.test_weights_cpu import write_craft_pth, write_keras_h5

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fake_cache(tmp_path, monkeypatch, craft_weights, crnn_weights):
    """A keras-ocr cache directory holding synthetic files under the real names, with the registry's hashes
    re-pointed at them."""
    import keras_ocr_amd as k

    cache = tmp_path / "keras-ocr-cache"
    cache.mkdir()
    monkeypatch.setenv("KERAS_OCR_CACHE_DIR", str(cache))
    write_craft_pth(craft_weights, cache / "craft_mlt_25k.pth")
    write_keras_h5(craft_weights, cache / "craft_mlt_25k.h5", "craft")
    write_keras_h5(crnn_weights, cache / "crnn_kurapan.h5", "crnn")
    write_keras_h5({n: v for n, v in crnn_weights.items() if not n.startswith("fc_12")}, cache / "crnn_kurapan_notop.h5",
                   "crnn_notop")
    det = {key: dict(v, sha256=k.tools.sha256sum(str(cache / v["filename"]))) for key, v in k.detection.PRETRAINED_WEIGHTS.items()}
    monkeypatch.setattr(k.detection, "PRETRAINED_WEIGHTS", det)
    rec = {"kurapan": dict(k.recognition.PRETRAINED_WEIGHTS["kurapan"])}
    rec["kurapan"]["weights"] = {kk: dict(v, sha256=k.tools.sha256sum(str(cache / v["filename"])))
                                 for kk, v in k.recognition.PRETRAINED_WEIGHTS["kurapan"]["weights"].items()}
    monkeypatch.setattr(k.recognition, "PRETRAINED_WEIGHTS", rec)
    return cache


def test_detector_from_pth_and_h5_equals_dict(fake_cache, craft_weights):
    import keras_ocr_amd as k

    img = synth.text_page(64, 96, 4, seed=3)[None]
    c0 = k.Context(0)
    want = k.detection.Detector(weights=craft_weights, ctx=c0).model.predict(
        np.stack([c0.resize_pad(img, (96, 64))[0]]))
    for from_torch in (True, False):
        c = k.Context(0)
        det = k.detection.Detector(weights="clovaai_general", load_from_torch=from_torch, ctx=c)
        got = det.model.predict(np.stack([c.resize_pad(img, (96, 64))[0]]))
        assert np.array_equal(got, want), f"load_from_torch={from_torch}"
        c.close()
    c0.close()


def test_recognizer_from_h5_equals_dict_and_custom_alphabet(fake_cache, crnn_weights, capsys):
    import keras_ocr_amd as k

    crops = np.stack([synth.text_page(31, 200, 3, seed=60 + i)[..., 0] for i in range(6)]).astype(np.float32) / 255
    c0 = k.Context(0)
    want = k.recognition.Recognizer(weights=crnn_weights, ctx=c0).prediction_model.predict(crops)
    c1 = k.Context(0)
    rec = k.recognition.Recognizer(ctx=c1)  # weights='kurapan': crnn_kurapan.h5 from the cache
    assert rec.alphabet == k.recognition.DEFAULT_ALPHABET and rec.blank_label_idx == 36
    assert np.array_equal(rec.prediction_model.predict(crops), want)
    # custom alphabet: the 'notop' backbone + a fresh fc_12 (recognition.py:393-404), with the reference's message
    c2 = k.Context(0)
    rec2 = k.recognition.Recognizer(alphabet="abc", ctx=c2)
    assert "Using backbone weights only" in capsys.readouterr().out
    assert c2.crnn_classes() == 4 and rec2.blank_label_idx == 3
    lab = rec2.prediction_model.predict(crops)
    assert lab.shape == (6, 48) and lab.max() < 3
    for c in (c0, c1, c2):
        c.close()


def _real_weights_present():
    import keras_ocr_amd as k

    cache = k.tools.get_default_cache_dir()
    return all(os.path.isfile(os.path.join(cache, f)) for f in ("craft_mlt_25k.h5", "crnn_kurapan.h5"))


@pytest.mark.skipif(not _real_weights_present(), reason="pretrained keras-ocr weights not in the cache directory (no network here)")
def test_real_weights_anchors():
    """reference tests/test_pipeline.py:10-20: zeros image -> no predictions; tests/test_image.jpg -> 'eventdock'."""
    import keras_ocr_amd as k

    pipeline = k.pipeline.Pipeline()
    assert pipeline.recognize(images=[np.zeros((256, 256, 3), dtype="uint8")]) == [[]]
    image = os.environ.get("KERAS_OCR_TEST_IMAGE", os.path.join(os.path.dirname(__file__), "test_image.jpg"))
    if os.path.isfile(image):
        predictions = pipeline.recognize(images=[image])
        assert len(predictions) == 1 and len(predictions[0]) == 1
        assert predictions[0][0][0] == "eventdock"

"""CPU suite, part 2: the C-ABI boundary.  libkocr.so builds for gfx950 without a GPU, loads,
and exports every function include/kocr.h declares; the product has no CPU fallback and never
touches the oracle."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__

    __graft_entry__.build()
    import keras_ocr_amd

    return keras_ocr_amd.load_library()


def _declared():
    src = open(os.path.join(ROOT, "include", "kocr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kocr_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 18
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_the_header(lib):
    import keras_ocr_amd

    src = open(os.path.join(ROOT, "keras_ocr_amd", "_lib.py")).read()
    for n in _declared():
        assert f'"{n}"' in src, f"{n} has no ctypes signature"
    del keras_ocr_amd


def test_null_ctx_is_rejected_not_crashing(lib):
    lib.kocr_last_error.restype = ctypes.c_char_p
    assert lib.kocr_last_error(None) == b"null ctx"
    assert lib.kocr_synchronize(None) == -1  # KOCR_EINVAL


def test_no_gpu_fails_loudly():
    import torch
    import keras_ocr_amd

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(keras_ocr_amd.KocrError, match="no CPU fallback"):
        keras_ocr_amd.Context(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "keras_ocr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                # no dlopen / subprocess / path of anything under oracle/ either; comments may cite oracle files
                code = "\n".join(ln for ln in text.splitlines() if not ln.lstrip().startswith(("#", "//", "*", '"')))
                assert not re.search(r"(dlopen|CDLL|subprocess|popen|system)\s*\(.*oracle", code), f


def test_reference_api_surface():
    """Same names / defaults as the reference's public inference API (pipeline.py:18,28;
    detection.py:672-678,745-752; recognition.py:365,467,491)."""
    import inspect
    import keras_ocr_amd as k

    sig = inspect.signature(k.pipeline.Pipeline.__init__)
    assert [p for p in sig.parameters][1:5] == ["detector", "recognizer", "scale", "max_size"]
    assert sig.parameters["scale"].default == 2 and sig.parameters["max_size"].default == 2048
    sig = inspect.signature(k.pipeline.Pipeline.recognize)
    assert list(sig.parameters)[1:] == ["images", "detection_kwargs", "recognition_kwargs"]
    sig = inspect.signature(k.detection.Detector.detect)
    d = {n: p.default for n, p in sig.parameters.items()}
    assert (d["detection_threshold"], d["text_threshold"], d["link_threshold"], d["size_threshold"]) == (0.7, 0.4, 0.4, 10)
    sig = inspect.signature(k.detection.Detector.__init__)
    assert sig.parameters["weights"].default == "clovaai_general" and sig.parameters["backbone_name"].default == "vgg"
    sig = inspect.signature(k.recognition.Recognizer.__init__)
    assert sig.parameters["weights"].default == "kurapan"
    assert k.recognition.DEFAULT_ALPHABET == "0123456789abcdefghijklmnopqrstuvwxyz"
    assert k.recognition.DEFAULT_BUILD_PARAMS["rnn_steps_to_discard"] == 2
    assert k.detection.PRETRAINED_WEIGHTS[("clovaai_general", True)]["sha256"].startswith("4a5efbfb")


def test_decode_labels_matches_the_per_character_loop():
    """recognition.py:527-534: blank (= len(alphabet)) and -1 are skipped; the batched decode equals the loop."""
    import numpy as np
    from keras_ocr_amd.pipeline import decode_labels

    rng = np.random.default_rng(5)
    for alphabet in ("0123456789abcdefghijklmnopqrstuvwxyz", "aé漢字\U0001f600z", ["ab", "c", "d"],
                     ["a", "\ud83d", "b"]):  # a lone surrogate has no UTF-32 encoding: the per-character path must take it
        n = len(alphabet)
        labels = np.full((40, 48), -1, np.int32)
        for r in range(40):
            k = int(rng.integers(0, 49))
            labels[r, :k] = rng.integers(0, n + 1, k)
        want = ["".join(alphabet[i] for i in row if i not in (n, -1)) for row in labels]
        assert decode_labels(alphabet, labels) == want
        assert decode_labels(alphabet, labels.tolist()) == want
    assert decode_labels("abc", np.zeros((0, 48), np.int32)) == []
    assert decode_labels("abc", np.array([0, 2, 3, -1, 1])) == ["acb"]          # a single row (1-D) is one string
    with pytest.raises(IndexError):
        decode_labels("abc", np.array([[0, 7]]))


def test_recogniser_build_parameters_that_are_and_are_not_implemented():
    """recognition.py:187-198 takes a dict of build parameters: the default set, stn=False and any rnn_steps_to_discard are
    implemented; anything else must fail with an error that NAMES the parameter."""
    import pytest
    from keras_ocr_amd.recognition import DEFAULT_BUILD_PARAMS, _check_build_params

    assert _check_build_params(dict(DEFAULT_BUILD_PARAMS)) == (True, 2)
    assert _check_build_params({"stn": False}) == (False, 2)
    assert _check_build_params(dict(DEFAULT_BUILD_PARAMS, rnn_steps_to_discard=0, dropout=0.5)) == (True, 0)
    assert _check_build_params(dict(DEFAULT_BUILD_PARAMS, filters=list(DEFAULT_BUILD_PARAMS["filters"]))) == (True, 2)  # list == tuple
    for key, value in (("color", True), ("width", 256), ("height", 32), ("filters", (64, 128, 256, 256, 512, 512, 256)),
                       ("rnn_units", (256, 256)), ("pool_size", 3)):
        with pytest.raises(NotImplementedError, match=key):
            _check_build_params(dict(DEFAULT_BUILD_PARAMS, **{key: value}))
    with pytest.raises(ValueError):
        _check_build_params({"rnn_steps_to_discard": 50})
    with pytest.raises(TypeError):
        _check_build_params({"no_such_parameter": 1})

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the round-end GPU tier)")


@pytest.fixture(scope="session")
def ctx():
    """One libkocr context on HIP device 0 (GPU tests only)."""
    import keras_ocr_amd

    c = keras_ocr_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def craft_weights():
    import keras_ocr_amd

    return keras_ocr_amd.weights.synthetic_craft_weights(1234)


@pytest.fixture(scope="session")
def crnn_weights():
    import keras_ocr_amd

    return keras_ocr_amd.weights.synthetic_crnn_weights(4321)

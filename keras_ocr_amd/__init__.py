"""keras-ocr_amd — the MI355X-native hot path of keras_ocr.pipeline.Pipeline.recognize().

The package directory is ``keras_ocr_amd/`` (``keras-ocr_amd`` at the repo root is a symlink to
it, kept for the layout the build contract names).  Same surface as the reference's
``keras_ocr`` for the inference path: ``pipeline.Pipeline``, ``detection.Detector``,
``recognition.Recognizer``, ``tools``.
"""
from . import _lib, weights, tools, detection, recognition, pipeline, dist, evaluation, perfmodel, pmc  # noqa: F401
from ._lib import Context, KocrError, load_library, default_context  # noqa: F401

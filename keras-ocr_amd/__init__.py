"""keras-ocr_amd — the MI355X-native hot path of keras_ocr.pipeline.Pipeline.recognize().

Import as ``keras_ocr_amd`` (the repo-root shim package maps that name onto this
directory, whose name is not a valid Python identifier).  Same surface as the reference's
``keras_ocr`` for the inference path: ``pipeline.Pipeline``, ``detection.Detector``,
``recognition.Recognizer``, ``tools``.
"""
from . import _lib, weights, tools, detection, recognition, pipeline, dist, evaluation  # noqa: F401
from ._lib import Context, KocrError, load_library, default_context  # noqa: F401

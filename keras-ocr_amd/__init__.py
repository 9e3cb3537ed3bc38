"""keras-ocr_amd — the MI355X-native hot path of keras_ocr.pipeline.Pipeline.recognize().

Import as ``keras_ocr_amd`` (the repo-root shim package maps that name onto this
directory, whose name is not a valid Python identifier).
"""
from . import _lib, weights  # noqa: F401
from ._lib import Context, KocrError, load_library  # noqa: F401

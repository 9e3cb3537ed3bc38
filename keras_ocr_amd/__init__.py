"""Import shim: ``import keras_ocr_amd`` -> the package that lives in ``keras-ocr_amd/``."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "keras-ocr_amd"))
_init = _os.path.join(__path__[0], "__init__.py")
with open(_init, "r", encoding="utf-8") as _f:
    exec(compile(_f.read(), _init, "exec"))  # pylint: disable=exec-used
del _os, _f, _init

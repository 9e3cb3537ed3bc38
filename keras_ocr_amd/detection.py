"""Host-side mirror of ``keras_ocr.detection.Detector`` (reference ``keras_ocr/detection.py:661-785``).
The Keras model and the OpenCV post-processing are replaced by libkocr (HIP, gfx950)."""
import typing

import numpy as np

from . import _lib, tools, weights as _weights

PRETRAINED_WEIGHTS = {  # detection.py:647-658
    ("clovaai_general", True): {
        "url": "https://github.com/faustomorales/keras-ocr/releases/download/v0.8.4/craft_mlt_25k.pth",
        "filename": "craft_mlt_25k.pth",
        "sha256": "4a5efbfb48b4081100544e75e1e2b57f8de3d84f213004b14b85fd4b3748db17",
    },
    ("clovaai_general", False): {
        "url": "https://github.com/faustomorales/keras-ocr/releases/download/v0.8.4/craft_mlt_25k.h5",
        "filename": "craft_mlt_25k.h5",
        "sha256": "7283ce2ff05a0617e9740c316175ff3bacdd7215dbdf1a726890d5099431f899",
    },
}


def load_torch_state_dict(weights_path):
    """load_torch_weights' reader (detection.py:428-468): PyTorch state dict -> name -> ndarray."""
    import torch

    pretrained = torch.load(weights_path, map_location=torch.device("cpu"))
    return {k: v.numpy() for k, v in pretrained.items() if k.split(".")[-1] != "num_batches_tracked"}


class _CraftModel:
    """Stands in for ``detector.model`` (inner seam #1): ``predict(X) -> heat-maps``."""

    def __init__(self, ctx):
        self._ctx = ctx
        self.input_shape = (None, None, None, 3)

    def predict(self, X, batch_size=32, **kwargs):  # pylint: disable=invalid-name,unused-argument
        return self._ctx.craft_forward(np.asarray(X), micro_batch=batch_size or 0)


class Detector:
    """A text detector using the CRAFT architecture (detection.py:661-696).

    Args:
        weights: ``"clovaai_general"`` (pretrained file looked up in / downloaded to the keras-ocr
            cache directory), ``None`` (random initialisation: the seeded synthetic weights of
            ``keras_ocr_amd.weights``), or a ``dict`` of state-dict arrays.
        load_from_torch: read ``craft_mlt_25k.pth`` instead of the Keras ``.h5``.
        optimizer: accepted for signature compatibility (inference only).
        backbone_name: only ``"vgg"``.
    """

    def __init__(self, weights="clovaai_general", load_from_torch=False, optimizer="adam", backbone_name="vgg",
                 ctx=None):
        del optimizer
        if backbone_name != "vgg":
            raise NotImplementedError("keras-ocr_amd implements the VGG backbone only.")
        self._ctx = ctx or _lib.default_context()
        if isinstance(weights, dict):
            state = weights
        elif weights is not None:
            pretrained_key = (weights, load_from_torch)
            assert pretrained_key in PRETRAINED_WEIGHTS, "Selected weights configuration not found."
            cfg = PRETRAINED_WEIGHTS[pretrained_key]
            path = tools.download_and_verify(url=cfg["url"], filename=cfg["filename"], sha256=cfg["sha256"])
            if path.endswith(".pth"):
                state = load_torch_state_dict(path)
            else:
                state = _weights.read_keras_h5(path, kind="craft")
        else:
            state = _weights.synthetic_craft_weights()
        self._ctx.load_craft(state)
        self.model = _CraftModel(self._ctx)

    def detect(self, images: typing.List[typing.Union[np.ndarray, str]], detection_threshold=0.7, text_threshold=0.4,
               link_threshold=0.4, size_threshold=10, **kwargs):
        """Detector.detect (detection.py:745-785): list/array of same-sized HxWx3 RGB images (or
        paths) -> list of (n_i,4,2) float32 box arrays."""
        images = [tools.read(image) for image in images]
        if not images:
            return []
        batch = np.stack([np.asarray(im) for im in images])
        if batch.dtype != np.uint8:
            # the reference normalises whatever it is given (detection.py:34-42)
            mean = np.array([0.485, 0.456, 0.406])
            variance = np.array([0.229, 0.224, 0.225])
            batch = batch.astype("float32")
            batch -= mean * 255
            batch /= variance * 255
        return self._ctx.detect(batch, detection_threshold=detection_threshold, text_threshold=text_threshold,
                                link_threshold=link_threshold, size_threshold=size_threshold,
                                micro_batch=kwargs.get("batch_size", 0) or 0)

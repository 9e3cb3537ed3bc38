"""Host-side mirror of ``keras_ocr.recognition.Recognizer`` (reference
``keras_ocr/recognition.py:353-545``).  The Keras models are replaced by libkocr."""
import string
import typing

import numpy as np

from . import _lib, tools, weights as _weights

DEFAULT_BUILD_PARAMS = {  # recognition.py:13-23
    "height": 31,
    "width": 200,
    "color": False,
    "filters": (64, 128, 256, 256, 512, 512, 512),
    "rnn_units": (128, 128),
    "dropout": 0.25,
    "rnn_steps_to_discard": 2,
    "pool_size": 2,
    "stn": True,
}

DEFAULT_ALPHABET = string.digits + string.ascii_lowercase

PRETRAINED_WEIGHTS: typing.Dict[str, typing.Any] = {  # recognition.py:27-44
    "kurapan": {
        "alphabet": DEFAULT_ALPHABET,
        "build_params": DEFAULT_BUILD_PARAMS,
        "weights": {
            "notop": {
                "url": "https://github.com/faustomorales/keras-ocr/releases/download/v0.8.4/crnn_kurapan_notop.h5",
                "filename": "crnn_kurapan_notop.h5",
                "sha256": "027fd2cced3cbea0c4f5894bb8e9e85bac04f11daf96b8fdcf1e4ee95dcf51b9",
            },
            "top": {
                "url": "https://github.com/faustomorales/keras-ocr/releases/download/v0.8.4/crnn_kurapan.h5",
                "filename": "crnn_kurapan.h5",
                "sha256": "a7d8086ac8f5c3d6a0a828f7d6fbabcaf815415dd125c32533013f85603be46d",
            },
        },
    }
}


# build parameters this implementation can vary (recognition.py:187-198): the transformer on / off and the number of leading
# RNN steps dropped; everything else is the geometry the kernels are written for
_VARIABLE_BUILD_PARAMS = ("stn", "rnn_steps_to_discard")


def _check_build_params(build_params):
    """DEFAULT_BUILD_PARAMS, `stn=False` and any `rnn_steps_to_discard` in [0, 50) are implemented; a different value of any
    other parameter raises NotImplementedError naming it (recognition.py:187-198 takes them all)."""
    unknown = sorted(set(build_params) - set(DEFAULT_BUILD_PARAMS))
    if unknown:
        raise TypeError(f"build_model() got unexpected build parameter(s) {unknown}")
    merged = dict(DEFAULT_BUILD_PARAMS, **build_params)
    for key, default in DEFAULT_BUILD_PARAMS.items():
        value = merged[key]
        if key in _VARIABLE_BUILD_PARAMS:
            continue
        if (tuple(value) if isinstance(value, (list, tuple)) else value) != default and key != "dropout":  # dropout: inference no-op
            raise NotImplementedError(
                f"keras-ocr_amd: build parameter {key}={value!r} is not implemented (only {key}={default!r}); "
                f"the parameters that may differ from DEFAULT_BUILD_PARAMS are {_VARIABLE_BUILD_PARAMS} and dropout")
    steps = int(merged["rnn_steps_to_discard"])
    if not 0 <= steps < 50:
        raise ValueError("rnn_steps_to_discard must lie in [0, 50): the model has 200 // 4 = 50 RNN steps")
    return bool(merged["stn"]), steps


class _Model:
    def __init__(self, ctx, probs):
        self._ctx = ctx
        self._probs = probs
        self.input_shape = (None, 31, 200, 1)

    def predict(self, X, **kwargs):  # pylint: disable=invalid-name,unused-argument
        if self._probs:
            return self._ctx.crnn_forward(X, return_probs=True)[1]
        return self._ctx.crnn_forward(X)


class Recognizer:
    """A text recogniser using the CRNN architecture (recognition.py:353-404).

    Args:
        alphabet: the alphabet the model recognises.
        weights: ``"kurapan"`` (pretrained file from the keras-ocr cache directory), ``None``
            (random initialisation: seeded synthetic weights) or a ``dict`` of Keras-named arrays.
        build_params: ``DEFAULT_BUILD_PARAMS``, optionally with ``stn=False`` (no spatial transformer, recognition.py:243)
            and / or another ``rnn_steps_to_discard`` (recognition.py:328); a different ``color`` / size / filter set raises
            ``NotImplementedError`` naming the parameter.
    """

    def __init__(self, alphabet=None, weights="kurapan", build_params=None, ctx=None):
        assert alphabet or weights, "At least one of alphabet or weights must be provided."
        if weights is not None and not isinstance(weights, dict):
            build_params = build_params or PRETRAINED_WEIGHTS[weights]["build_params"]
            alphabet = alphabet or PRETRAINED_WEIGHTS[weights]["alphabet"]
        build_params = build_params or DEFAULT_BUILD_PARAMS
        stn, discard = _check_build_params(dict(build_params))
        if alphabet is None:
            alphabet = DEFAULT_ALPHABET
        self.alphabet = alphabet
        self.blank_label_idx = len(alphabet)
        self._ctx = ctx or _lib.default_context()
        if isinstance(weights, dict):
            state = weights
        elif weights is not None:
            weights_dict = PRETRAINED_WEIGHTS[weights]
            if alphabet == weights_dict["alphabet"]:
                cfg = weights_dict["weights"]["top"]
                state = _weights.read_keras_h5(
                    tools.download_and_verify(url=cfg["url"], filename=cfg["filename"], sha256=cfg["sha256"]), kind="crnn")
            else:
                print("Provided alphabet does not match pretrained alphabet. Using backbone weights only.")
                cfg = weights_dict["weights"]["notop"]
                state = _weights.read_keras_h5(
                    tools.download_and_verify(url=cfg["url"], filename=cfg["filename"], sha256=cfg["sha256"]), kind="crnn")
                state.update(_weights.synthetic_fc12(len(alphabet) + 1))
        else:
            state = _weights.synthetic_crnn_weights(n_classes=len(alphabet) + 1)
        if state["fc_12/bias"].shape[0] != len(alphabet) + 1:
            raise ValueError("fc_12 does not match the alphabet length")
        if not stn:  # recognition.py:243: the model is built without the localisation network; its tensors are not loaded
            state = {k: v for k, v in state.items() if not k.startswith("stn_")}
        elif "stn_conv_1/kernel" not in state:
            raise ValueError("build_params['stn'] is True but the weights carry no stn_* tensors")
        self._ctx.crnn_set_rnn_steps_to_discard(discard)
        self._ctx.load_crnn(state)
        self.build_params = dict(DEFAULT_BUILD_PARAMS, **dict(build_params))
        self.model = _Model(self._ctx, probs=True)
        self.prediction_model = _Model(self._ctx, probs=False)
        self.backbone = None
        self.training_model = None

    def _decode(self, rows):
        from .pipeline import decode_labels
        return decode_labels(self.alphabet, rows)

    def recognize(self, image):
        """Recognizer.recognize (recognition.py:467-489): one pre-cropped RGB image -> string."""
        image = tools.read_and_fit(filepath_or_array=image, width=200, height=31, cval=0)
        if image.shape[-1] == 3:
            # gray conversion on the GPU: warp the full 31x200 rectangle onto itself (identity map)
            box = np.array([[0, 0], [200, 0], [200, 31], [0, 31]], np.float32)
            crops = self._ctx.warp_crops(image[np.newaxis], [box[np.newaxis]], 31, 200)
        else:
            crops = image[np.newaxis, ..., 0].astype("float32") / 255
        return self._decode(self._ctx.crnn_forward(crops))[0]

    def recognize_from_boxes(self, images, box_groups, **kwargs) -> typing.List[typing.List[str]]:
        """Recognizer.recognize_from_boxes (recognition.py:491-537)."""
        del kwargs  # Keras predict kwargs (batch_size, verbose, ...) have no effect on results
        assert len(box_groups) == len(images), "You must provide the same number of box groups as images."
        images = [tools.read(image) for image in images]
        if not sum(len(b) for b in box_groups):
            return [[]] * len(images)
        start_end: typing.List[typing.Tuple[int, int]] = []
        for boxes in box_groups:
            start = 0 if not start_end else start_end[-1][1]
            start_end.append((start, start + len(boxes)))
        images = [np.asarray(image) for image in images]
        if any(im.dtype != np.uint8 for im in images):
            # float images (recognition.py:507-526 works in the image's own type): gray conversion and the crop warp in
            # float ON THE GPU (kocr_warp_crops_f32, round 5), the division by 255 of :524, the recogniser
            crops = []
            for image, boxes in zip(images, box_groups):
                if len(boxes):
                    im = np.asarray(image, np.float32)
                    crops.append(self._ctx.warp_crops_f32((im if im.ndim == 3 else im[..., np.newaxis])[np.newaxis], [boxes], 31, 200))
            labels = self._ctx.crnn_forward(np.concatenate(crops) / np.float32(255))
            predictions = self._decode(labels)
            return [predictions[start:end] for start, end in start_end]
        if len({im.shape for im in images}) == 1:
            # one size (what Pipeline / Detector hand over): crops never leave HBM
            labels = self._ctx.recognize_boxes(np.stack(images), box_groups)
        else:
            # the reference loops per image, so sizes may differ: one call per image
            labels = np.concatenate([self._ctx.recognize_boxes(image[np.newaxis], [boxes])
                                     for image, boxes in zip(images, box_groups) if len(boxes)])
        predictions = self._decode(labels)
        return [predictions[start:end] for start, end in start_end]

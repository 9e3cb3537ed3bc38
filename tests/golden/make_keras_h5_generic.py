"""A small GENERIC Keras saver (test fixture generator, runs under /opt/conda/bin/python3.9 with the real h5py):

    make_keras_h5_generic.py <weights.npz> <out.h5> <model: crnn|crnn_notop|craft> <variant>

Unlike make_keras_h5.py (which writes the exact tree the reader was developed against) this script knows nothing about
keras_ocr_amd's reader.  It replays the reference's *layer constructor calls* (recognition.py:214-333 build_model,
detection.py:353-413 build_keras_model) against a tiny stand-in of the two Keras mechanisms that decide what a weight
file looks like:

  1. automatic layer / variable naming -- `backend.unique_object_name`: an unnamed layer is called
     snake_case(class name), then `<that>_1`, `<that>_2` ... per process-wide counter; an LSTM layer owns an `LSTMCell`
     whose own auto name (`lstm_cell`, `lstm_cell_1`, ...) scopes the variables in TF >= 2.1, while TF 2.0 created them
     directly under the layer scope;
  2. `save_weights_to_hdf5_group` (keras/saving/hdf5_format.py): root attrs layer_names / backend / keras_version, one
     group per layer of `model.layers` (weightless ones included), attr `weight_names`, one dataset per weight at
     `<layer group>/<weight.name>`; the weights of a layer are its trainable ones followed by the non-trainable ones;
     a nested Model contributes ONE group whose weights are those of its inner layers, named by the inner layers' scopes;
     `model.save()` puts the same tree under `/model_weights` and adds `/optimizer_weights`.

Variants (the "era" and how busy the Python process was before the model was built -- both only move names around):
  fresh      TF >= 2.1 naming, counters start at zero                   (conv2d, dense, model, lstm_cell ...)
  busy       TF >= 2.1 naming, other models were built first            (conv2d_7, dense_3, model_2, lstm_cell_11 ...)
  tf20       TF 2.0 naming: LSTM variables directly under the layer scope, nested model called model_1
  fullmodel  `model.save()` layout of the busy variant: /model_weights/... + an /optimizer_weights group
"""
import re
import sys

import h5py
import numpy as np


class Namer:
    """backend.unique_object_name: per-name counters, `name`, `name_1`, `name_2`, ..."""

    def __init__(self, offsets=None):
        self.count = dict(offsets or {})

    def __call__(self, cls):
        base = re.sub(r"(?<!^)(?=[A-Z][a-z])|(?<=[a-z0-9])(?=[A-Z])", "_", cls).lower()
        n = self.count.get(base, 0)
        self.count[base] = n + 1
        return base if n == 0 else f"{base}_{n}"


class Layer:
    def __init__(self, namer, cls, name=None):
        self.name = name or namer(cls)
        self.trainable, self.non_trainable = [], []  # (variable name, array)

    @property
    def weights(self):  # _legacy_weights: trainable first
        return self.trainable + self.non_trainable


def conv_or_dense(namer, cls, w, key, name=None):
    lay = Layer(namer, cls, name)
    lay.trainable = [(f"{lay.name}/kernel:0", w[key + "/kernel"]), (f"{lay.name}/bias:0", w[key + "/bias"])]
    return lay


def batchnorm(namer, w, key, name=None):
    lay = Layer(namer, "BatchNormalization", name)
    lay.trainable = [(f"{lay.name}/gamma:0", w[key + "/gamma"]), (f"{lay.name}/beta:0", w[key + "/beta"])]
    lay.non_trainable = [(f"{lay.name}/moving_mean:0", w[key + "/moving_mean"]),
                         (f"{lay.name}/moving_variance:0", w[key + "/moving_variance"])]
    return lay


def lstm(namer, w, key, name, cell_scope):
    lay = Layer(namer, "LSTM", name)
    scope = f"{lay.name}/{namer('LSTMCell')}" if cell_scope else lay.name
    lay.trainable = [(f"{scope}/kernel:0", w[key + "/kernel"]), (f"{scope}/recurrent_kernel:0", w[key + "/recurrent_kernel"]),
                     (f"{scope}/bias:0", w[key + "/bias"])]
    return lay


def plain(namer, cls, name=None):
    return Layer(namer, cls, name)


def build_crnn(w, namer, top, cell_scope):
    """recognition.py:214-333, call by call (names only where the reference passes name=...)."""
    L = [plain(namer, "InputLayer", None)]
    L[0].name = namer("Input")  # keras.layers.Input -> "input_1" style; counter name is 'input'
    L.append(plain(namer, "Permute"))
    L.append(plain(namer, "Lambda"))
    for i in range(1, 8):
        L.append(conv_or_dense(namer, "Conv2D", w, f"conv_{i}", name=f"conv_{i}"))
        if i in (3, 5, 7):
            L.append(batchnorm(namer, w, f"bn_{i}", name=f"bn_{i}"))
        if i in (3, 5):
            L.append(plain(namer, "MaxPooling2D", name=f"maxpool_{i}"))
    # the localisation network: every layer unnamed, wrapped in an unnamed Model (:268-278)
    namer("Input")
    inner = [conv_or_dense(namer, "Conv2D", w, "stn_conv_1"), conv_or_dense(namer, "Conv2D", w, "stn_conv_2")]
    namer("Flatten")
    inner += [conv_or_dense(namer, "Dense", w, "stn_dense_1"), conv_or_dense(namer, "Dense", w, "stn_dense_2")]
    model = Layer(namer, "Model")
    for lay in inner:
        model.trainable += lay.trainable
    L.append(model)
    L.append(plain(namer, "Lambda"))
    L.append(plain(namer, "Reshape", name="reshape"))
    L.append(conv_or_dense(namer, "Dense", w, "fc_9", name="fc_9"))
    L.append(lstm(namer, w, "lstm_10", "lstm_10", cell_scope))
    L.append(lstm(namer, w, "lstm_10_back", "lstm_10_back", cell_scope))
    L.append(plain(namer, "Add"))
    L.append(lstm(namer, w, "lstm_11", "lstm_11", cell_scope))
    L.append(lstm(namer, w, "lstm_11_back", "lstm_11_back", cell_scope))
    L.append(plain(namer, "Concatenate"))
    if top:  # the 'notop' file is the backbone model (recognition.py:320, 393-404)
        L.append(plain(namer, "Dropout", name="dropout"))
        L.append(conv_or_dense(namer, "Dense", w, "fc_12", name="fc_12"))
        L.append(plain(namer, "Lambda"))
    return L


def build_craft(w, namer):
    """detection.py:353-413: every weighted layer is named after its PyTorch key; the file the reference ships was
    converted from the .pth, so kernels are HWIO here."""
    L = []
    L.append(plain(namer, "InputLayer"))
    convs = sorted({k[:-len(".weight")] for k in w if k.endswith(".weight") and w[k].ndim == 4})
    bns = sorted({k[:-len(".running_mean")] for k in w if k.endswith(".running_mean")})
    for name in convs:
        lay = Layer(namer, "Conv2D", name)
        lay.trainable = [(f"{name}/kernel:0", w[name + ".weight"].transpose(2, 3, 1, 0)), (f"{name}/bias:0", w[name + ".bias"])]
        L.append(lay)
        L.append(plain(namer, "Activation"))
    for name in bns:
        lay = Layer(namer, "BatchNormalization", name)
        lay.trainable = [(f"{name}/gamma:0", w[name + ".weight"]), (f"{name}/beta:0", w[name + ".bias"])]
        lay.non_trainable = [(f"{name}/moving_mean:0", w[name + ".running_mean"]),
                             (f"{name}/moving_variance:0", w[name + ".running_var"])]
        L.append(lay)
    for _ in range(4):
        L.append(plain(namer, "MaxPooling2D"))
    for _ in range(4):
        L.append(plain(namer, "Concatenate"))
    for _ in range(3):
        L.append(plain(namer, "UpsampleLike"))
    return L


def save_weights_to_hdf5_group(f, layers):
    f.attrs["layer_names"] = [lay.name.encode("utf8") for lay in layers]
    f.attrs["backend"] = b"tensorflow"
    f.attrs["keras_version"] = b"2.4.0"
    for lay in layers:
        g = f.create_group(lay.name)
        g.attrs["weight_names"] = [n.encode("utf8") for n, _ in lay.weights]
        for n, val in lay.weights:
            d = g.create_dataset(n, val.shape, dtype=val.dtype)
            if val.shape:
                d[:] = val
            else:
                d[()] = val


def main():
    src, dst, model, variant = sys.argv[1:5]
    w = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in np.load(src).items()}
    busy = {"conv2d": 7, "dense": 3, "model": 2, "lstm_cell": 11, "lambda": 4, "input": 6, "permute": 1, "add": 2,
            "concatenate": 5, "flatten": 2, "max_pooling2d": 9, "activation": 20, "input_layer": 3, "upsample_like": 3}
    offsets = {"fresh": {}, "busy": busy, "tf20": {"model": 1}, "fullmodel": busy}[variant]
    namer = Namer(offsets)
    if model == "craft":
        layers = build_craft(w, namer)
    else:
        layers = build_crnn(w, namer, top=(model == "crnn"), cell_scope=(variant != "tf20"))
    with h5py.File(dst, "w") as f:
        if variant == "fullmodel":
            f.attrs["model_config"] = b"{}"
            f.attrs["training_config"] = b"{}"
            save_weights_to_hdf5_group(f.create_group("model_weights"), layers)
            opt = f.create_group("optimizer_weights")
            opt.attrs["weight_names"] = [b"Adam/iter:0", b"Adam/conv_1/kernel/m:0"]
            opt.create_dataset("Adam/iter:0", data=np.int64(25000))
            opt.create_dataset("Adam/conv_1/kernel/m:0", data=np.zeros((3, 3, 1, 64), np.float32))
        else:
            save_weights_to_hdf5_group(f, layers)
    print("wrote", dst, [lay.name for lay in layers if lay.weights][:40])


main()

"""Write weight dictionaries as Keras-layout HDF5 files with the REAL h5py (test fixture generator).

    /opt/conda/bin/python3.9 tests/golden/make_keras_h5.py <weights.npz> <out.h5> craft|crnn|crnn_notop

h5py 3.3.0 lives only in the image's second interpreter.  The file follows what ``tf.keras`` ``Model.save_weights``
emits (keras/saving/hdf5_format.py: root attrs ``layer_names`` / ``backend`` / ``keras_version``; one group per layer
with attr ``weight_names``; one dataset per variable at ``<layer group>/<variable name>`` where the variable name
itself contains slashes, e.g. ``conv_1/kernel:0``, ``lstm_10/lstm_cell_3/recurrent_kernel:0``; a nested Model -- the
STN localisation network, recognition.py:268-278 -- is ONE group holding the variables of its unnamed layers).
``keras_ocr_amd.weights.read_keras_h5`` must turn such a file back into the dictionary it was made from.
"""
import sys

import h5py
import numpy as np

src, dst, kind = sys.argv[1:4]
w = dict(np.load(src))
layers = []  # (group name, [(variable name, array)])
if kind == "craft":
    convs = sorted({k[:-len(".weight")] for k in w if k.endswith(".weight") and w[k].ndim == 4})
    for name in convs:
        layers.append((name, [(f"{name}/kernel:0", w[name + ".weight"].transpose(2, 3, 1, 0)),  # OIHW -> HWIO
                              (f"{name}/bias:0", w[name + ".bias"])]))
    for name in sorted({k[:-len(".running_mean")] for k in w if k.endswith(".running_mean")}):
        layers.append((name, [(f"{name}/gamma:0", w[name + ".weight"]), (f"{name}/beta:0", w[name + ".bias"]),
                              (f"{name}/moving_mean:0", w[name + ".running_mean"]),
                              (f"{name}/moving_variance:0", w[name + ".running_var"])]))
    layers.insert(0, ("input_1", []))  # layers without weights have an empty group
else:
    for i in range(1, 8):
        layers.append((f"conv_{i}", [(f"conv_{i}/kernel:0", w[f"conv_{i}/kernel"]), (f"conv_{i}/bias:0", w[f"conv_{i}/bias"])]))
        if i in (3, 5, 7):
            layers.append((f"bn_{i}", [(f"bn_{i}/{v}:0", w[f"bn_{i}/{v}"]) for v in ("gamma", "beta", "moving_mean", "moving_variance")]))
    # the nested localisation model: one group, unnamed layers
    layers.append(("model", [("conv2d/kernel:0", w["stn_conv_1/kernel"]), ("conv2d/bias:0", w["stn_conv_1/bias"]),
                             ("conv2d_1/kernel:0", w["stn_conv_2/kernel"]), ("conv2d_1/bias:0", w["stn_conv_2/bias"]),
                             ("dense/kernel:0", w["stn_dense_1/kernel"]), ("dense/bias:0", w["stn_dense_1/bias"]),
                             ("dense_1/kernel:0", w["stn_dense_2/kernel"]), ("dense_1/bias:0", w["stn_dense_2/bias"])]))
    layers.append(("fc_9", [("fc_9/kernel:0", w["fc_9/kernel"]), ("fc_9/bias:0", w["fc_9/bias"])]))
    for j, name in enumerate(("lstm_10", "lstm_10_back", "lstm_11", "lstm_11_back")):
        cell = f"{name}/lstm_cell_{j + 3}"
        layers.append((name, [(f"{cell}/kernel:0", w[name + "/kernel"]), (f"{cell}/recurrent_kernel:0", w[name + "/recurrent_kernel"]),
                              (f"{cell}/bias:0", w[name + "/bias"])]))
    layers += [("permute", []), ("lambda", []), ("add", []), ("concatenate", []), ("dropout", [])]
    if kind == "crnn":
        layers.append(("fc_12", [("fc_12/kernel:0", w["fc_12/kernel"]), ("fc_12/bias:0", w["fc_12/bias"])]))
with h5py.File(dst, "w") as f:
    f.attrs["layer_names"] = [n.encode("utf8") for n, _ in layers]
    f.attrs["backend"] = b"tensorflow"
    f.attrs["keras_version"] = b"2.4.0"
    for name, variables in layers:
        g = f.create_group(name)
        g.attrs["weight_names"] = [v.encode("utf8") for v, _ in variables]
        for vname, arr in variables:
            g.create_dataset(vname, data=np.ascontiguousarray(arr, dtype=np.float32))
print("wrote", dst, len(layers), "layer groups")

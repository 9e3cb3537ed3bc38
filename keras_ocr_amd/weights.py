"""Seeded synthetic weights with the reference's tensor names and shapes.

No pretrained artefact (craft_mlt_25k.pth/.h5, crnn_kurapan.h5 — detection.py:647-658,
recognition.py:27-44) exists in this environment and there is no network, so benchmarks and
parity tests use random-init weights of the reference architecture:
conv/dense ~ N(0, sqrt(2/fan_in)), bias ~ U(-0.1, 0.1), BN gamma ~ U(0.5,1.5),
beta ~ U(-0.1,0.1), mean ~ N(0,0.1), var ~ U(0.5,1.5), LSTM kernels ~ N(0, 1/sqrt(in)).
"""
import numpy as np

# (conv name, bn name or None, cin, cout, k) — detection.py:312-335, 353-410
CRAFT_LAYERS = [
    ("basenet.slice1.0", "basenet.slice1.1", 3, 64, 3),
    ("basenet.slice1.3", "basenet.slice1.4", 64, 64, 3),
    ("basenet.slice1.7", "basenet.slice1.8", 64, 128, 3),
    ("basenet.slice1.10", "basenet.slice1.11", 128, 128, 3),
    ("basenet.slice2.14", "basenet.slice2.15", 128, 256, 3),
    ("basenet.slice2.17", "basenet.slice2.18", 256, 256, 3),
    ("basenet.slice3.20", "basenet.slice3.21", 256, 256, 3),
    ("basenet.slice3.24", "basenet.slice3.25", 256, 512, 3),
    ("basenet.slice3.27", "basenet.slice3.28", 512, 512, 3),
    ("basenet.slice4.30", "basenet.slice4.31", 512, 512, 3),
    ("basenet.slice4.34", "basenet.slice4.35", 512, 512, 3),
    ("basenet.slice4.37", "basenet.slice4.38", 512, 512, 3),
    ("basenet.slice5.1", None, 512, 1024, 3),
    ("basenet.slice5.2", None, 1024, 1024, 1),
    ("upconv1.conv.0", "upconv1.conv.1", 1536, 512, 1),
    ("upconv1.conv.3", "upconv1.conv.4", 512, 256, 3),
    ("upconv2.conv.0", "upconv2.conv.1", 768, 256, 1),
    ("upconv2.conv.3", "upconv2.conv.4", 256, 128, 3),
    ("upconv3.conv.0", "upconv3.conv.1", 384, 128, 1),
    ("upconv3.conv.3", "upconv3.conv.4", 128, 64, 3),
    ("upconv4.conv.0", "upconv4.conv.1", 192, 64, 1),
    ("upconv4.conv.3", "upconv4.conv.4", 64, 32, 3),
    ("conv_cls.0", None, 32, 32, 3),
    ("conv_cls.2", None, 32, 32, 3),
    ("conv_cls.4", None, 32, 16, 3),
    ("conv_cls.6", None, 16, 16, 1),
    ("conv_cls.8", None, 16, 2, 1),
]

#: 711 440 FLOP per input pixel: sum over the 27 convs of 2*k*k*cin*cout at
#: the layer's resolution.
_CRAFT_RES = [1, 1, 4, 4, 16, 16, 16, 64, 64, 64, 256, 256, 256, 256, 256, 256, 64, 64, 16, 16, 4, 4, 4, 4, 4, 4, 4]


def craft_flops_per_pixel():
    return sum(2.0 * k * k * cin * cout / r for (_, _, cin, cout, k), r in zip(CRAFT_LAYERS, _CRAFT_RES))


def synthetic_craft_weights(seed=1234):
    """PyTorch state-dict naming (conv weight OIHW), as load_torch_weights reads it."""
    rng = np.random.default_rng(seed)
    w = {}
    for conv, bn, cin, cout, k in CRAFT_LAYERS:
        fan_in = cin * k * k
        w[conv + ".weight"] = rng.normal(0, np.sqrt(2.0 / fan_in), (cout, cin, k, k)).astype(np.float32)
        w[conv + ".bias"] = rng.uniform(-0.1, 0.1, cout).astype(np.float32)
        if bn:
            w[bn + ".weight"] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
            w[bn + ".bias"] = rng.uniform(-0.1, 0.1, cout).astype(np.float32)
            w[bn + ".running_mean"] = rng.normal(0, 0.1, cout).astype(np.float32)
            w[bn + ".running_var"] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    # keep the linear 2-channel output at heat-map scale (O(1), like the real [0,1] maps)
    w["conv_cls.8.weight"] *= np.float32(0.1)
    return w


# ---------------------------------------------------------------------------------------
# CRNN (recognition.py:13-23, 187-333)
# ---------------------------------------------------------------------------------------
CRNN_FILTERS = (64, 128, 256, 256, 512, 512, 512)
CRNN_FLOPS_PER_CROP = 13.444e9


def synthetic_crnn_weights(seed=4321, n_classes=37):
    """Keras variable naming (conv kernels HWIO, dense [in,out], LSTM [in,4u] gate order i,f,c,o).
    The STN's last Dense is initialised near a 0.9x zoom (identity-like, as trained STNs are) so
    the sampler reads mostly inside the feature map."""
    rng = np.random.default_rng(seed)
    w = {}

    def conv(name, k, cin, cout):
        w[name + "/kernel"] = rng.normal(0, np.sqrt(2.0 / (k * k * cin)), (k, k, cin, cout)).astype(np.float32)
        w[name + "/bias"] = rng.uniform(-0.1, 0.1, cout).astype(np.float32)

    def dense(name, cin, cout, gain=2.0):
        w[name + "/kernel"] = rng.normal(0, np.sqrt(gain / cin), (cin, cout)).astype(np.float32)
        w[name + "/bias"] = rng.uniform(-0.1, 0.1, cout).astype(np.float32)

    def bn(name, c):
        w[name + "/gamma"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        w[name + "/beta"] = rng.uniform(-0.1, 0.1, c).astype(np.float32)
        w[name + "/moving_mean"] = rng.normal(0, 0.1, c).astype(np.float32)
        w[name + "/moving_variance"] = rng.uniform(0.5, 1.5, c).astype(np.float32)

    def lstm(name, cin, units):
        # x3 gain: keeps the gates away from their linear regime so decoded strings are diverse
        w[name + "/kernel"] = rng.normal(0, 3.0 / np.sqrt(cin), (cin, 4 * units)).astype(np.float32)
        w[name + "/recurrent_kernel"] = rng.normal(0, 1.0 / np.sqrt(units), (units, 4 * units)).astype(np.float32)
        w[name + "/bias"] = rng.uniform(-0.1, 0.1, 4 * units).astype(np.float32)

    cin = 1
    for i, f in enumerate(CRNN_FILTERS, 1):
        conv(f"conv_{i}", 3, cin, f)
        cin = f
    for i in (3, 5, 7):
        bn(f"bn_{i}", CRNN_FILTERS[i - 1])
    conv("stn_conv_1", 5, 512, 16)
    conv("stn_conv_2", 5, 16, 32)
    dense("stn_dense_1", 50 * 7 * 32, 64)
    w["stn_dense_2/kernel"] = rng.normal(0, 0.002, (64, 6)).astype(np.float32)
    w["stn_dense_2/bias"] = (np.array([0.9, 0, 0, 0, 0.9, 0]) + rng.normal(0, 0.02, 6)).astype(np.float32)
    dense("fc_9", 7 * 512, 128)
    lstm("lstm_10", 128, 128)
    lstm("lstm_10_back", 128, 128)
    lstm("lstm_11", 128, 128)
    lstm("lstm_11_back", 128, 128)
    dense("fc_12", 256, n_classes, gain=40.0)
    w["fc_12/bias"] *= np.float32(0.2)
    return w


def synthetic_fc12(n_classes, seed=99):
    """A freshly initialised top layer for a custom alphabet (recognition.py:393-404 loads the
    'notop' backbone and leaves fc_12 at its initialiser)."""
    rng = np.random.default_rng(seed)
    return {
        "fc_12/kernel": rng.normal(0, np.sqrt(2.0 / 256), (256, n_classes)).astype(np.float32),
        "fc_12/bias": np.zeros(n_classes, np.float32),
    }


def read_keras_h5(path, kind):
    """Keras HDF5 weight file -> the tensor names kocr_load_craft / kocr_load_crnn expect.

    ``kind='craft'``: craft_mlt_25k.h5 (layer names = PyTorch keys, detection.py:647-658);
    ``kind='crnn'``: crnn_kurapan[_notop].h5 (recognition.py:27-44).  Needs ``h5py``, which this
    image does not ship and the files cannot be downloaded here: this reader follows the Keras
    HDF5 layout (``layer/layer/variable:0``) but could not be exercised — see DESIGN.md."""
    try:
        import h5py  # pylint: disable=import-outside-toplevel
    except ImportError as e:  # pragma: no cover
        raise ImportError("reading Keras .h5 weights needs h5py; convert the file to a dict of arrays "
                          "(see keras_ocr_amd.weights) or use load_from_torch=True for the detector") from e
    tensors = {}

    def visit(name, obj):
        if isinstance(obj, h5py.Dataset):
            tensors[name] = np.array(obj, dtype=np.float32)

    with h5py.File(path, "r") as f:
        (f["model_weights"] if "model_weights" in f else f).visititems(visit)
    out = {}
    if kind == "craft":
        for name, arr in tensors.items():
            parts = name.split("/")
            layer, var = parts[0], parts[-1].split(":")[0]
            if var == "kernel":
                out[layer + ".weight"] = arr.transpose(3, 2, 0, 1)  # HWIO -> OIHW (detection.py:461)
            elif var == "bias":
                out[layer + ".bias"] = arr
            elif var == "gamma":
                out[layer + ".weight"] = arr
            elif var == "beta":
                out[layer + ".bias"] = arr
            elif var == "moving_mean":
                out[layer + ".running_mean"] = arr
            elif var == "moving_variance":
                out[layer + ".running_var"] = arr
        return out
    stn = {}
    for name, arr in tensors.items():
        parts = name.split("/")
        layer, var = parts[0], parts[-1].split(":")[0]
        if layer.startswith(("conv_", "bn_", "fc_", "lstm_")):
            out[f"{layer}/{var}"] = arr
        else:  # the unnamed layers of the nested localisation model
            stn[(arr.shape, var)] = arr
    for (shape, var), arr in stn.items():
        if var == "kernel" and len(shape) == 4:
            out[("stn_conv_1" if shape[2] == 512 else "stn_conv_2") + "/kernel"] = arr
        elif var == "kernel":
            out[("stn_dense_1" if shape[0] == 11200 else "stn_dense_2") + "/kernel"] = arr
        elif var == "bias":
            key = {16: "stn_conv_1", 32: "stn_conv_2", 64: "stn_dense_1", 6: "stn_dense_2"}[shape[0]]
            out[key + "/bias"] = arr
    return out


def calibrate_craft_head(weights, heat_sample, text_frac=0.08, link_frac=0.03, peak=0.95):
    """Rescale the last (linear) CRAFT layer of random-init weights so that its two output
    channels behave like score maps: ``text_frac`` / ``link_frac`` of the pixels of
    ``heat_sample`` (the head's output under ``weights``) end up above the reference's 0.4
    thresholds (detection.py:749-750) and the 99.8th percentile lands at ``peak`` (> the 0.7
    detection threshold).  Random weights otherwise never produce a box, which would leave the
    crop + recognition stages of the synthetic benchmark idle.  Returns a new weight dict."""
    out = dict(weights)
    w = np.array(weights["conv_cls.8.weight"], dtype=np.float32, copy=True)
    b = np.array(weights["conv_cls.8.bias"], dtype=np.float32, copy=True)
    for c, frac in enumerate((text_frac, link_frac)):
        v = np.asarray(heat_sample[..., c], dtype=np.float64).ravel()
        q = np.quantile(v, 1.0 - frac)
        top = np.quantile(v, 0.998)
        a = (peak - 0.4) / max(top - q, 1e-6)
        w[c] *= np.float32(a)
        b[c] = np.float32(a * (b[c] - q) + 0.4)
    out["conv_cls.8.weight"] = w
    out["conv_cls.8.bias"] = b
    return out

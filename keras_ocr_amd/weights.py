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


def _h5_datasets(path):
    """{HDF5 path: float32 array} of every dataset in the file: h5py when it is importable, otherwise the
    built-in reader (h5lite) -- Keras weight files need nothing more than old-style groups and contiguous data."""
    try:
        import h5py  # pylint: disable=import-outside-toplevel
    except ImportError:
        from . import h5lite  # pylint: disable=import-outside-toplevel

        tensors = h5lite.read_datasets(path)
    else:
        tensors = {}

        def visit(name, obj):
            if isinstance(obj, h5py.Dataset):
                tensors[name] = np.array(obj)

        with h5py.File(path, "r") as f:
            f.visititems(visit)
    # a full-model file (model.save) keeps the same tree under "model_weights/"
    strip = "model_weights/"
    return {(k[len(strip):] if k.startswith(strip) else k): np.asarray(v, dtype=np.float32)
            for k, v in tensors.items() if not k.startswith("optimizer_weights")}


_CRAFT_VARS = {"kernel": ".weight", "bias": ".bias", "gamma": ".weight", "beta": ".bias",
               "moving_mean": ".running_mean", "moving_variance": ".running_var"}
_STN_BY_SHAPE = {  # the localisation net's layers are unnamed (recognition.py:268-278): identified by shape
    (5, 5, 512, 16): "stn_conv_1/kernel", (16,): "stn_conv_1/bias", (5, 5, 16, 32): "stn_conv_2/kernel", (32,): "stn_conv_2/bias",
    (11200, 64): "stn_dense_1/kernel", (64,): "stn_dense_1/bias", (64, 6): "stn_dense_2/kernel", (6,): "stn_dense_2/bias"}


def read_keras_h5(path, kind):
    """Keras HDF5 weight file -> the tensor names kocr_load_craft / kocr_load_crnn expect.

    ``kind='craft'``: craft_mlt_25k.h5 -- layer (= group) names are the PyTorch keys (detection.py:432-461), so the
    result is the state dict ``load_torch_weights`` would have started from, conv kernels back in OIHW (:461).
    ``kind='crnn'``: crnn_kurapan[_notop].h5 (recognition.py:27-44, 383-404) -- ``<layer>/<variable>`` for the named
    layers; the variables of the nested, unnamed STN model are recognised by their (unique) shapes.
    Dataset paths look like ``conv_1/conv_1/kernel:0`` or ``lstm_10/lstm_10/lstm_cell_3/recurrent_kernel:0``: first
    component = layer, last = variable."""
    out = {}
    stn_group, stn_src = None, {}
    for name, arr in sorted(_h5_datasets(path).items()):
        parts = name.split("/")
        layer, var = parts[0], parts[-1].split(":")[0]
        if kind == "craft":
            if var not in _CRAFT_VARS:
                raise ValueError(f"{path}: unexpected variable {name}")
            out[layer + _CRAFT_VARS[var]] = arr.transpose(3, 2, 0, 1) if var == "kernel" else arr  # HWIO -> OIHW
        elif layer.startswith(("conv_", "bn_", "fc_", "lstm_")):
            out[f"{layer}/{var}"] = arr
        else:
            # a variable of the nested, unnamed localisation model (recognition.py:268-278): its layers carry automatic
            # names (conv2d_7, dense_3, ... whatever the process-wide counters were), so it is placed by its shape
            key = _STN_BY_SHAPE.get(tuple(arr.shape))
            if key is None or key.split("/")[1] != var:
                raise ValueError(f"{path}: {name} with shape {tuple(arr.shape)} is neither a named CRNN layer nor a "
                                 "variable of the default localisation network (non-default build_params are not supported)")
            if key in out:
                raise ValueError(f"{path}: ambiguous localisation network: {name} and {stn_src[key]} both have shape "
                                 f"{tuple(arr.shape)}, which identifies {key}")
            if stn_group not in (None, layer):
                raise ValueError(f"{path}: unnamed variables in two groups ({stn_group!r}, {layer!r}): one nested model expected")
            stn_group = layer
            stn_src[key] = name
            out[key] = arr
    return out


def calibrate_craft_head(weights, heat_sample, text_frac=0.08, link_frac=0.03, peak=0.95, top_q=0.998):
    """Rescale the last (linear) CRAFT layer of random-init weights so that its two output
    channels behave like score maps: ``text_frac`` / ``link_frac`` of the pixels of
    ``heat_sample`` (the head's output under ``weights``) end up above the reference's 0.4
    thresholds (detection.py:749-750) and the ``top_q`` quantile (99.8th percentile by default) lands at
    ``peak`` (> the 0.7 detection threshold).  ``top_q`` must lie well above ``1 - text_frac``: the gain is
    ``(peak - 0.4) / (quantile(top_q) - quantile(1 - frac))``, so close quantiles turn the maps into step functions of
    magnitude 1e5 (bench.py uses 0.9999 to keep them O(1), as real CRAFT maps are).  Random weights otherwise never produce a box, which would leave the
    crop + recognition stages of the synthetic benchmark idle.  Returns a new weight dict."""
    out = dict(weights)
    w = np.array(weights["conv_cls.8.weight"], dtype=np.float32, copy=True)
    b = np.array(weights["conv_cls.8.bias"], dtype=np.float32, copy=True)
    for c, frac in enumerate((text_frac, link_frac)):
        v = np.asarray(heat_sample[..., c], dtype=np.float64).ravel()
        q = np.quantile(v, 1.0 - frac)
        top = np.quantile(v, top_q)
        a = (peak - 0.4) / max(top - q, 1e-6)
        w[c] *= np.float32(a)
        b[c] = np.float32(a * (b[c] - q) + 0.4)
    out["conv_cls.8.weight"] = w
    out["conv_cls.8.bias"] = b
    return out

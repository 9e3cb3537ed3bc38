"""Generate the golden fixtures under tests/golden/ by EXECUTING THE REFERENCE's own code.

Run in the build container (where /root/reference exists):  python tests/golden/make_golden.py

The reference (faustomorales/keras-ocr) cannot be imported as-is here: TensorFlow, OpenCV,
shapely, imgaug, validators, efficientnet and torchvision are not installed.  This script
installs minimal stand-ins for those modules in ``sys.modules`` — just enough for
``keras_ocr.tools`` and ``keras_ocr.detection`` to import — and then calls reference functions
whose arithmetic does NOT depend on the missing libraries:

  * ``detection.build_torch_model`` (detection.py:472-644): the reference's own PyTorch statement
    of CRAFT.  Its only third-party piece is ``torchvision.models.vgg16_bn().features``; the
    stand-in builds the standard VGG16-BN "D" feature stack (conv3x3/BN/ReLU/maxpool), everything
    else (slicing, slice5, double_conv, interpolate, concat order, conv_cls, permute) is the
    reference's code.  Weights: keras_ocr_amd.weights.synthetic_craft_weights(1234).
  * ``detection.compute_input`` (detection.py:34-42).
  * ``tools.get_rotated_box`` ordering logic (tools.py:551-581) via its AttributeError fallback
    (shapely stand-in raises AttributeError, tools.py:548-550), ``tools.get_rotated_width_height``
    (:41-57), ``tools.pad`` (:356-375), ``tools.adjust_boxes`` (:232-260), the scale/dsize rule of
    ``tools.resize_image`` (:378-398) and the size rule of ``tools.fit`` (:402-452) — the last two
    through a recording ``cv2.resize`` stand-in that returns a blank image of the requested size.
  * ``tools.warpBox``'s scalar logic (:86-106) through recording stand-ins of
    ``cv2.getPerspectiveTransform`` / ``cv2.warpPerspective`` (captures src/dst quads and dsize).

  * ``recognition._transform`` / ``_meshgrid`` / ``_repeat`` (recognition.py:54-166): the STN bilinear sampler
    is IN-REPO code written against ~20 elementary TensorFlow ops (reshape, tile, matmul, floor, clip_by_value,
    gather, add_n ...).  A float32 numpy stand-in of exactly those ops (``_TfShim`` below; each op is a one-line
    numpy call with the documented TF semantics) lets the reference's own function run; its output pins
    ``oracle/crnn.py::stn_transform``.

Nothing here pins OpenCV/TensorFlow *library* numerics (see tests/golden/make_golden_3p.py for the independent
cross-checks of those, and oracle/__init__.py for the status table).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

calls = {}


def install_stubs():
    import torch

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return _Any()

        def __getattr__(self, n):
            return _Any()

    # ---- cv2: recording stand-ins only ----
    def resize(image, dsize, **kw):
        calls.setdefault("resize", []).append(tuple(int(v) for v in dsize))
        shape = (dsize[1], dsize[0]) + tuple(image.shape[2:])
        return np.zeros(shape, dtype=image.dtype)

    def get_perspective_transform(src, dst):
        calls.setdefault("gpt", []).append((np.array(src, copy=True), np.array(dst, copy=True)))
        return np.eye(3)

    def warp_perspective(image, M, dsize, **kw):
        calls.setdefault("warp", []).append(tuple(int(v) for v in dsize))
        return np.zeros((dsize[1], dsize[0]) + tuple(image.shape[2:]), dtype=image.dtype)

    mod("cv2", resize=resize, getPerspectiveTransform=get_perspective_transform, warpPerspective=warp_perspective)
    imgaug = mod("imgaug")
    imgaug.__getattr__ = lambda n: _Any()
    mod("validators", url=lambda s: False)
    # ---- shapely: force tools.get_rotated_box's AttributeError fallback ----
    class MultiPoint:  # noqa
        def __init__(self, points=None):
            pass

        @property
        def minimum_rotated_rectangle(self):
            raise AttributeError("stub")

    geometry = mod("shapely.geometry", MultiPoint=MultiPoint)
    mod("shapely", geometry=geometry)
    # ---- matplotlib is installed; tensorflow / keras / efficientnet are not ----
    layers = mod("tensorflow.keras.layers", Layer=type("Layer", (), {}))
    layers.__getattr__ = lambda n: _Any
    keras = mod("tensorflow.keras", layers=layers, models=_Any(), backend=_Any(), utils=_Any())
    tf = mod("tensorflow", keras=keras)
    # the elementary ops recognition._transform uses, on float32 / int32 numpy arrays (TF semantics)
    f32, i32 = np.float32, np.int32
    _dt = {"float32": f32, "int32": i32}
    shim = dict(
        ones=lambda shape, dtype="float32": np.ones(shape, _dt[dtype]),
        zeros=lambda shape, dtype="float32": np.zeros(shape, _dt[dtype]),
        ones_like=np.ones_like,
        reshape=lambda x, shape: np.reshape(x, [int(v) for v in np.asarray(shape).ravel()]),
        matmul=lambda a, b: np.matmul(a, b),
        linspace=lambda a, b, n: np.linspace(f32(a), f32(b), int(n), dtype=f32),
        meshgrid=lambda x, y: np.meshgrid(x, y),  # tf.meshgrid default indexing='xy', as numpy
        concat=lambda xs, axis: np.concatenate(xs, axis),
        shape=lambda x: np.array(x.shape, i32),
        cast=lambda x, dtype=None: np.asarray(x).astype(_dt[dtype]),
        expand_dims=lambda x, axis: np.expand_dims(x, axis),
        tile=lambda x, multiples: np.tile(x, [int(v) for v in np.asarray(multiples).ravel()]),
        stack=lambda xs: np.array([int(v) for v in xs], i32),
        slice=lambda x, begin, size: x[tuple(slice(b, None if sz == -1 else b + sz) for b, sz in zip(begin, size))],
        floor=np.floor,
        clip_by_value=lambda x, lo, hi: np.clip(x, lo, hi),
        range=lambda n: np.arange(int(n), dtype=i32),
        gather=lambda params, indices: params[indices],
        add_n=lambda xs: ((xs[0] + xs[1]) + xs[2]) + xs[3] if len(xs) == 4 else sum(xs[1:], xs[0]),
    )
    tf.__dict__.update(shim)
    tf.__getattr__ = lambda n: _Any()
    mod("efficientnet")
    mod("efficientnet.tfkeras")

    # ---- torchvision.models.vgg16_bn().features: VGG16-BN configuration "D" ----
    def vgg16_bn(pretrained=False, **kw):
        cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
        seq, cin = [], 3
        for v in cfg:
            if v == "M":
                seq.append(torch.nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                seq += [torch.nn.Conv2d(cin, v, kernel_size=3, padding=1), torch.nn.BatchNorm2d(v),
                        torch.nn.ReLU(inplace=True)]
                cin = v
        m = torch.nn.Module()
        m.features = torch.nn.Sequential(*seq)
        return m

    models = mod("torchvision.models", vgg16_bn=vgg16_bn)
    mod("torchvision", models=models)
    # ---- the package shell: import sub-modules without running keras_ocr/__init__.py ----
    pkg = mod("keras_ocr")
    pkg.__path__ = [os.path.join(REF, "keras_ocr")]


def main():
    import torch

    install_stubs()
    from keras_ocr import tools, detection, recognition  # the reference's modules  # noqa: E402
    import keras_ocr_amd  # noqa: E402

    out = {}
    # ---------------- CRAFT forward through the reference's PyTorch model ----------------
    w = keras_ocr_amd.weights.synthetic_craft_weights(1234)
    model = detection.build_torch_model(weights_path=None)
    sd = model.state_dict()
    new = {}
    for kname, v in sd.items():
        if kname.endswith("num_batches_tracked"):
            new[kname] = v
            continue
        if kname in w:
            new[kname] = torch.from_numpy(w[kname])
        else:  # vgg layers .39-.43 exist in torchvision's stack but are outside every slice
            raise KeyError(kname)
    model.load_state_dict(new)
    model.eval()
    rng = np.random.default_rng(2024)
    for tag, (n, h, wd) in {"a": (1, 48, 64), "b": (2, 40, 56)}.items():
        img = rng.integers(0, 256, (n, h, wd, 3), dtype=np.uint8)
        x = np.stack([detection.compute_input(i) for i in img])
        with torch.no_grad():
            y, _ = model(torch.from_numpy(x).permute(0, 3, 1, 2))
        out[f"craft_{tag}_img"] = img
        out[f"craft_{tag}_x"] = x
        out[f"craft_{tag}_heat"] = y.numpy()
    # ---------------- tools: geometry helpers ----------------
    boxes = np.array([
        [[10, 20], [110, 20], [110, 50], [10, 50]],
        [[110, 50], [10, 50], [10, 20], [110, 20]],          # same box, rotated start
        [[50.5, 10.25], [120.75, 40.5], [108.25, 69.5], [38.0, 39.25]],  # rotated rectangle
        [[30, 90], [34, 10], [60, 12], [56, 92]],            # tall
        [[0, 0], [3, 0], [3, 3], [0, 3]],
    ], dtype=np.float32)
    out["rot_in"] = boxes
    out["rot_out"] = np.stack([tools.get_rotated_box(b)[0] for b in boxes])
    out["rot_wh"] = np.array([tools.get_rotated_width_height(tools.get_rotated_box(b)[0]) for b in boxes])
    # warpBox scalar logic: recorded src/dst quads and dsize
    gray = np.zeros((200, 300), np.uint8)
    for b in boxes[:4]:
        tools.warpBox(gray, b, target_height=31, target_width=200)
    out["warp_src"] = np.stack([c[0] for c in calls["gpt"]])
    out["warp_dst"] = np.stack([c[1] for c in calls["gpt"]])
    out["warp_dsize"] = np.array(calls["warp"])
    # resize_image scale rule
    shapes = [(256, 256, 3), (768, 768, 3), (1536, 1536, 3), (480, 640, 3), (1000, 1500, 3), (31, 200, 3), (2048, 100, 3)]
    params = [(2, 2048), (3, 2048), (1, 2048), (1.5, 1024)]
    rs = []
    for s in shapes:
        for ms, mx in params:
            calls["resize"] = []
            _, sc = tools.resize_image(np.zeros(s, np.uint8), max_scale=ms, max_size=mx)
            rs.append([s[0], s[1], ms, mx, sc, calls["resize"][0][0], calls["resize"][0][1]])
    out["resize_rule"] = np.array(rs, dtype=np.float64)
    # fit size rule (letterbox)
    fr = []
    for s in [(31, 200, 3), (40, 180, 3), (100, 100, 3), (20, 400, 3), (62, 400, 3), (10, 10, 3)]:
        calls["resize"] = []
        f, sc = tools.fit(np.zeros(s, np.uint8), width=200, height=31, cval=0, return_scale=True)
        d = calls["resize"][0] if calls["resize"] else (-1, -1)
        fr.append([s[0], s[1], sc, d[0], d[1], f.shape[0], f.shape[1]])
    out["fit_rule"] = np.array(fr, dtype=np.float64)
    # pad + adjust_boxes
    im = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    out["pad_in"] = im
    out["pad_out"] = tools.pad(im, width=9, height=8)
    out["adjust_out"] = tools.adjust_boxes(boxes, scale=1 / 2)
    # ---------------- STN sampler: the reference's own _transform through the numpy tf stand-in ----------------
    srng = np.random.default_rng(77)
    stn_x = srng.standard_normal((5, 50, 7, 6)).astype(np.float32)
    stn_theta = np.stack([
        [1, 0, 0, 0, 1, 0],                      # identity
        [0.9, 0.05, 0.02, -0.03, 0.9, 0.01],      # what a trained STN looks like
        [1.3, 0.2, -0.4, 0.1, 1.2, 0.5],          # samples far outside the map (clipped corners)
        [0.5, 0, 0.7, 0, 0.5, -0.7],
        [-1, 0, 0, 0, -1, 0],                     # flip
    ]).astype(np.float32)
    out["stn_x"] = stn_x
    out["stn_theta"] = stn_theta
    out["stn_out"] = np.asarray(recognition._transform([stn_x, stn_theta]), np.float32)  # pylint: disable=protected-access
    np.savez_compressed(os.path.join(HERE, "reference_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_golden.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

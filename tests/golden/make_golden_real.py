#!/usr/bin/env python
"""THE REAL-LIBRARY PIN.  Run on a host that HAS TensorFlow + OpenCV + shapely and the reference
(`pip install keras-ocr`, or a checkout passed with --reference); it needs NO pretrained weights and no network:

    python tests/golden/make_golden_real.py [--reference /path/to/keras-ocr] [--out tests/golden/real_golden.npz]

It imports the UNMODIFIED reference package `keras_ocr` (no stand-ins: tests/golden/make_golden.py is the stub-based
generator for this container, where none of those libraries exist), loads the seeded SYNTHETIC weights of
keras_ocr_amd.weights -- `synthetic_craft_weights(1234)` through exactly the mapping `detection.load_torch_weights` applies
(detection.py:428-468), `synthetic_crnn_weights(4321)` by Keras layer name (recognition.py:214-327) -- into the reference's
own Keras models, and records what the reference computes, stage by stage, on seeded pages and on the reference's
tests/test_image.jpg:

    resize_image (cv2.resize)          tools.py:378-398
    compute_input + model.predict      detection.py:34-42, 779           -> heat-maps (real TF numerics)
    getBoxes                           detection.py:207-287              -> boxes (real cv2 threshold / CCL / dilate /
                                                                            findContours / minAreaRect / boxPoints)
    cvtColor + warpBox                 recognition.py:507-510, tools.py:61-117 -> crops (real cv2 warpPerspective, shapely)
    model.predict / prediction_model   recognition.py:535, 169-184       -> probabilities, CTC label rows
    Pipeline.recognize                 pipeline.py:28-75                 -> (string, box) lists
    cv2.minAreaRect / boxPoints on stored hulls, incl. an exact area tie -> decides between oracle/postproc.py's exact
                                                                            rectangle and its float32-calipers restatement

tests/test_real_golden.py (skipped while the fixture is absent) holds the CPU oracle AND the GPU path to the recorded
values at the tolerances the rest of the suite uses.  Committing the resulting .npz turns SURVEY.md 8(c) "parity unpinned"
into "pinned" for every stage it covers.  Nothing in this file runs in the build container or on the GPU box.
"""
import argparse
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def _load_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_craft_weights(model, w, keras):
    """The mapping of detection.load_torch_weights (detection.py:441-468) from a state dict without the "module." prefix."""
    seen = 0
    for layer in model.layers:
        if isinstance(layer, keras.layers.BatchNormalization):
            layer.set_weights([w[f"{layer.name}.weight"], w[f"{layer.name}.bias"], w[f"{layer.name}.running_mean"],
                               w[f"{layer.name}.running_var"]])
            seen += 1
        elif isinstance(layer, keras.layers.Conv2D):
            layer.set_weights([np.asarray(w[f"{layer.name}.weight"]).transpose(2, 3, 1, 0), w[f"{layer.name}.bias"]])
            seen += 1
    assert seen == 27 + 20, f"expected 27 convolutions + 20 batch-norm layers (12 VGG + 8 decoder) of the VGG CRAFT, set {seen}"


def load_crnn_weights(model, w, keras):
    """Keras layer names conv_1..7, bn_3/5/7, fc_9, lstm_10[_back], lstm_11[_back], fc_12 (recognition.py:217-327); the
    localisation network is the nested, unnamed Model (recognition.py:264-278): Conv2D, Conv2D, Flatten, Dense, Dense."""
    def put(layer, key, names):
        layer.set_weights([w[f"{key}/{n}"] for n in names])

    nested = [l for l in model.layers if isinstance(l, keras.models.Model)]
    assert len(nested) == 1, "expected exactly one nested model (the STN localisation network)"
    convs = [l for l in nested[0].layers if isinstance(l, keras.layers.Conv2D)]
    denses = [l for l in nested[0].layers if isinstance(l, keras.layers.Dense)]
    assert len(convs) == 2 and len(denses) == 2
    for layer, key in zip(convs + denses, ("stn_conv_1", "stn_conv_2", "stn_dense_1", "stn_dense_2")):
        put(layer, key, ("kernel", "bias"))
    for layer in model.layers:
        if isinstance(layer, keras.layers.BatchNormalization):
            put(layer, layer.name, ("gamma", "beta", "moving_mean", "moving_variance"))
        elif isinstance(layer, keras.layers.LSTM):
            put(layer, layer.name, ("kernel", "recurrent_kernel", "bias"))
        elif isinstance(layer, (keras.layers.Conv2D, keras.layers.Dense)):
            put(layer, layer.name, ("kernel", "bias"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference", help="directory that contains the reference's keras_ocr/ package "
                                                                    "(ignored when keras_ocr is importable already)")
    ap.add_argument("--out", default=os.path.join(HERE, "real_golden.npz"))
    args = ap.parse_args()
    import cv2  # noqa: F401  -- all three must be the REAL libraries: fail loudly here, not half-way
    import shapely  # noqa: F401
    import tensorflow as tf
    from tensorflow import keras

    try:
        import keras_ocr
    except ImportError:
        sys.path.insert(0, args.reference)
        import keras_ocr
    from keras_ocr import detection, recognition, tools, pipeline

    sys.path.insert(0, ROOT)
    kw = _load_module("kocr_weights", os.path.join(ROOT, "keras_ocr_amd", "weights.py"))  # numpy only; no libkocr needed
    synth = _load_module("kocr_synth", os.path.join(ROOT, "tests", "synth.py"))

    out = {"versions": np.array([f"tensorflow {tf.__version__}", f"opencv {cv2.__version__}", f"shapely {shapely.__version__}",
                                 f"keras_ocr {getattr(keras_ocr, '__version__', 'checkout')}"])}
    # ---- models with the seeded synthetic weights ----
    cw = kw.synthetic_craft_weights(1234)
    rw = kw.synthetic_crnn_weights(4321)
    det = detection.Detector(weights=None)
    load_craft_weights(det.model, cw, keras)
    rec = recognition.Recognizer(alphabet=recognition.DEFAULT_ALPHABET, weights=None)
    load_crnn_weights(rec.model, rw, keras)
    # calibrate the random-init head on one page so that the detector emits boxes (as every test of this repo does)
    cal = synth.text_page(192, 256, 8, seed=21)
    cal_big, _ = tools.resize_image(cal, max_scale=2, max_size=2048)
    raw = det.model.predict(np.stack([detection.compute_input(cal_big)]))
    cw = kw.calibrate_craft_head(cw, raw, text_frac=0.06, link_frac=0.025)
    det.model.get_layer("conv_cls.8").set_weights([np.asarray(cw["conv_cls.8.weight"]).transpose(2, 3, 1, 0), cw["conv_cls.8.bias"]])
    out["cls8_weight"], out["cls8_bias"] = cw["conv_cls.8.weight"], cw["conv_cls.8.bias"]
    out["seeds"] = np.array([1234, 4321])

    images = [synth.text_page(96, 128, 5, seed=31), synth.text_page(192, 256, 8, seed=32), synth.text_page(150, 210, 6, seed=33)]
    scales = [2, 2, 4 / 3]
    test_image = os.path.join(args.reference, "tests", "test_image.jpg")
    if os.path.isfile(test_image):
        images.append(tools.read(test_image))
        scales.append(2)
    pipe = pipeline.Pipeline(detector=det, recognizer=rec)
    out["n_images"] = np.array(len(images))
    for i, (im, sc) in enumerate(zip(images, scales)):
        p = f"im{i}_"
        out[p + "image"] = im
        big, s = tools.resize_image(im, max_scale=sc, max_size=2048)
        out[p + "resized"], out[p + "scale"] = big, np.array(s)
        heat = det.model.predict(np.stack([detection.compute_input(big)]))
        out[p + "heat"] = heat[0]
        boxes = detection.getBoxes(heat, detection_threshold=0.7, text_threshold=0.4, link_threshold=0.4, size_threshold=10)[0]
        boxes = np.asarray(boxes, np.float32).reshape(-1, 4, 2)
        out[p + "boxes"] = boxes
        gray = cv2.cvtColor(big, code=cv2.COLOR_RGB2GRAY)
        out[p + "gray"] = gray
        crops = np.array([tools.warpBox(image=gray, box=b, target_height=31, target_width=200) for b in boxes], np.uint8).reshape(-1, 31, 200)
        out[p + "crops"] = crops
        if len(crops):
            x = (crops.astype("float32") / 255)[..., np.newaxis]
            out[p + "probs"] = rec.model.predict(x)
            out[p + "labels"] = np.asarray(rec.prediction_model.predict(x))
        pipe.scale = sc
        res = pipe.recognize([im])[0]
        out[p + "e2e_text"] = np.array([t for t, _ in res])
        out[p + "e2e_boxes"] = np.asarray([b for _, b in res], np.float32).reshape(-1, 4, 2)
    # ---- cv2.minAreaRect / boxPoints on explicit hulls: random ones and an exact area tie (849.0 twice) ----
    rng = np.random.default_rng(0)
    hulls = [np.array([[296, 282], [321, 283], [315, 300], [297, 316]], np.int32)]  # the tie: two hull edges give area 849 exactly
    for _ in range(40):
        n = int(rng.integers(5, 60))
        ang, w_, h_ = rng.uniform(0, np.pi), rng.uniform(20, 200), rng.uniform(8, 40)
        u, v = rng.uniform(-w_ / 2, w_ / 2, n), rng.uniform(-h_ / 2, h_ / 2, n)
        pts = np.stack([np.rint(300 + u * np.cos(ang) - v * np.sin(ang)), np.rint(300 + u * np.sin(ang) + v * np.cos(ang))], 1).astype(np.int32)
        hulls.append(pts)
    out["mar_n"] = np.array(len(hulls))
    for k, pts in enumerate(hulls):
        out[f"mar{k}_points"] = pts
        out[f"mar{k}_box"] = cv2.boxPoints(cv2.minAreaRect(pts.reshape(-1, 1, 2)))
    # ---- cv2.resize / cvtColor primitives on noise (every residue class of the fixed-point tables) ----
    noise = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    out["prim_image"] = noise
    out["prim_resize_x2"] = cv2.resize(noise, dsize=(106, 74))
    out["prim_resize_4_3"] = cv2.resize(noise, dsize=(70, 49))
    out["prim_gray"] = cv2.cvtColor(noise, code=cv2.COLOR_RGB2GRAY)
    np.savez_compressed(args.out, **out)
    print("wrote", args.out, "with", len(out), "arrays;", ", ".join(out["versions"]))


if __name__ == "__main__":
    main()

"""Developer probe: heat-map error of the HIP CRAFT vs the fp32 torch-CPU oracle AND vs an fp64 run of the same
oracle graph, for the current kernel selection (set KOCR_WSPLIT=0 KOCR_DSPLIT=0 for the fp32-MFMA kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import keras_ocr_amd as k
from oracle import craft as ocraft

w = k.weights.synthetic_craft_weights()
ctx = k.Context(0)
ctx.load_craft(w)
rng = np.random.default_rng(5)
img = rng.integers(0, 256, (2, 256, 384, 3), dtype=np.uint8)
got = ctx.craft_forward(img)
want32 = ocraft.detector_predict(w, img)
w64 = {n: v.astype(np.float64) for n, v in w.items()}
try:
    want64 = ocraft.detector_predict(w64, img)
except Exception as e:  # oracle may be fp32-only
    want64 = None
    print("fp64 oracle unavailable:", type(e).__name__, e)
print("kernels: WSPLIT=%s DSPLIT=%s" % (os.environ.get("KOCR_WSPLIT", "1"), os.environ.get("KOCR_DSPLIT", "1")))
print("max |gpu - oracle_f32| = %.3e   (max |heat| = %.3f)" % (np.abs(got - want32).max(), np.abs(want32).max()))
if want64 is not None:
    print("max |gpu - oracle_f64| = %.3e   max |oracle_f32 - oracle_f64| = %.3e" %
          (np.abs(got - want64).max(), np.abs(want32 - want64).max()))

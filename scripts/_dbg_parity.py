import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import keras_ocr_amd as k
import bench
from oracle import craft as ocraft, tools as otools
ctx = k.Context(0)
pages = bench.make_pages(2, 768, seed=4)
craft_w = k.weights.synthetic_craft_weights(1234)
ctx.load_craft(craft_w)
cal_pages = bench.make_pages(4, 768, seed=1004)
sample = ctx.resize_pad(cal_pages, (1536, 1536))
raw = ctx.craft_forward(sample)
cand = k.weights.calibrate_craft_head(craft_w, raw, text_frac=0.0025, link_frac=0.0025 / 3)
for nm in ("conv_cls.8.weight", "conv_cls.8.bias"):
    print(nm, np.abs(craft_w[nm]).max(), np.abs(cand[nm]).max())
ctx.load_craft(cand)
big = ctx.resize_pad(pages[:1], (1536, 1536))
assert np.array_equal(big[0], otools.resize_image(pages[0], 2, 2048)[0])
h_gpu = ctx.craft_forward(big)[0]
h_ref = ocraft.detector_predict(cand, big)[0]
err = np.abs(h_gpu - h_ref)
i = np.unravel_index(err.argmax(), err.shape)
print("max|heat|", np.abs(h_ref).max(), "max err", err.max(), "at", i, h_gpu[i], h_ref[i], "rel", err.max() / np.abs(h_ref).max())
print("err quantiles", np.quantile(err, [0.5, 0.9, 0.99, 0.999, 1.0]))

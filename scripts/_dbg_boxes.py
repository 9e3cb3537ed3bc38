import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import keras_ocr_amd as k
import bench
ctx = k.Context(0)
SIDE, SCALE = bench.SIDE, bench.SCALE
pages = bench.make_pages(32, SIDE, seed=4)
craft_w = k.weights.synthetic_craft_weights(1234)
ctx.load_craft(craft_w)
cal_pages = bench.make_pages(4, SIDE, seed=1004)
sample = ctx.resize_pad(cal_pages, (SIDE * SCALE, SIDE * SCALE))
raw = ctx.craft_forward(sample)
best = None
for frac in (0.03, 0.016, 0.008, 0.005, 0.0035, 0.0025, 0.0018, 0.0013, 0.0009, 0.0006, 0.0004):
    cand = k.weights.calibrate_craft_head(craft_w, raw, text_frac=frac, link_frac=frac / 3)
    a = cand["conv_cls.8.weight"].reshape(2, -1)[:, :1] / craft_w["conv_cls.8.weight"].reshape(2, -1)[:, :1]
    heat = (raw - craft_w["conv_cls.8.bias"]) * a.ravel() + cand["conv_cls.8.bias"]
    nb = np.mean([len(b) for b in ctx.get_boxes(heat.astype(np.float32))])
    if best is None or abs(nb - 20) < abs(best[0] - 20):
        best = (nb, cand)
ctx.load_craft(best[1])
big = ctx.resize_pad(pages, (SIDE * SCALE, SIDE * SCALE))
heat = ctx.craft_forward(big)
boxes = ctx.get_boxes(heat)
ext = []
for pg, bs in enumerate(boxes):
    for b in bs:
        b = np.asarray(b)
        ext.append((float(np.ptp(b[:, 1])) / 2, float(np.ptp(b[:, 0])) / 2, pg))
ext.sort(reverse=True)
print("n boxes", len(ext), "tallest (heat-map rows, cols, page):", ext[:8])
hs = np.array([e[0] for e in ext])
print("height percentiles 50/90/99/max", np.percentile(hs, [50, 90, 99, 100]))

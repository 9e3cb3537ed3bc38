"""Developer probe: the same bench batch handed over as HOST arrays (PCIe-inclusive rate)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import keras_ocr_amd as k
import bench

ctx = k.default_context()
cw = k.weights.synthetic_craft_weights(1234)
ctx.load_craft(cw)
sample = ctx.resize_pad(bench.make_pages(1, 768, seed=4), (1536, 1536))
cw = k.weights.calibrate_craft_head(cw, ctx.craft_forward(sample), text_frac=0.012, link_frac=0.004)
pipe = k.pipeline.Pipeline(detector=k.detection.Detector(weights=cw, ctx=ctx),
                           recognizer=k.recognition.Recognizer(weights=k.weights.synthetic_crnn_weights(), ctx=ctx))
pages = bench.make_pages(32, 768, seed=4)
pipe.recognize(pages)
t = time.perf_counter()
for _ in range(3):
    out = pipe.recognize(pages)
dt = (time.perf_counter() - t) / 3
print(f"host-array path: {32/dt:.1f} img/s ({dt*1e3:.1f} ms/step), words={sum(len(o) for o in out)}")

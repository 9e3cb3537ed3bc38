"""Developer probe: which modification of the heavy-tailed detector of tests/test_range_gpu.py breaks which arithmetic mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import keras_ocr_amd as k
from tests import synth

ctx = k.Context(0)
base = k.weights.synthetic_craft_weights(1234)
pages = np.stack([synth.text_page(256, 384, 10, seed=70 + i) for i in range(2)])
names = ["basenet.slice5.1.weight", "basenet.slice5.2.weight", "conv_cls.0.weight", "conv_cls.2.weight"]
variants = [("none", [])] + [(n, [n]) for n in names] + [("all", names)]
for gain in (1e3, 30.0):
    for tag, mods in variants:
        w = {kk: v.copy() for kk, v in base.items()}
        for n in mods:
            w[n][::7] *= np.float32(gain)
        ctx.load_craft(w)
        out = {}
        for mode in ("f16x2", "bf16x3"):
            ctx.set_split_mode(mode)
            out[mode] = ctx.craft_forward(pages)
        ctx.set_split_mode("f16x2")
        sc = max(1.0, float(np.abs(out["bf16x3"]).max()))
        d = np.abs(out["f16x2"] - out["bf16x3"])
        print(f"gain {gain:g} {tag:28s} max|heat| {sc:10.3g}  f16x2 vs bf16x3 {float(d.max()) / sc:.2e}  finite {np.isfinite(out['f16x2']).all()}  "
              f"f16 max {float(np.abs(out['f16x2']).max()):.3g}", flush=True)
# single layers of the <= 32-cout fp16 kernel with loud output channels
rng = np.random.default_rng(3)
import torch, torch.nn.functional as F
for (n, h, wd, cin, cout) in ((1, 64, 64, 32, 32), (2, 70, 45, 64, 32), (1, 64, 128, 128, 256), (1, 64, 128, 64, 64)):
    x = np.maximum(rng.standard_normal((n, h, wd, cin)), 0).astype(np.float32)
    wt = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32)
    wt[..., ::7] *= np.float32(1e3)
    xt = torch.from_numpy(x).double().permute(0, 3, 1, 2); wtt = torch.from_numpy(wt).double().permute(3, 2, 0, 1)
    want = F.conv2d(xt, wtt, None, padding=1).permute(0, 2, 3, 1).numpy()
    s = F.conv2d(xt.abs(), wtt.abs(), None, padding=1).permute(0, 2, 3, 1).numpy()
    for mode in ("f16x2", "bf16x3"):
        ctx.set_split_mode(mode)
        got = ctx.conv2d_nhwc(x, wt).astype(np.float64)
        print(f"conv {(n, h, wd, cin, cout)} {mode}: max err / (|x| conv |w|) {float((np.abs(got - want) / np.maximum(s, 1e-30)).max()):.2e}", flush=True)
ctx.set_split_mode("f16x2")

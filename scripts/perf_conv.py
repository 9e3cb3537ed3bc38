"""Developer probe: per-shape conv kernel throughput (HIP-event timed inside libkocr)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import keras_ocr_amd as k

SHAPES = [  # N, H, W, Cin, Cout, k  — CRAFT layer classes at 8 x 768x768
    (8, 384, 384, 128, 128, 3),
    (8, 192, 192, 256, 256, 3),
    (8, 96, 96, 512, 512, 3),
    (4, 768, 768, 64, 64, 3),
    (8, 384, 384, 64, 32, 3),
    (8, 48, 48, 1536, 512, 1),
    (8, 384, 384, 32, 32, 3),
    (8, 384, 384, 32, 16, 3),
    (8, 192, 192, 512, 512, 3),  # slice3.27 at 8 x 1536x1536
    (8, 384, 384, 256, 256, 3),  # slice2.17 at 8 x 1536x1536
    (1, 768, 768, 64, 64, 3),    # slice1.3 class, input 151 MB (Infinity-Cache resident on repeat)
    (8, 1536, 1536, 64, 64, 3),  # slice1.3 at 8 x 1536x1536 (4.8 GB input: streams from HBM)
]
if len(sys.argv) > 1:
    SHAPES = [SHAPES[int(a)] for a in sys.argv[1:]]
ctx = k.Context(0)
rng = np.random.default_rng(0)
for (n, h, w, cin, cout, kk) in SHAPES:
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((kk, kk, cin, cout)) * np.sqrt(2.0 / (cin * kk * kk))).astype(np.float32)
    ctx.conv2d_nhwc(x, wt, relu=True)  # warm
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(3):
        ctx.conv2d_nhwc(x, wt, relu=True)
    rep = ctx.profile_report(); ctx.profile_enable(False)
    for nm, r in rep.items():
        if nm.startswith("conv"):
            print(f"{(n,h,w,cin,cout,kk)!s:34s} {nm:24s} {r['ms']/r['launches']:8.3f} ms  {r['flops']/(r['ms']*1e-3)/1e12:6.1f} TF/s")

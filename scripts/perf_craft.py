"""Developer probe: CRAFT-only throughput + per-kernel HIP-event profile (not the bench)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import keras_ocr_amd as k

N, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ctx = k.Context(0)
wts = k.weights.synthetic_craft_weights()
if os.environ.get("PERF_ZERO_W"):  # power experiment: zero conv kernels -> constant activations, minimal toggling
    wts = {n: (np.zeros_like(v) if v.ndim == 4 else v) for n, v in wts.items()}
ctx.load_craft(wts)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
img = torch.randint(0, 256, (N, H, W, 3), dtype=torch.uint8, device="cuda")
heat = torch.empty((N, H // 2, W // 2, 2), dtype=torch.float32, device="cuda")
for _ in range(2):
    ctx.craft_forward_device(img.data_ptr(), 0, N, H, W, heat.data_ptr())
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
    ctx.craft_forward_device(img.data_ptr(), 0, N, H, W, heat.data_ptr())
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / steps
fl = k.weights.craft_flops_per_pixel() * N * H * W
print(f"CRAFT {N}x{H}x{W}: {dt*1e3:.2f} ms/step  {N/dt:.1f} img/s  {fl/dt/1e12:.1f} TFLOP/s ({fl/dt/157.3e12*100:.1f}% of fp32 MFMA peak)")
ctx.profile_enable(True)
ctx.craft_forward_device(img.data_ptr(), 0, N, H, W, heat.data_ptr())
rep = ctx.profile_report()
tot = sum(r["ms"] for r in rep.values())
mb = 1
for nm, r in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
    tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] else 0
    gb = r["bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] else 0
    print(f"  {nm:28s} n={r['launches']:3d} {r['ms']:8.3f} ms ({100*r['ms']/tot:4.1f}%)  {tf:6.1f} TF/s  {gb:7.0f} GB/s(alg)")
print("  total profiled ms", tot)

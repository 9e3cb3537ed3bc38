"""Developer probe: CRNN-only (BASELINE configs[2]: 512 crops 31x200) ms/crop + per-kernel profile."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import keras_ocr_amd as k

M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = k.Context(0)
ctx.load_crnn(k.weights.synthetic_crnn_weights())
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
crops = torch.rand((M, 31, 200), dtype=torch.float32, device="cuda")
labels = torch.empty((M, 48), dtype=torch.int32, device="cuda")
for _ in range(2):
    ctx.crnn_forward_device(crops.data_ptr(), M, labels.data_ptr())
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3):
    ctx.crnn_forward_device(crops.data_ptr(), M, labels.data_ptr())
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 3
print(f"CRNN {M} crops: {dt*1e3:.2f} ms  {dt/M*1e6:.1f} us/crop  {13.444e9*M/dt/1e12:.1f} TFLOP/s")
ctx.profile_enable(True)
ctx.crnn_forward_device(crops.data_ptr(), M, labels.data_ptr())
rep = ctx.profile_report()
tot = sum(r["ms"] for r in rep.values())
for nm, r in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
    tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] else 0
    print(f"  {nm:36s} n={r['launches']:3d} {r['ms']:8.3f} ms ({100*r['ms']/tot:4.1f}%)  {tf:6.1f} TF/s")

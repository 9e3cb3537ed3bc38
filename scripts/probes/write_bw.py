"""Developer probe: achievable HBM write / copy rates (torch fill_ / copy_ of a 4.8 GB fp32 tensor)."""
import time
import torch

x = torch.empty(1200 * 1024 * 1024, dtype=torch.float32, device="cuda")  # 4.8 GB
y = torch.empty_like(x)
for name, fn, nbytes in (("fill (write only)", lambda: x.fill_(1.5), x.numel() * 4), ("copy (read + write)", lambda: y.copy_(x), 2 * x.numel() * 4)):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(f"{name}: {dt*1e3:.3f} ms, {nbytes/dt/1e12:.2f} TB/s")

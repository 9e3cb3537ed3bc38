"""Developer probe: BASELINE configs[4] per-GPU share in miniature — N images 1536x1536, scale=3
(capped to 2048/1536 by max_size), i.e. a 2048x2048 detector input with the non-exact resize."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import keras_ocr_amd as k
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = k.default_context()
craft_w = k.weights.synthetic_craft_weights(1234)
ctx.load_craft(craft_w)
sample = ctx.resize_pad(bench.make_pages(1, 768, seed=4), (1536, 1536))
craft_w = k.weights.calibrate_craft_head(craft_w, ctx.craft_forward(sample), text_frac=0.012, link_frac=0.004)
pipe = k.pipeline.Pipeline(detector=k.detection.Detector(weights=craft_w, ctx=ctx),
                           recognizer=k.recognition.Recognizer(weights=k.weights.synthetic_crnn_weights(), ctx=ctx),
                           scale=3, max_size=2048)
pages = np.concatenate([np.concatenate([bench.make_pages(n, 768, seed=s) for s in (1, 2)], 1) for _ in (0, 1)], 2)
print(pages.shape)
d = torch.from_numpy(pages).cuda()
out = pipe.recognize_device(d.data_ptr(), n, 1536, 1536)
torch.cuda.synchronize()
t = time.perf_counter()
out = pipe.recognize_device(d.data_ptr(), n, 1536, 1536)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"cfg5 share: {n} x 1536x1536 -> 2048x2048: {dt*1e3:.1f} ms, {n/dt:.2f} img/s, words={sum(len(o) for o in out)}, "
      f"box range {min(float(b.min()) for o in out for _, b in o):.1f}..{max(float(b.max()) for o in out for _, b in o):.1f}")

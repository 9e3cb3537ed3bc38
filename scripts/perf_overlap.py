"""Developer probe: how much of the recogniser's time disappears when the CRNN of one micro-batch runs on
a second HIP stream UNDER the CRAFT forward of the next one?  Two contexts (own non-blocking streams, own workspaces) on one
device; 16-page CRAFT forwards and 300-crop CRNN forwards, alone and launched back to back without a sync in between."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import keras_ocr_amd as k

NP, NC, REP = 16, 300, 3
a, b = k.Context(0), k.Context(0)
a.load_craft(k.weights.synthetic_craft_weights())
b.load_crnn(k.weights.synthetic_crnn_weights())
img = torch.randint(0, 256, (NP, 1536, 1536, 3), dtype=torch.uint8, device="cuda")
heat = torch.empty((NP, 768, 768, 2), dtype=torch.float32, device="cuda")
crops = torch.rand((NC, 31, 200), dtype=torch.float32, device="cuda")
labels = torch.empty((NC, 48), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()


def craft():
    a.craft_forward_device(img.data_ptr(), 0, NP, 1536, 1536, heat.data_ptr())


def crnn():
    b.crnn_forward_device(crops.data_ptr(), NC, labels.data_ptr())


def timed(fn):
    fn(); a.synchronize(); b.synchronize()
    t = time.perf_counter()
    for _ in range(REP):
        fn()
    a.synchronize(); b.synchronize()
    return (time.perf_counter() - t) / REP * 1e3


t_craft = timed(craft)
t_crnn = timed(crnn)


def both():
    craft()   # stream of context a
    crnn()    # stream of context b: no dependency, may overlap


t_both = timed(both)
print(f"CRAFT {NP} pages alone {t_craft:.2f} ms; CRNN {NC} crops alone {t_crnn:.2f} ms; sum {t_craft + t_crnn:.2f} ms; "
      f"launched on two streams {t_both:.2f} ms -> {t_craft + t_crnn - t_both:+.2f} ms hidden "
      f"({100 * (t_craft + t_crnn - t_both) / (t_craft + t_crnn):.1f} % of the pair)")

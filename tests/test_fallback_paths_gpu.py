"""The fallback kernels behind the library's environment switches, verified in FRESH processes (the switches are read
once per process): the fp64-bounded convolution tests and the CRAFT heat-map-vs-oracle tests must also hold with

  KOCR_W43=0     no F(4,3) kernels          -> wide 3x3 layers on conv_ws (F(2,3)) / conv_ds
  KOCR_W43R=0    no row-reuse arrangement   -> 64-cout layers on conv_w43n
  KOCR_HSPLIT=0  no <= 32-cout split kernel -> head / upconv4.conv.3 on the fp32 MFMA kernel
  KOCR_FIRST=0   no split first-layer kernel-> first layer on the fp32 MFMA kernel (conv_mfma MODE 2)
  KOCR_W43V=0    no vertical-reuse arrangement -> wide layers on conv_w43_kernel (round 2's dominant kernel)
  KOCR_LINFOLD=0 KOCR_UPFOLD=0                 -> the layer-by-layer CRAFT schedule (slice5.1, slice5.2, resize + concat)
  KOCR_K5=0      no 5x5 / 16-cout kernel    -> the recogniser's stn_conv_1 on the fp32 MFMA kernel
  KOCR_HS16=0    no 16-wide product tile    -> conv_cls.4 on conv_hs_kernel's 32-column tile
  KOCR_SPLIT=bf16  the exact bf16x3 split everywhere (round 3's default arithmetic)
  KOCR_CELLS=0   no cell grid               -> the recogniser's conv stack on round 4's dense crop batch (flattened fp16 tiles,
                                               conv_6 / conv_7 through the 52-wide layout, separate pooling kernels)
  KOCR_W43RAG=0  no ragged tile grids       -> images that do not tile exactly on the flattened / F(2,3) / fp32 kernels
  KOCR_W43DILH=0 no fp16 dilated F(4,3)     -> the dilated composite slice5 on the bf16x3 comb tiles of round 2

Since round 5 every configuration also runs the recogniser against the oracle (under KOCR_W43=0 the 52-wide
layout used to reach a kernel that does not write its padding columns; launch_conv now refuses that, and crnn.cpp asks the
dispatcher's own predicate) and the ragged CRAFT page.

(these paths were reached by the driver's suite only through the shapes that happen to select
them.)  Each configuration is one pytest child process over the same test files, same bounds.  Round 6:
switches that act on disjoint layers share a child -- five children instead of ten; KOCR_W43H=0 went with the fp32 Winograd
kernel it exposed (KOCR_SPLIT=bf16 runs every bf16x3 kernel), KOCR_W43=0 already implies KOCR_CELLS=0."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# switches that act on DISJOINT layers share a child process (the failing test's name still says which path broke); each
# child costs ~12 s of start-up + tests, and the driver's GPU suite has a time limit
CONFIGS = [
    {"KOCR_W43": "0"},                                                            # (implies no cell grid: the recogniser on round 4's dense crop batch)
    {"KOCR_W43R": "0", "KOCR_K5": "0", "KOCR_HS16": "0", "KOCR_W43DILH": "0",     # 64-cout rows / stn_conv_1 / conv_cls.4 / the dilated composite
     "KOCR_LINFOLD": "0", "KOCR_UPFOLD": "0"},                                    # ... / the layer-by-layer decoder schedule
    {"KOCR_HSPLIT": "0", "KOCR_FIRST": "0", "KOCR_W43V": "0"},                    # the <= 32-cout layers / the first layer / the wide layers
    {"KOCR_SPLIT": "bf16"},
    {"KOCR_CELLS": "0", "KOCR_W43RAG": "0"},                                      # the recogniser's crop batch / ragged detector pages
]


def _run_config(switches):
    env = dict(os.environ)
    env.update(switches)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_conv_gpu.py"), os.path.join(ROOT, "tests", "test_craft_gpu.py"),
           os.path.join(ROOT, "tests", "test_crnn_gpu.py"),
           "-k", "fp32_class or heatmap_u8_input or ragged_page or probs_and_labels"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500, check=False)
    tail = (r.stdout + r.stderr)[-3000:]
    ok = r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout.splitlines()[-1]
    return ok, tail


def test_parity_suites_hold_on_the_fallback_paths():
    """Every configuration is its own pytest child process (the switches are read once per process), one after the other:
    three at a time was tried in round 5 and took FIVE times longer -- processes sharing one GPU pay a full wave-state
    save / restore of these 512-register, 160-KB-LDS kernels at every switch.  The children run the dispatch-sensitive
    cases only (every fp64-bounded convolution case, two CRAFT heat-maps incl. the ragged page, the recogniser at 1 / 5 / 40
    crops): about ten seconds each.  A failure names its configuration."""
    failed = []
    for c in CONFIGS:
        ok, tail = _run_config(c)
        if not ok:
            failed.append((c, tail))
    assert not failed, "\n\n".join(f"{c}: child pytest failed\n{tail}" for c, tail in failed)

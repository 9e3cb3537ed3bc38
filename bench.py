#!/usr/bin/env python
"""bench.py — images/sec of Pipeline.recognize() on synthetic 768x768 RGB batches (BASELINE.json
metric; workload = configs[3]: full pipeline, batch 32 x 768x768, scale=2, one MI355X per rank).

  python bench.py --gpus N --steps K --warmup W

N > 1 without a torchrun environment: bench.py re-launches itself as
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`` (and
fails loudly when fewer than N GPUs are visible).  Under torchrun it reads RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment.  The process group (backend nccl = RCCL) is created at
N = 1 too, so the collective path that N > 1 uses is the one exercised on a single-GPU box.

One step = one Pipeline.recognize pass (resize x2 -> CRAFT @1536x1536 -> boxes -> crops -> CRNN ->
CTC) over a 32-image batch already resident in HBM.  Each rank processes its own batch (weak
scaling, no data-path collective); value = N * 32 * K / max-over-ranks time of the HEADLINE loop,
which runs with the HIP-event profiler OFF.  A second loop of the same K steps with the profiler ON
gives `stage_ms_per_step` and the `roofline` of the dominant kernel; further legs (never `value`):
host-array input (`value_host_arrays`, PCIe inclusive), CRNN only (configs[2]),
CRAFT only (configs[1]), one rank's share of configs[4], and -- on every rank, also at N = 1 -- `cfg5_sharded`:
ONE configs[4] batch of 32 x N pages of 1536x1536 (256 pages at N = 8) through `dist.ShardedPipeline`, each rank's block
resident in its HBM, with the three RCCL result all-gathers INSIDE the timed region (`gather_ms`).  `parity` compares
eight pages of the timed batch with the CPU oracle (page 0 is the `cpu_baseline` run), in both fp32-class arithmetic modes, counting the heat-map
pixels that sit on the other side of a getBoxes threshold (oracle/parity.py).  `odd_sizes` (rank 0, never `value`): the same
pipeline on 32 pages of 750 x 1000 (detector input 1500 x 2000: no pyramid level tiles), images/s and the convolutions' TFLOP/s.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = 32
SIDE = 768
SCALE = 2
WORDS_PER_PAGE = 20       # SURVEY.md 8(d) cfg 4: "~20 words each"
FP32_MFMA_PEAK_TF = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
BF16_MFMA_PEAK_TF = 2500.0  # dense bf16 / fp16 MFMA peak, same table
PARITY_PAGES = (0, 4, 9, 13, 18, 22, 27, 31)  # pages of the timed batch compared with the CPU oracle
HEAT_TOL = 5e-5  # north_star's "stated fp32 tolerance on heatmaps": the floor of oracle.parity.heat_tolerance (1.5e-5 per unit of max |heat|
                 # on the calibrated pages = 6.5e-5, plus an rms bound) -- the bound of tests/test_baseline_sizes_gpu.py


def make_pages(n, side, seed, words=WORDS_PER_PAGE, width=None):
    """Seeded synthetic pages (side x side, or side x width): white background, `words` well-separated rendered words on a
    jittered grid (PIL DejaVuSans if present), so that the detector finds about that many boxes per page."""
    rng = np.random.default_rng(seed)
    try:
        from PIL import Image, ImageDraw, ImageFont

        font_path = "/usr/share/fonts/truetype/dejavu/DejaVuSans.ttf"
        fonts = [ImageFont.truetype(font_path, s) for s in (18, 22, 26)] if os.path.isfile(font_path) else None
    except Exception:  # pragma: no cover
        fonts = None
    alphabet = "abcdefghijklmnopqrstuvwxyz0123456789"
    width = width or side
    pages = np.full((n, side, width, 3), 255, np.uint8)
    cols = max(1, width // 190)
    rows = max(1, -(-words // cols))
    cw, ch = width // cols, side // rows
    for i in range(n):
        cells = [(r, c) for r in range(rows) for c in range(cols)]
        picks = [cells[j] for j in rng.permutation(len(cells))[:words]]
        if fonts:
            im = Image.fromarray(pages[i])
            dr = ImageDraw.Draw(im)
            for r, c in picks:
                word = "".join(rng.choice(list(alphabet), size=int(rng.integers(3, 8))))
                x = c * cw + int(rng.integers(6, max(7, cw - 150)))
                y = r * ch + int(rng.integers(4, max(5, ch - 34)))
                dr.text((x, y), word, fill=(int(rng.integers(0, 90)),) * 3, font=fonts[int(rng.integers(0, 3))])
            pages[i] = np.asarray(im)
        else:
            for r, c in picks:
                w, h = int(rng.integers(40, 120)), int(rng.integers(14, 26))
                x, y = c * cw + int(rng.integers(0, max(1, cw - w))), r * ch + int(rng.integers(0, max(1, ch - h)))
                patch = rng.integers(0, 120, (h, w, 3), dtype=np.uint8)
                patch[:, ::7] = 255
                pages[i, y:y + h, x:x + w] = patch
    return pages


def make_crops(m, seed=3):
    """BASELINE configs[2] / SURVEY.md 8(d) cfg 3: m pre-cropped 31 x 200 gray strips, float32 in [0, 1] = uint8 rendered
    words / 255 (noise strips without PIL)."""
    rng = np.random.default_rng(seed)
    try:
        from PIL import Image, ImageDraw, ImageFont

        font_path = "/usr/share/fonts/truetype/dejavu/DejaVuSans.ttf"
        font = ImageFont.truetype(font_path, 22) if os.path.isfile(font_path) else None
    except Exception:  # pragma: no cover
        font = None
    alphabet = list("abcdefghijklmnopqrstuvwxyz0123456789")
    out = np.empty((m, 31, 200), np.uint8)
    for i in range(m):
        if font is None:
            out[i] = rng.integers(0, 256, (31, 200), dtype=np.uint8)
            continue
        im = Image.new("L", (200, 31), 255)
        word = "".join(rng.choice(alphabet, size=int(rng.integers(3, 11))))
        ImageDraw.Draw(im).text((int(rng.integers(2, 12)), int(rng.integers(0, 5))), word, fill=int(rng.integers(0, 90)), font=font)
        out[i] = np.asarray(im)
    return out.astype(np.float32) / np.float32(255)


CPU_REPEATS = 3      # runs of the page-0 sample (min / median reported; `value` = 1 / median)
CRNN_CPU_CROPS = 64  # "cfg 3 on 64 crops"


def cpu_baseline(craft_w, crnn_w, page, crops=None, cfg1_pages=None):
    """The CPU oracle (torch-CPU + numpy restatement of the reference path; NOT TensorFlow) on bounded samples of the
    BASELINE workloads: (a) one 768x768 page of the timed batch through the whole pipeline, CPU_REPEATS times; (b) the
    recogniser alone on CRNN_CPU_CROPS of the configs[2] crops -> ms/crop; (c) configs[0] in full (4 pages 256x256, scale 2)."""
    import torch
    from oracle import pipeline as opipe, crnn as ocrnn

    cores = min(os.cpu_count() or 1, 32)  # more torch threads than this slow the oracle down
    torch.set_num_threads(cores)
    runs, out, heat = [], None, None
    for _ in range(CPU_REPEATS):
        t = time.perf_counter()
        heat = []
        out = opipe.recognize(craft_w, crnn_w, [page], scale=SCALE, heat_out=heat)
        runs.append(time.perf_counter() - t)
    med = float(np.median(runs))
    res = {"value": 1.0 / med, "unit": "images/s", "cores": cores, "kind": "port",
           "sample": f"1 synthetic {SIDE}x{SIDE} page of the timed batch, scale={SCALE}, full pipeline, {len(out[0])} words, "
                     f"{CPU_REPEATS} runs on the torch-CPU oracle (not TF); value = 1 / median",
           "runs_s": [round(r, 3) for r in runs], "min_s": round(min(runs), 3), "median_s": round(med, 3)}
    extra = {}
    if crops is not None:
        x = np.ascontiguousarray(crops[:CRNN_CPU_CROPS, ..., None])
        ocrnn.crnn_forward(crnn_w, x[:8])  # warm the thread pool
        t = time.perf_counter()
        probs = ocrnn.crnn_forward(crnn_w, x)
        lab = ocrnn.ctc_greedy_decode(probs)
        dtc = time.perf_counter() - t
        res["crnn_ms_per_crop"] = dtc / len(x) * 1e3
        res["crnn_sample"] = f"{len(x)} of the 512 configs[2] crops (31x200), CRNN forward + CTC greedy, {dtc:.2f} s"
        extra["crnn_labels"] = lab
        extra["crnn_probs"] = probs
    if cfg1_pages is not None:
        t = time.perf_counter()
        h1 = []
        o1 = opipe.recognize(craft_w, crnn_w, list(cfg1_pages), scale=SCALE, heat_out=h1)
        dt1 = time.perf_counter() - t
        res["cfg1_images_per_s"] = len(cfg1_pages) / dt1
        res["cfg1_sample"] = (f"BASELINE configs[0] in full: {len(cfg1_pages)} pages {cfg1_pages[0].shape[0]}x{cfg1_pages[0].shape[1]}, "
                              f"scale={SCALE}, {sum(len(g) for g in o1)} words, {dt1:.2f} s")
        extra["cfg1_out"] = o1
        extra["cfg1_heat"] = h1[0]
    return res, out[0], heat[0][0], extra


def parity_of(gpu_page, oracle_page, flipped=None, page=0):
    """One page of the timed batch: GPU pipeline vs the CPU oracle (strings exact, boxes in input pixels).  With
    `flipped` (the heat-map pixels that sit on the other side of a getBoxes threshold, oracle/parity.py) a box the
    oracle has and the GPU has not -- or the reverse -- is accepted only when such a pixel lies in its neighbourhood."""
    gs, os_ = [t for t, _ in gpu_page], [t for t, _ in oracle_page]
    same_n = len(gpu_page) == len(oracle_page)
    diff = None
    if same_n and gpu_page:
        diff = float(max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max()
                         for (_, a), (_, b) in zip(gpu_page, oracle_page)))
    res = {"page": page, "words_gpu": len(gpu_page), "words_oracle": len(oracle_page),
           "strings_equal": gs == os_, "boxes_max_abs_diff_px": diff,
           "ok": bool(same_n and gs == os_ and (diff is None or diff <= 1e-3)),
           "note": "oracle = oracle/ (CPU restatement of the reference path); box tolerance 1e-3 px, strings exact"}
    if flipped is not None and not res["ok"]:
        from oracle.parity import page_report

        rep = page_report(gpu_page, oracle_page, flipped, float(SCALE))
        res["flip_accounting"] = rep
        res["ok"] = rep["ok"]
    if flipped is not None:
        res["flipped_threshold_pixels"] = int(len(flipped))
    return res


def parity_pages(ctx, craft_w, crnn_w, pages, gpu_out, page0_oracle, page0_heat, oracle_cache, alt_split=None):
    """`parity` object of the bench line: pages PARITY_PAGES of the timed batch against the CPU oracle.  The oracle's
    result of every page (words, heat-map) is kept in `oracle_cache` for the fast-mode leg."""
    from oracle import pipeline as opipe
    from oracle.parity import flips, heat_tolerance, heat_within_tolerance

    per_page, ok = [], True
    for i in [q for q in PARITY_PAGES if q < len(pages)]:
        if i == 0:
            want, h_ref = page0_oracle, page0_heat
        else:
            heat = []
            want = opipe.recognize(craft_w, crnn_w, [pages[i]], scale=SCALE, heat_out=heat)[0]
            h_ref = heat[0][0]
        oracle_cache[i] = (want, h_ref)
        big = ctx.resize_pad(pages[i][None], (pages[i].shape[1] * SCALE, pages[i].shape[0] * SCALE))
        h_gpu = ctx.craft_forward(big)[0]
        r = parity_of(gpu_out[i], want, flips(h_gpu, h_ref), page=i)
        # ABSOLUTE heat-map error against north_star's fp32 tolerance: the head is calibrated so that the maps stay O(1)
        # (weights.calibrate_craft_head(top_q=0.9999)), as real CRAFT score maps are
        d = np.abs(h_gpu - h_ref)
        if alt_split is not None:  # the same page's heat-map in the other fp32-class arithmetic
            ctx.set_split_mode(alt_split[0])
            try:
                r["heat_max_abs_err_" + alt_split[0]] = float(np.abs(ctx.craft_forward(big)[0] - h_ref).max())
            finally:
                ctx.set_split_mode(alt_split[1])
        r["heat_max_abs_err"] = float(d.max())
        r["heat_rms_err"] = float(np.sqrt((d.astype(np.float64) ** 2).mean()))
        r["heat_max_abs"] = float(np.abs(h_ref).max())
        near = (np.abs(h_ref - np.float32(0.4)) <= 1.0) | (np.abs(h_ref[..., :1] - np.float32(0.7)) <= 1.0)
        r["heat_max_abs_err_within_1_of_a_threshold"] = float(d[near].max()) if near.any() else 0.0
        r["pixels_within_1_of_a_threshold"] = int(near.sum())
        r.pop("note")
        per_page.append(r)
        r["heat_tolerance_max_rms"] = list(heat_tolerance(h_ref))
        ok = ok and r["ok"] and heat_within_tolerance(h_gpu, h_ref)
    return {"pages": [r["page"] for r in per_page], "ok": bool(ok),
            "words_gpu": sum(r["words_gpu"] for r in per_page), "words_oracle": sum(r["words_oracle"] for r in per_page),
            "strings_equal": all(r["strings_equal"] for r in per_page),
            "boxes_max_abs_diff_px": max([r["boxes_max_abs_diff_px"] for r in per_page if r["boxes_max_abs_diff_px"] is not None],
                                         default=None),
            "heat_max_abs_err": max(r["heat_max_abs_err"] for r in per_page),
            "heat_max_abs": max(r["heat_max_abs"] for r in per_page),
            "heat_rms_err": max(r["heat_rms_err"] for r in per_page),
            "heat_tolerance_abs": max(r["heat_tolerance_max_rms"][0] for r in per_page),
            "heat_tolerance_rms": max(r["heat_tolerance_max_rms"][1] for r in per_page),
            # round 3's RELATIVE criterion (max error / max |heat| <= 5e-5) next to the absolute one
            "heat_max_err_over_max_abs": max(r["heat_max_abs_err"] for r in per_page) / max(max(r["heat_max_abs"] for r in per_page), 1e-30),
            "ok_relative_5e-5": bool(max(r["heat_max_abs_err"] for r in per_page) <= 5e-5 * max(r["heat_max_abs"] for r in per_page)),
            **({("heat_max_abs_err_" + alt_split[0]): max(r["heat_max_abs_err_" + alt_split[0]] for r in per_page)} if alt_split else {}),
            "heat_max_abs_err_within_1_of_a_threshold": max(r["heat_max_abs_err_within_1_of_a_threshold"] for r in per_page),
            "flipped_threshold_pixels": sum(r["flipped_threshold_pixels"] for r in per_page),
            "per_page": per_page,
            "note": "oracle = oracle/ (CPU restatement of the reference path); strings exact, boxes to 1e-3 px, heat-maps to "
                    f"max(5e-5, 1.5e-5 max|heat|) absolute and 1.5e-6 max|heat| rms (oracle/parity.py::heat_tolerance); a missing / extra box is accepted only next to a "
                    "heat-map pixel that lies on the other side of a getBoxes threshold (counted: flipped_threshold_pixels)"}


def fast_mode_leg(ctx, pipe, step, timed, args, world, out_default, pages, oracle_cache):
    """SURVEY 8(f).4 / VERDICT r03 item 2: the opt-in REDUCED-PRECISION mode (KOCR_SPLIT_F16X1: one fp16 piece per operand in
    the Winograd F(4,3) layers) on the same workload -- its throughput and what it costs in accuracy, against the oracle
    pages of the parity leg.  Never `value`."""
    from oracle.parity import flips
    from keras_ocr_amd import evaluation

    ctx.set_split_mode("f16x1")
    try:
        step()
        dt_f, out_f = timed(step, args.steps)
        heat_err, heat_rms, n_flip, n_same, n_oracle, n_fast, same_str = 0.0, 0.0, 0, 0, 0, 0, 0
        true, pred = {}, {}
        for i, (want, h_ref) in sorted(oracle_cache.items()):
            big = ctx.resize_pad(pages[i][None], (pages[i].shape[1] * SCALE, pages[i].shape[0] * SCALE))
            h = ctx.craft_forward(big)[0]
            d = np.abs(h - h_ref)
            heat_err = max(heat_err, float(d.max()))
            heat_rms = max(heat_rms, float(np.sqrt((d.astype(np.float64) ** 2).mean())))
            n_flip += int(len(flips(h, h_ref)))
            got = out_f[i]
            n_oracle += len(want)
            n_fast += len(got)
            for t, b in want:
                dd = [float(np.abs(np.asarray(b, np.float64) - np.asarray(gb, np.float64)).max()) for _, gb in got]
                if dd and min(dd) <= 1e-3:
                    n_same += 1
                    same_str += int(got[int(np.argmin(dd))][0] == t)
            true[i] = [{"text": t, "vertices": np.asarray(b, np.float64)} for t, b in want]
            pred[i] = [{"text": t, "vertices": np.asarray(b, np.float64)} for t, b in got]
        prec = rec_ = None
        if n_oracle and n_fast:
            _, (prec, rec_) = evaluation.score(true, pred)
        same_default = sum(1 for ga, gb in zip(out_default, out_f) for (ta, _), (tb, _) in zip(ga, gb) if ta == tb)
        return {"mode": "f16x1", "value": world * args.batch * args.steps / dt_f, "unit": "images/s",
                "ms_per_step": dt_f / args.steps * 1e3, "words": sum(len(g) for g in out_f),
                "vs_oracle_on_parity_pages": {
                    "heat_max_abs_err": heat_err, "heat_rms_err": heat_rms, "flipped_threshold_pixels": n_flip,
                    "oracle_boxes": n_oracle, "fast_boxes": n_fast, "boxes_identical_to_1e-3_px": n_same,
                    "of_those_strings_identical": same_str,
                    "precision_recall_vs_oracle_iou0.5_similarity0.5": [prec, rec_]},
                "strings_identical_to_default_mode_by_position": same_default,
                "stated_tolerance": "heat-maps within 2e-2 absolute (2e-3 rms) of the oracle on full-size pages with maps of "
                                    "magnitude ~4 -- 500x the fp32-class modes; 5e-3 on the small images of "
                                    "tests/test_split_modes_gpu.py::test_fast_mode_heatmaps_and_boxes; per layer "
                                    "|err| <= 1e-3 |x| conv |w| (same file)",
                "note": "REDUCED PRECISION (relative operand error 2^-12 in the F(4,3) layers): opt-in via "
                        "kocr_set_split_mode(KOCR_SPLIT_F16X1) / KOCR_SPLIT=f16x1, never the default, never `value`"}
    finally:
        ctx.set_split_mode(args.split)


def compact_line(res):
    """The ONE stdout line: the contract's keys, `config`, a trimmed `roofline`, `cpu_baseline`, and one number per extra leg --
    short enough that the driver's record holds it whole.  Everything else goes to stderr / gpurun_out/bench_full.json."""
    keep = ("metric", "value", "unit", "n_gpus", "ranks_seen", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config")
    out = {kk: res[kk] for kk in keep if kk in res}
    rf = res.get("roofline", {})
    out["roofline"] = {kk: rf.get(kk) for kk in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                  "traffic_over_algorithmic", "issued_tflops", "issued_frac_of_pipe_peak",
                                                  "issued_per_algorithmic", "pmc_clock_ghz", "pmc_mfma_pipe_busy",
                                                  "algorithmic_bytes_per_launch", "avg_launch_ms", "launches", "all_conv_tflops")}
    if "cpu_baseline" in res:
        out["cpu_baseline"] = res["cpu_baseline"]
    if "crnn_only" in res:
        out["crnn_only"] = {kk: v for kk, v in res["crnn_only"].items() if kk != "metric"}
    if "cfg1" in res:
        out["cfg1"] = {kk: v for kk, v in res["cfg1"].items() if kk != "workload"}
    if "parity" in res:
        out["parity"] = {kk: v for kk, v in res["parity"].items() if kk not in ("per_page", "note")}
    legs = {}
    for kk in ("value_host_arrays", "cfg2_craft_only", "cfg5_share", "cfg5_sharded", "cfg5_scattered", "odd_sizes", "fast_mode",
               "alt_split_mode"):
        if kk in res and isinstance(res[kk], dict) and "value" in res[kk]:
            legs[kk] = round(res[kk]["value"], 2)
    out["legs_images_per_s"] = legs
    top = sorted(res.get("stage_ms_per_step", {}).items(), key=lambda kv: -kv[1])[:8]
    out["top_kernels_ms_per_step"] = dict(top)
    out["full_record"] = "stderr line '[bench full] {...}' and gpurun_out/bench_full.json"
    return out


def respawn_under_torchrun(args, env):
    """`python bench.py --gpus N` outside torchrun: launch N ranks of this script on this node."""
    import socket

    have = env.device_count()
    if have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node; refusing to "
                         "report a multi-GPU number from fewer devices")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


class GpuEnv:
    """The device plumbing of the bench: torch-ROCm tensors in HBM, torch.distributed over RCCL (backend "nccl").
    tests/test_bench_cpu.py replaces it by a host double (gloo, mocked libkocr context) to execute main() at world
    size 2 without a GPU -- everything else in main() is the code the driver runs."""
    backend = "nccl"

    def check(self, local_rank):
        import torch

        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an AMD GPU (no CPU fallback)")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
        torch.cuda.set_device(local_rank)

    def device_count(self):
        import torch

        return torch.cuda.device_count()

    def context(self, k, local_rank):
        import torch

        ctx = k.Context(local_rank)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        return ctx

    def sync(self):
        import torch

        torch.cuda.synchronize()

    def to_dev(self, arr):
        import torch

        return torch.from_numpy(arr).cuda()

    def empty(self, shape, dtype):
        import torch

        return torch.empty(shape, dtype=dtype, device="cuda")

    def rand(self, shape):
        import torch

        return torch.rand(shape, dtype=torch.float32, device="cuda")

    def scalar(self, v):
        import torch

        return torch.tensor([v], dtype=torch.float64, device="cuda")

    def cpu_baseline_ok(self):
        return True


def main(argv=None, env=None):
    env = env or GpuEnv()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--side5", type=int, default=1536, help="page side of the configs[4] legs (1536; smaller only in tests)")
    ap.add_argument("--cfg5-pages", type=int, default=0,
                    help="total pages of the ONE sharded configs[4] batch (default batch x N: equal blocks); any other number "
                         "gives ragged / empty shards (tests)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--split", choices=["bf16x3", "f16x2"], default="f16x2",
                    help="arithmetic of the wide convolutions for the headline number (include/kocr.h KOCR_SPLIT_*)")
    ap.add_argument("--alt-mode", action="store_true",
                    help="also time the other fp32-class split mode (bf16x3 everywhere: the round-3 default) on the same workload")
    ap.add_argument("--no-alt-mode", action="store_true", help="(default; kept for old command lines)")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[1] / configs[4]-share / host-array legs")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic with two rocprofv3 --pmc passes of a short child run; read the "
                         "committed profiles/*_pmc_traffic.json instead")
    ap.add_argument("--profile-all", action="store_true",
                    help="keep the HIP-event profiler on for the WHOLE process (headline loop included) and report the "
                         "per-launch averages over every launch: the numbers a rocprofv3 --stats / --pmc run of the "
                         "same command must agree with (scripts/profile_final.sh)")
    args = ap.parse_args(argv)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        respawn_under_torchrun(args, env)
    if "WORLD_SIZE" not in os.environ and "MASTER_PORT" not in os.environ:
        import socket  # single process: rendezvous on a free local port (a fixed one may still be in TIME_WAIT)

        with socket.socket() as sck:
            sck.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sck.getsockname()[1])
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # stdout carries exactly ONE JSON line: RCCL / HIP banners written to fd 1 by native code go to stderr
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import keras_ocr_amd as k

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    env.check(local_rank)
    rank, world = k.dist.init_from_env(backend=env.backend, force=True)  # RCCL, also at N = 1
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but the process group has {world} rank(s)")
    seen = k.dist.ranks_seen()  # all-reduce of 1 over RCCL
    if seen != world:
        raise SystemExit(f"bench.py: all-reduce saw {seen} ranks, expected {world}")
    ctx = env.context(k, local_rank)
    ctx.set_split_mode(args.split)
    ctx.profile_reset()
    ctx.profile_enable(bool(args.profile_all))
    prof_process = {}

    def fold_process_profile():
        """--profile-all: accumulate the profiler rows seen so far (every launch of the process) and restart it"""
        if not args.profile_all:
            return
        for kk, v in ctx.profile_report().items():
            row = prof_process.setdefault(kk, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            for f in row:
                row[f] += v[f]
        ctx.profile_reset()

    pages = make_pages(args.batch, SIDE, seed=4 + rank)
    craft_w = k.weights.synthetic_craft_weights(1234)
    crnn_w = k.weights.synthetic_crnn_weights(4321)
    # calibrate the random-init head (same on every rank) so that the detector emits ~20 boxes per page
    ctx.load_craft(craft_w)
    cal_pages = make_pages(8, SIDE, seed=4)  # = the first 8 pages of rank 0's timed batch (the same on every rank)
    sample = ctx.resize_pad(cal_pages, (SIDE * SCALE, SIDE * SCALE))
    raw = ctx.craft_forward(sample)
    best = None
    for frac in (0.012, 0.0095, 0.008, 0.007, 0.0062, 0.0055, 0.0049, 0.0044, 0.0039, 0.0034, 0.003, 0.0025):
        # top_q = 0.9999 keeps the gain (and with it the maps) O(1) -- see weights.calibrate_craft_head
        cand = k.weights.calibrate_craft_head(craft_w, raw, text_frac=frac, link_frac=frac / 3, top_q=0.9999)
        a = cand["conv_cls.8.weight"].reshape(2, -1)[:, :1] / craft_w["conv_cls.8.weight"].reshape(2, -1)[:, :1]
        heat = (raw - craft_w["conv_cls.8.bias"]) * a.ravel() + cand["conv_cls.8.bias"]
        nb = np.mean([len(b) for b in ctx.get_boxes(heat.astype(np.float32))])
        if rank == 0:
            print(f"[bench] head calibration: text_frac {frac}: {nb:.1f} boxes/page", file=sys.stderr)
        if best is None or abs(nb - WORDS_PER_PAGE) < abs(best[0] - WORDS_PER_PAGE):
            best = (nb, cand)
    craft_w = best[1]
    det = k.detection.Detector(weights=craft_w, ctx=ctx)
    rec = k.recognition.Recognizer(weights=crnn_w, ctx=ctx)
    pipe = k.pipeline.Pipeline(detector=det, recognizer=rec, scale=SCALE)

    d_pages = env.to_dev(pages)
    n, h, w = args.batch, SIDE, SIDE

    def step():
        return pipe.recognize_device(d_pages.data_ptr(), n, h, w)

    def barrier():
        torch.distributed.barrier()
        env.sync()

    def timed(fn, steps):
        """barrier + synchronize on both sides, max over ranks"""
        barrier()
        t0 = time.perf_counter()
        res = None
        for _ in range(steps):
            res = fn()
        barrier()
        tt = env.scalar(time.perf_counter() - t0)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        return float(tt.item()), res

    def timed_local(fn, steps):
        """rank-local timing for the rank-0-only legs (no collective: the other ranks sit in the final barrier)"""
        env.sync()
        t0 = time.perf_counter()
        res = None
        for _ in range(steps):
            res = fn()
        env.sync()
        return time.perf_counter() - t0, res

    out = None
    for _ in range(args.warmup):
        out = step()
    # ---- headline: profiler off ----------------------------------------------------------------------
    dt, out = timed(step, args.steps)
    n_words = sum(len(g) for g in out)
    # ---- the same K steps with the HIP-event profiler on: stage times + roofline of the dominant kernel --
    fold_process_profile()
    ctx.profile_reset()
    ctx.profile_enable(True)
    dt_prof, _ = timed(step, args.steps)
    prof = {kk: v for kk, v in ctx.profile_report().items() if v["launches"]}
    fold_process_profile()
    ctx.profile_enable(bool(args.profile_all))

    # ---- other arithmetic mode of the wide convolutions, same workload; reported beside the headline ------
    alt = None
    if args.alt_mode and not args.no_alt_mode:
        alt_mode = "f16x2" if args.split == "bf16x3" else "bf16x3"
        ctx.set_split_mode(alt_mode)
        step()
        dt_alt, out_alt = timed(step, args.steps)
        same = sum(1 for ga, gb in zip(out, out_alt) for (ta, _), (tb, _) in zip(ga, gb) if ta == tb)
        alt = {"mode": alt_mode, "value": world * args.batch * args.steps / dt_alt, "unit": "images/s",
               "ms_per_step": dt_alt / args.steps * 1e3,
               "words": sum(len(g) for g in out_alt), "identical_strings_vs_headline_mode": same,
               "note": "f16x2 = the Winograd F(4,3) layers on the fp16 cores: 2 round-to-nearest fp16 pieces per fp32 operand, "
                       "3 products, exact per-image / per-cout power-of-two scaling (everything else bf16x3); bf16x3 = 3 exact "
                       "bf16 pieces, 6 products everywhere; both within fp32 round-off of an fp64 reference "
                       "(tests/test_conv_gpu.py, tests/test_split_modes_gpu.py); the whole GPU suite passes in either mode"}
        ctx.set_split_mode(args.split)

    extra = {}
    if not args.no_extra:
        # host-array input, the reference's calling convention: H2D of the batch inside the timed region
        pipe.recognize(pages)
        dt_host, out_host = timed(lambda: pipe.recognize(pages), args.steps)
        extra["value_host_arrays"] = {
            "value": world * args.batch * args.steps / dt_host, "unit": "images/s", "ms_per_step": dt_host / args.steps * 1e3,
            "same_result_as_device_path": [[t for t, _ in g] for g in out_host] == [[t for t, _ in g] for g in out],
            "note": "pipe.recognize(numpy pages): 56.6 MB H2D per 32 pages + the result D2H inside the timed region"}
    if not args.no_extra:
        # BASELINE configs[4] as ONE batch over the whole job: 32 x N pages of 1536x1536 (256 at N = 8), scale 3 (capped:
        # detector input 2048x2048), each rank's contiguous block of 32 already in its HBM; dist.ShardedPipeline runs the
        # local chain and the three RCCL all-gathers of the packed results -- all inside the timed region
        side5 = args.side5
        p5s_host = make_pages(args.batch, side5, seed=5 + rank, words=80)
        p5s = env.to_dev(p5s_host)
        sp = k.dist.ShardedPipeline(k.pipeline.Pipeline(detector=det, recognizer=rec, scale=3))
        n_tot = args.cfg5_pages if args.cfg5_pages > 0 else args.batch * world
        if -(-n_tot // world) > args.batch:
            raise SystemExit("bench.py --cfg5-pages: a rank's block must fit in its resident batch (ceil(pages / N) <= --batch)")
        sp.recognize_device(p5s.data_ptr(), n_tot, side5, side5)
        tm = {}
        reps = 2
        dt5, o5s = timed(lambda: sp.recognize_device(p5s.data_ptr(), n_tot, side5, side5, timing=tm), reps)
        extra["cfg5_sharded"] = {
            "workload": f"BASELINE configs[4]: ONE batch of {n_tot} pages 1536x1536 (scale 3 -> 2048x2048) sharded over {world} "
                        "rank(s) by dist.ShardedPipeline, contiguous blocks resident in each rank's HBM; result all-gathers "
                        "(counts, boxes, label rows) over RCCL inside the timed region; every rank ends with the full result",
            "value": n_tot * reps / dt5, "unit": "images/s (whole job)", "ms_per_batch": dt5 / reps * 1e3,
            "gather_ms": tm.get("gather_s", 0.0) / reps * 1e3, "pages_returned_on_every_rank": len(o5s),
            "words": sum(len(g) for g in o5s), "gather_payload_bytes_per_rank": tm.get("gather_payload_bytes_per_rank"),
            "backend": torch.distributed.get_backend()}
        # the same batch when it STARTS on rank 0 -- the raw uint8 pages are scattered over RCCL (one
        # torch.distributed.scatter of equal blocks) before the local chains; scatter + chains + gathers inside the timed region
        if world > 1:
            full = torch.cat([torch.empty_like(p5s) for _ in range(world)]) if rank == 0 else None
            if rank == 0:
                # the blocks the ranks hold in the resident leg, in shard order (block r = the first ceil(n / N) pages of rank r's batch)
                per5 = -(-n_tot // world)
                full[:per5] = p5s[:per5]
                for r in range(1, world):
                    full[r * per5:(r + 1) * per5] = env.to_dev(make_pages(args.batch, side5, seed=5 + r, words=80))[:per5]
                full = full[:n_tot].contiguous()
        else:
            full = p5s[:n_tot]
        tms = {}
        sp.recognize_scattered(full, n_tot, side5, side5, src_rank=0)
        dt5s, o5c = timed(lambda: sp.recognize_scattered(full, n_tot, side5, side5, src_rank=0, timing=tms), reps)
        extra["cfg5_scattered"] = {
            "workload": f"the same batch of {n_tot} pages starting in rank 0's HBM: torch.distributed.scatter (RCCL send/recv over "
                        "xGMI) of the raw uint8 pages before the resize, then the local chains and the result all-gathers",
            "value": n_tot * reps / dt5s, "unit": "images/s (whole job)", "ms_per_batch": dt5s / reps * 1e3,
            "scatter_ms": tms.get("scatter_s", 0.0) / reps * 1e3, "gather_ms": tms.get("gather_s", 0.0) / reps * 1e3,
            "scatter_bytes_sent_by_rank0": tms.get("scatter_bytes_sent"),
            "same_strings_as_resident_blocks": [[t for t, _ in g] for g in o5c] == [[t for t, _ in g] for g in o5s]}
        del p5s, full
    # Every leg that involves a collective is over.  The other ranks leave NOW (one last barrier, then they destroy their
    # side of the group and return): rank 0's solo legs below -- CRNN / CRAFT only, the configs[4] share, the PMC child
    # runs, the CPU oracle (tens of seconds) -- must not run while peers sit in a collective waiting for a timeout.
    fold_process_profile()
    torch.distributed.barrier()
    if rank != 0:
        torch.distributed.destroy_process_group()
        ctx.close()
        return None
    crnn_us_per_crop = None
    if rank == 0:
        # secondary BASELINE metric: ms/crop of the CRNN alone (configs[2]: 512 pre-cropped 31x200 strips)
        m = 512
        crops_host = make_crops(m, seed=3)
        crops = env.to_dev(crops_host)
        labels = env.empty((m, 48), torch.int32)
        ctx.crnn_forward_device(crops.data_ptr(), m, labels.data_ptr())
        env.sync()
        t1 = time.perf_counter()
        for _ in range(3):
            ctx.crnn_forward_device(crops.data_ptr(), m, labels.data_ptr())
        env.sync()
        crnn_us_per_crop = (time.perf_counter() - t1) / 3 / m * 1e6
        # the first CRNN_CPU_CROPS crops with their probabilities, for the comparison with the oracle (cpu_baseline leg)
        crnn_probs = env.empty((CRNN_CPU_CROPS, 48, 37), torch.float32)
        crnn_lab64 = env.empty((CRNN_CPU_CROPS, 48), torch.int32)
        ctx.crnn_forward_device(crops.data_ptr(), CRNN_CPU_CROPS, crnn_lab64.data_ptr(), crnn_probs.data_ptr())
        env.sync()
        crnn_probs, crnn_lab64 = crnn_probs.cpu().numpy(), crnn_lab64.cpu().numpy()
        del crops, labels
    if rank == 0 and not args.no_extra:
        # configs[1]: CRAFT detector only, batch 8 x 768x768 (no resize), heat-maps stay in HBM
        x8 = env.to_dev(make_pages(8, SIDE, seed=2))
        heat8 = env.empty((8, SIDE // 2, SIDE // 2, 2), torch.float32)
        ctx.craft_forward_device(x8.data_ptr(), 0, 8, SIDE, SIDE, heat8.data_ptr())
        env.sync()
        t1 = time.perf_counter()
        for _ in range(5):
            ctx.craft_forward_device(x8.data_ptr(), 0, 8, SIDE, SIDE, heat8.data_ptr())
        env.sync()
        d2 = (time.perf_counter() - t1) / 5
        extra["cfg2_craft_only"] = {"workload": "BASELINE configs[1]: CRAFT forward only, 8 x 768x768 u8, input and heat-maps in HBM",
                                    "value": 8 / d2, "unit": "images/s", "ms_per_batch": d2 * 1e3,
                                    "algorithmic_tflops": 8 * 419.624e9 / d2 / 1e12}
        del x8, heat8
        # configs[4], one rank's share: 32 x 1536x1536 pages, scale=3 -> capped to 2048x2048 (one micro-batch)
        p5 = env.to_dev(make_pages(32, args.side5, seed=5, words=80))
        pipe3 = k.pipeline.Pipeline(detector=det, recognizer=rec, scale=3)
        pipe3.recognize_device(p5.data_ptr(), 32, args.side5, args.side5)
        env.sync()
        t1 = time.perf_counter()
        for _ in range(2):
            o5 = pipe3.recognize_device(p5.data_ptr(), 32, args.side5, args.side5)
        env.sync()
        d5 = (time.perf_counter() - t1) / 2
        extra["cfg5_share"] = {"workload": "BASELINE configs[4] per-GPU share: 32 pages 1536x1536, scale=3 "
                                           "(internally 2048x2048), full pipeline, single rank",
                               "value": 32 / d5, "unit": "images/s per GPU", "ms_per_32_pages": d5 * 1e3,
                               "words": sum(len(g) for g in o5),
                               # SURVEY 8(d) cfg 5: the collective payload of the result gather
                               "gather_payload_bytes_per_rank": k.dist.packed_payload_bytes(32, sum(len(g) for g in o5))}
        del p5
        # a page size NO pyramid level of which tiles -- 32 pages of 750 x 1000 at scale 2 -> detector input
        # 1500 x 2000, levels 750 x 1000 ... 93 x 125 (tools.resize_image hands the detector any int(W s) x int(H s),
        # tools.py:387-397).  Reported: images/s and the algorithmic TFLOP/s of all convolutions of one profiled step, beside
        # the headline's.  Never `value`.
        po = env.to_dev(make_pages(args.batch, 750, seed=6, words=26, width=1000))
        pipe.recognize_device(po.data_ptr(), args.batch, 750, 1000)
        dto, oo = timed_local(lambda: pipe.recognize_device(po.data_ptr(), args.batch, 750, 1000), 2)
        fold_process_profile()
        ctx.profile_reset()
        ctx.profile_enable(True)
        pipe.recognize_device(po.data_ptr(), args.batch, 750, 1000)
        env.sync()
        pro = {kk: v for kk, v in ctx.profile_report().items() if v["launches"]}
        fold_process_profile()
        ctx.profile_enable(bool(args.profile_all))
        cms = sum(v["ms"] for kk, v in pro.items() if kk.startswith("conv_"))
        cfl = sum(v["flops"] for kk, v in pro.items() if kk.startswith("conv_"))
        extra["odd_sizes"] = {
            "workload": f"Pipeline.recognize, {args.batch} pages 750 x 1000, scale 2 -> detector input 1500 x 2000 (no pyramid level is a "
                        "multiple of the 4 x 64 / 8 x 32 tiles; the 1/8 level's width is not a multiple of 4, the 1/16 level's is odd)",
            "value": args.batch * 2 / dto, "unit": "images/s", "ms_per_step": dto / 2 * 1e3, "words": sum(len(g) for g in oo),
            "megapixels_per_s": args.batch * 2 / dto * 3.0, "all_conv_tflops": cfl / (cms * 1e-3) / 1e12 if cms else None,
            "kernels_ms": {kk: round(v["ms"], 3) for kk, v in sorted(pro.items(), key=lambda kv: -kv[1]["ms"])[:12]}}
        del po

    fold_process_profile()
    if rank == 0:
        dom = max((kv for kv in prof.items() if kv[0].startswith("conv_")), key=lambda kv: kv[1]["ms"])
        name, r = dom
        achieved = r["flops"] / (r["ms"] * 1e-3) / 1e12
        conv_ms = sum(v["ms"] for kk, v in prof.items() if kk.startswith("conv_"))
        conv_fl = sum(v["flops"] for kk, v in prof.items() if kk.startswith("conv_"))
        # issued matrix-core work per algorithmic FLOP of the kernel family
        issue = k.perfmodel.issued_per_algorithmic(name)
        split = issue["pipe"] != "fp32"
        executed = achieved * issue["factor"]
        peak = BF16_MFMA_PEAK_TF if split else FP32_MFMA_PEAK_TF
        stage_ms = {kk: round(v["ms"] / args.steps, 3) for kk, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
        # HBM traffic of the dominant kernel: PMC passes cannot run inside this process; the number
        # is read from the committed summary of scripts/pmc_bench.sh over this same command
        traffic, traffic_src, traffic_live = None, "no profiles/*_pmc_traffic.json entry for this kernel", None
        import glob
        if world == 1 and not args.no_live_traffic:
            # live: FETCH_SIZE and WRITE_SIZE passes (one counter per run, kernel-trace only) over a short child run of
            # this same command; average over every launch of the kernel in that child process
            try:
                traffic_live = k.pmc.measure_traffic(os.path.abspath(__file__), name)
                traffic = traffic_live["hbm_bytes_per_launch"]
                traffic_src = (f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes of "
                               f"`bench.py --steps 1 --warmup 1 --profile-all`, average over {traffic_live['launches']} launches")
            except Exception as e:  # noqa: BLE001 -- fall back to the committed summary, say why
                traffic_src = f"live PMC passes failed ({type(e).__name__}: {str(e)[:120]}); "
        for cand in ([] if traffic is not None else sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True)):
            pm = json.load(open(cand))
            row = pm.get("by_prof_name", {}).get(name)
            if row and "hbm_bytes_per_launch" in row:
                traffic = row["hbm_bytes_per_launch"]
                traffic_src = (traffic_src if traffic_src.startswith("live PMC") else "") + "profiles/" + os.path.basename(cand) + \
                    f", average over {row['launches']} launches of the bench process"
                break
        res = {
            "metric": "images/sec end-to-end Pipeline.recognize() @768x768",
            "value": world * args.batch * args.steps / dt,
            "unit": "images/s",
            "n_gpus": world,
            "ranks_seen": seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_profiled": dt_prof / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32 (fp32 operands as 3 exact bf16 pieces, 6 MFMA products, fp32 accumulate; fp32-class error vs fp64: "
                      "tests/test_conv_gpu.py)")
            if args.split == "bf16x3" else
            ("f32 (fp32 operands as 2 fp16 pieces [F(4,3) / head layers] or 3 bf16 pieces [1x1, dilated], 3 / 6 MFMA products, fp32 "
             "accumulate; fp32-class error vs fp64: tests/test_conv_gpu.py)"),
            "data": "synthetic (seeded rendered-text pages; random-init weights of the reference "
                    "architectures, detector head calibrated to emit word boxes)",
            "config": {"workload": f"Pipeline.recognize full pipeline, batch {args.batch} x {SIDE}x{SIDE} RGB u8 per GPU, "
                                   f"scale={SCALE} (detector input {SIDE*SCALE}x{SIDE*SCALE}), BASELINE configs[3]",
                       "global_batch": world * args.batch, "words_per_batch": n_words,
                       "parallelism": f"dp{world} (images sharded, one process per GPU, RCCL process group; no data-path collective)",
                       "split_mode": args.split},
            "roofline": {"bound": "mfma", "kernel": name,
                         "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak,
                         "note": "achieved = ALGORITHMIC direct-convolution fp32 FLOPs of the launches (SURVEY.md 8(d)) / kernel "
                                 "time (HIP events, profiled loop of the same K steps); peak = dense peak of the matrix pipe the "
                                 "kernel runs on; issued_* = the matrix-core FLOPs the kernel actually issues for them "
                                 "(algorithmic x " + issue["why"] + "): the utilisation of that pipe, not the roofline fraction",
                         "issued_tflops": executed,
                         "issued_frac_of_pipe_peak": executed / peak,
                         "algorithmic_fp32_tflops": achieved,
                         "algorithmic_frac_of_bf16_peak": achieved / BF16_MFMA_PEAK_TF,
                         "algorithmic_vs_fp32_mfma_peak": achieved / FP32_MFMA_PEAK_TF,
                         "issued_per_algorithmic": issue["factor"],
                         "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE*2 + WRITE_SIZE, " + traffic_src + ")",
                         # traffic / the algorithmic bytes of THE SAME launches (the PMC child run's own profiler rows);
                         # `algorithmic_bytes_per_launch` below belongs to the headline loop's launches, a different set
                         "traffic_over_algorithmic": (traffic_live or {}).get("traffic_over_algorithmic"),
                         "traffic_algorithmic_bytes_same_launches": (traffic_live or {}).get("algorithmic_bytes_per_launch_same_process"),
                         "pmc_clock_ghz": (traffic_live or {}).get("clock_ghz"),
                         "pmc_mfma_pipe_busy": (traffic_live or {}).get("mfma_pipe_busy"),
                         "traffic_live": ({kk: traffic_live[kk] for kk in ("fetch_bytes_per_launch", "write_bytes_per_launch", "launches",
                                                                           "calibration_maxpool2x2_read_over_write") if kk in traffic_live}
                                          if traffic_live else None),
                         "algorithmic_bytes_per_launch": r["bytes"] / r["launches"],
                         "avg_launch_ms": r["ms"] / r["launches"], "launches": r["launches"],
                         "all_conv_tflops": conv_fl / (conv_ms * 1e-3) / 1e12},
            "stage_ms_per_step": stage_ms,
            "headline_all_conv_tflops": conv_fl / (conv_ms * 1e-3) / 1e12,
            "crnn_only": {"metric": "ms/crop CRNN (BASELINE configs[2]: 512 crops 31x200, CTC greedy)",
                          "value": crnn_us_per_crop / 1e3, "unit": "ms/crop",
                          "fp32_mfma_floor_ms": 13.444e9 / (FP32_MFMA_PEAK_TF * 1e12) * 1e3},
        }
        res.update(extra)
        if alt is not None:
            res["alt_split_mode"] = alt
        if not args.no_cpu_baseline and env.cpu_baseline_ok():
            # BASELINE configs[0]: Pipeline.recognize on 4 synthetic 256x256 pages, scale 2 -- the reference's own CPU-runnable
            # case: GPU and oracle IN FULL, strings / boxes compared
            pages1 = make_pages(4, 256, seed=1, words=3)
            pipe.recognize(pages1)
            dt1, out1 = timed_local(lambda: pipe.recognize(pages1), 5)
            res["cpu_baseline"], oracle_page, oracle_heat, cpu_x = cpu_baseline(craft_w, crnn_w, pages[0], crops=crops_host,
                                                                              cfg1_pages=pages1)
            from oracle.parity import flips as _flips

            per1 = []
            for i in range(len(pages1)):
                big1 = ctx.resize_pad(pages1[i][None], (256 * SCALE, 256 * SCALE))
                per1.append(parity_of(out1[i], cpu_x["cfg1_out"][i], _flips(ctx.craft_forward(big1)[0], cpu_x["cfg1_heat"][i]), page=i))
            res["cfg1"] = {"workload": "BASELINE configs[0]: Pipeline.recognize on 4 synthetic 256x256 pages, scale=2 (host arrays in, "
                                       "H2D + D2H inside the timed region)",
                           "value": 4 * 5 / dt1, "unit": "images/s", "cpu_images_per_s": res["cpu_baseline"]["cfg1_images_per_s"],
                           "words_gpu": sum(len(g) for g in out1), "words_oracle": sum(len(g) for g in cpu_x["cfg1_out"]),
                           "strings_equal": all(r["strings_equal"] for r in per1),
                           "boxes_max_abs_diff_px": max([r["boxes_max_abs_diff_px"] for r in per1 if r["boxes_max_abs_diff_px"] is not None],
                                                        default=None),
                           "ok": all(r["ok"] for r in per1)}
            # configs[2] against the oracle on the crops the CPU figure was timed on
            want_p = cpu_x["crnn_probs"]
            srt = np.sort(want_p, -1)
            safe = ((srt[..., -1] - srt[..., -2]) > 1e-3).all(1)
            res["crnn_only"]["cpu_ms_per_crop"] = res["cpu_baseline"]["crnn_ms_per_crop"]
            res["crnn_only"]["parity"] = {"crops": int(len(want_p)), "max_abs_prob_err": float(np.abs(crnn_probs - want_p).max()),
                                          "rows_with_margin_gt_1e-3": int(safe.sum()),
                                          "label_rows_identical": int((crnn_lab64 == cpu_x["crnn_labels"]).all(1).sum()),
                                          "ok": bool(np.abs(crnn_probs - want_p).max() <= 1e-4 and
                                                     np.array_equal(crnn_lab64[safe], cpu_x["crnn_labels"][safe]))}
            oracle_cache = {}
            res["parity"] = parity_pages(ctx, craft_w, crnn_w, pages, out, oracle_page, oracle_heat, oracle_cache,
                                         alt_split=("bf16x3" if args.split == "f16x2" else "f16x2", args.split))
            if not args.no_extra:
                fold_process_profile()
                res["fast_mode"] = fast_mode_leg(ctx, pipe, step, timed_local, args, world, out, pages, oracle_cache)
        fold_process_profile()  # the parity leg's single-page detector forwards are launches of this process too
        if args.profile_all and name in prof_process:
            pr = prof_process[name]
            res["roofline"]["process"] = {
                "note": "--profile-all: HIP-event averages over EVERY launch of this kernel in the process (calibration, "
                        "warm-up, headline and profiled loops, extra legs, the parity leg's single-page forwards) -- the set a "
                        "rocprofv3 run of this command sees",
                "launches": pr["launches"], "avg_launch_ms": pr["ms"] / pr["launches"],
                "algorithmic_bytes_per_launch": pr["bytes"] / pr["launches"],
                "algorithmic_fp32_tflops": pr["flops"] / (pr["ms"] * 1e-3) / 1e12}
        # BOTH BASELINE metrics and their CPU figures inside `config` / `cpu_baseline` (objects the driver's record keeps whole)
        res["config"]["crnn_ms_per_crop"] = res["crnn_only"]["value"]
        res["config"]["crnn_workload"] = "BASELINE configs[2]: 512 crops 31x200, CRNN forward + CTC greedy, crops and labels in HBM"
        res["config"]["parity"] = ("oracle-exact (strings, boxes, crops bit-exact to oracle/; heat-maps within "
                                   "max(5e-5, 1.5e-5 max|heat|) absolute, 1.5e-6 max|heat| rms); cv2/TF numerics unpinned (no TF / cv2 / weights in the image: "
                                   "tests/golden/make_golden_real.py is the one-command pin for a host that has them)")
        if "cpu_baseline" in res:
            res["config"]["crnn_ms_per_crop_cpu"] = res["cpu_baseline"].get("crnn_ms_per_crop")
            res["config"]["cfg1_images_per_s"] = {"gpu": res["cfg1"]["value"], "cpu": res["cfg1"]["cpu_images_per_s"],
                                                  "ok_vs_oracle": res["cfg1"]["ok"]}
        json_out.write(json.dumps(compact_line(res)) + "\n")
        json_out.flush()
        # the full record (per-page parity, stage times, every leg's notes): stderr and gpurun_out/bench_full.json
        full = json.dumps(res)
        print("[bench full] " + full, file=sys.stderr)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_full.json"), "w") as f:
                f.write(full + "\n")
        except OSError:
            pass
    torch.distributed.destroy_process_group()
    ctx.close()
    return res if rank == 0 else None


if __name__ == "__main__":
    main()

"""HBM-side traffic of libkocr's kernels from rocprofv3 PMC passes (used by bench.py and scripts/pmc_traffic_summary.py).

Collection and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): one counter per run
(`rocprofv3 --kernel-trace --pmc FETCH_SIZE`, then `... WRITE_SIZE`: kernel-trace only, no other trace domain);
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts a 128-B request as 64 B, i.e. reports exactly
half of the bytes of a wide (16 B per lane) coalesced read stream, so it is doubled.  The factor is checked on each run's
own `maxpool2x2` launches, a pure float4 streaming kernel that reads exactly 4x what it writes.
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

# libkocr's profiler row names (what bench.py's roofline object is keyed on) -> prefix of the kernel name rocprofv3 prints
PROF_TO_KERNEL = {
    "conv_w4s_256x128": "void conv_w43_kernel<0, 0, 0", "conv_w4s_256x128_pool": "void conv_w43_kernel<1",
    "conv_w4s_256x128_dil": "void conv_w43_kernel<0, 0, 1", "conv_w4s_512x64": "void conv_w43n_kernel<0",
    "conv_w4s_512x64_pool": "void conv_w43n_kernel<1",
    "conv_w4v_256x128": "void conv_w43v_kernel<0, 1", "conv_w4v_256x128_pool": "void conv_w43v_kernel<1, 1",
    "conv_w4t_256x128": "void conv_w43v_kernel<0, 2",
    # round 4: the same arrangements in fp16 arithmetic (h: two pieces, q: one piece), conv_w43vh_kernel<POOL, GEO, NP, DBG>
    # (round 5: a fifth / third template argument MODE: 0 = exact tiling, 1 = ragged images "_rag", 2 = cell grids "_cells")
    "conv_w4hv_256x128": "void conv_w43vh_kernel<0, 1, 2, 0, 0>", "conv_w4hv_256x128_pool": "void conv_w43vh_kernel<1, 1, 2, 0, 0>",
    "conv_w4ht_256x128": "void conv_w43vh_kernel<0, 2, 2, 0, 0>",
    "conv_w4hv_256x128_rag": "void conv_w43vh_kernel<0, 1, 2, 0, 1>", "conv_w4hv_256x128_pool_rag": "void conv_w43vh_kernel<1, 1, 2, 0, 1>",
    "conv_w4ht_256x128_rag": "void conv_w43vh_kernel<0, 2, 2, 0, 1>",
    "conv_w4hv_256x128_cells": "void conv_w43vh_kernel<0, 1, 2, 0, 2>", "conv_w4hv_256x128_pool_cells": "void conv_w43vh_kernel<1, 1, 2, 0, 2>",
    "conv_w4ht_256x128_cells": "void conv_w43vh_kernel<0, 2, 2, 0, 2>",
    "conv_w4qv_256x128": "void conv_w43vh_kernel<0, 1, 1, 0, 0>", "conv_w4qv_256x128_pool": "void conv_w43vh_kernel<1, 1, 1, 0, 0>",
    "conv_w4qt_256x128": "void conv_w43vh_kernel<0, 2, 1, 0, 0>",
    "conv_w4hf_256x128": "void conv_w43fh_kernel<2, 0>", "conv_w4qf_256x128": "void conv_w43fh_kernel<1, 0>",
    "conv_w4hf_256x128_dil": "void conv_w43fh_kernel<2, 1>",
    "conv_w4hr_256x64": "void conv_w43rh_kernel<0, 2, 0>", "conv_w4hr_256x64_pool": "void conv_w43rh_kernel<1, 2, 0>",
    "conv_w4hr_256x64_rag": "void conv_w43rh_kernel<0, 2, 1>", "conv_w4hr_256x64_pool_rag": "void conv_w43rh_kernel<1, 2, 1>",
    "conv_w4qr_256x64": "void conv_w43rh_kernel<0, 1, 0>", "conv_w4qr_256x64_pool": "void conv_w43rh_kernel<1, 1, 0>",
    "conv_w4s_256x64": "void conv_w43r_kernel<0", "conv_w4s_256x64_pool": "void conv_w43r_kernel<1",
    "conv_ws_128x128": "void conv_ws_kernel<0, 1, 4, 0", "conv_ws_128x128_pool": "void conv_ws_kernel<1, 1, 4, 0",
    "conv_ws_256x64": "void conv_ws_kernel<0, 2, 2, 0", "conv_ws_256x64_pool": "void conv_ws_kernel<1, 2, 2, 0",
    "conv_ds_256x128": "void conv_ds_kernel<1, 4, 0, 0", "conv_ds_512x64": "void conv_ds_kernel<2, 2, 0, 0",
    # exact 2x up-sampling (every CRAFT shape); the generic-ratio variant <..., 1> shares the profiler row
    "conv_ds_256x128_up": "void conv_ds_kernel<1, 4, 0, 2", "conv_ds_512x64_up": "void conv_ds_kernel<2, 2, 0, 2",
    "conv_hh_256x32": "conv_hsh_kernel", "conv_hs_256x32": "conv_hs_kernel", "conv_hs_256x16": "conv_hs16_kernel", "conv_hs_first_256x64": "conv_first_kernel", "conv_k5_352x16": "conv_k5_kernel",
}


def load_counters(directory):
    """Sum every counter of every `*counter_collection.csv` under `directory` per kernel name.
    -> ({kernel: {counter: sum, "_ns_<counter>": summed dispatch durations}}, {kernel: number of dispatches})"""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"]
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[k].add(r["Dispatch_Id"])
                if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"):
                    agg[k]["_ns_" + r["Counter_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg, {k: len(v) for k, v in cnt.items()}


def fetch_bytes(counter_sum_kib):
    return counter_sum_kib * 1024 * 2


def write_bytes(counter_sum_kib):
    return counter_sum_kib * 1024


def find_kernel(names, prof_name):
    prefix = PROF_TO_KERNEL.get(prof_name)
    if prefix is None:
        return None
    for k in names:
        if k.startswith(prefix):
            return k
    return None


def under_profiler():
    """True when this process already runs under a rocprofiler tool (nesting two of them is not supported)."""
    blob = " ".join(os.environ.get(v, "") for v in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD"))
    return "rocprof" in blob


def measure_traffic(bench_path, prof_name, timeout=150, clock=True):
    """Run the bench command under `rocprofv3 --kernel-trace --pmc <one counter>` (short: 1 warm-up + 1 step, no extra legs),
    once per counter, and return the per-launch traffic of `prof_name`'s kernel, averaged over every launch of that child
    process, next to the algorithmic bytes of the same launches (the child's `roofline.process`).  `clock`: two more passes
    (GRBM_GUI_ACTIVE, SQ_VALU_MFMA_BUSY_CYCLES) give the clock the chip sustained under that kernel and the share of its
    cycles the matrix pipe was busy (scripts/pmc_clock.py's formulas).  Raises on any failure of the traffic passes."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.isfile(exe):
        raise RuntimeError("rocprofv3 not found")
    if under_profiler():
        raise RuntimeError("already running under a rocprofiler tool")
    out = tempfile.mkdtemp(prefix="kocr_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    # the child is a stand-alone single-rank run: drop everything a torchrun parent exported (with
    # TORCHELASTIC_USE_AGENT_STORE set the child would wait for the parent agent's store on its own fresh port)
    for v in list(env):
        if v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                 "ROLE_NAME", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE", "TORCH_NCCL_ASYNC_ERROR_HANDLING") or v.startswith("TORCHELASTIC_"):
            env.pop(v, None)
    child = [sys.executable, os.path.abspath(bench_path), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-alt-mode",
             "--no-extra", "--profile-all", "--no-live-traffic"]
    sums, launches, line = {}, {}, None
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, counter.lower())
            p = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--"] + child,
                               cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, text=True)
            if p.returncode != 0:
                raise RuntimeError(f"rocprofv3 --pmc {counter} exited with {p.returncode}: {p.stdout[-400:]}")
            agg, n = load_counters(d)
            k = find_kernel(agg.keys(), prof_name)
            if k is None or counter not in agg[k]:
                raise RuntimeError(f"no {counter} rows for {prof_name}")
            sums[counter], launches[counter] = agg[k][counter], n[k]
            if counter == "FETCH_SIZE":
                mp = [kk for kk in agg if kk.startswith("maxpool2x2")]
                sums["_mp_fetch"] = agg[mp[0]]["FETCH_SIZE"] if mp else None
                for ln in p.stdout.splitlines():  # the child's FULL record (stderr, merged into stdout here) carries roofline.process
                    if ln.startswith("[bench full] "):
                        line = json.loads(ln[len("[bench full] "):])
                    elif ln.startswith('{"metric"') and line is None:
                        line = json.loads(ln)
            else:
                mp = [kk for kk in agg if kk.startswith("maxpool2x2")]
                sums["_mp_write"] = agg[mp[0]]["WRITE_SIZE"] if mp else None
        if clock:
            try:
                cyc = {}
                for counter in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"):
                    d = os.path.join(out, counter.lower())
                    p = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--"] + child,
                                       cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, text=True)
                    agg, n = load_counters(d)
                    k = find_kernel(agg.keys(), prof_name)
                    if p.returncode == 0 and k is not None and counter in agg[k]:
                        cyc[counter] = (agg[k][counter], agg[k]["_ns_" + counter])
                if len(cyc) == 2:
                    ghz = cyc["GRBM_GUI_ACTIVE"][0] / 8 / cyc["GRBM_GUI_ACTIVE"][1]  # summed over the 8 XCDs
                    sums["_clock_ghz"] = ghz
                    # SQ_VALU_MFMA_BUSY_CYCLES sums over 256 CUs x 4 SIMDs
                    sums["_mfma_busy"] = cyc["SQ_VALU_MFMA_BUSY_CYCLES"][0] / 1024 / (cyc["SQ_VALU_MFMA_BUSY_CYCLES"][1] * ghz)
            except Exception:  # noqa: BLE001 -- the clock passes are a bonus
                pass
    finally:
        shutil.rmtree(out, ignore_errors=True)
    res = {"fetch_bytes_per_launch": fetch_bytes(sums["FETCH_SIZE"]) / launches["FETCH_SIZE"],
           "write_bytes_per_launch": write_bytes(sums["WRITE_SIZE"]) / launches["WRITE_SIZE"],
           "launches": launches["FETCH_SIZE"]}
    res["hbm_bytes_per_launch"] = res["fetch_bytes_per_launch"] + res["write_bytes_per_launch"]
    if "_clock_ghz" in sums:
        res["clock_ghz"] = sums["_clock_ghz"]
        res["mfma_pipe_busy"] = sums["_mfma_busy"]
    if sums.get("_mp_fetch") and sums.get("_mp_write"):
        res["calibration_maxpool2x2_read_over_write"] = fetch_bytes(sums["_mp_fetch"]) / write_bytes(sums["_mp_write"])
    pr = (line or {}).get("roofline", {}).get("process")
    if pr and (line or {}).get("roofline", {}).get("kernel") == prof_name:
        res["algorithmic_bytes_per_launch_same_process"] = pr["algorithmic_bytes_per_launch"]
        res["launches_hip_events"] = pr["launches"]
        res["traffic_over_algorithmic"] = res["hbm_bytes_per_launch"] / pr["algorithmic_bytes_per_launch"]
    return res

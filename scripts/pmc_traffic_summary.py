"""Per-kernel, per-launch HBM traffic + MFMA utilisation from the PMC passes of scripts/pmc_bench.sh.

Corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950
FETCH_SIZE counts 128-B requests as 64 B, i.e. reports exactly half of the bytes of a wide
(16 B / lane) coalesced read stream -> doubled here.  The factor is calibrated on this run's own
maxpool2x2 launches, a pure float4 streaming kernel whose byte count is known exactly
(reads 4x what it writes).  WRITE_SIZE is used as reported (calibrated the same way)."""
import collections, csv, glob, json, os, sys

root, out_json = sys.argv[1], sys.argv[2]

def load(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for f in glob.glob(os.path.join(root, sub, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE"):
                agg[k]["_ns_" + r["Counter_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg, {k: len(v) for k, v in cnt.items()}

fetch, nf = load("fetch")
write, nw = load("write")
mfma, nm = load("mfma")
res = {"units": "bytes per launch (average over all launches of the kernel in the bench process)",
       "corrections": "FETCH_SIZE[KiB]*1024*2 (gfx950 half-count of wide coalesced reads), WRITE_SIZE[KiB]*1024",
       "kernels": {}}
mp = [k for k in fetch if k.startswith("maxpool2x2")]
if mp:
    k = mp[0]
    rd = fetch[k]["FETCH_SIZE"] * 1024 * 2
    wr = write[k]["WRITE_SIZE"] * 1024 if k in write else float("nan")
    res["calibration_maxpool2x2"] = {"read_bytes_corrected": rd, "write_bytes": wr, "read_over_write": rd / wr if wr else None,
                                     "expected_read_over_write": 4.0}
for k in sorted(fetch, key=lambda k: -fetch[k]["FETCH_SIZE"]):
    row = {"launches": nf[k], "fetch_bytes_per_launch": fetch[k]["FETCH_SIZE"] * 1024 * 2 / nf[k]}
    if k in write:
        row["write_bytes_per_launch"] = write[k]["WRITE_SIZE"] * 1024 / nw[k]
        row["hbm_bytes_per_launch"] = row["fetch_bytes_per_launch"] + row["write_bytes_per_launch"]
    if k in mfma and mfma[k].get("GRBM_GUI_ACTIVE"):
        cyc = mfma[k]["GRBM_GUI_ACTIVE"] / 8  # summed over the 8 XCDs
        row["mfma_busy_frac"] = mfma[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)  # 256 CUs x 4 SIMDs
        row["clock_ghz"] = cyc / mfma[k]["_ns_GRBM_GUI_ACTIVE"]
        row["waves_per_simd"] = mfma[k]["SQ_WAVE_CYCLES"] * 4 / (cyc * 1024)
    res["kernels"][k] = row
# libkocr's profiler row names (what bench.py's roofline object is keyed on) -> rocprof kernel names
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "keras_ocr_amd"))
from pmc import PROF_TO_KERNEL as PROF  # noqa: E402  (stdlib-only module; the package itself is not imported)
res["by_prof_name"] = {}
for pn, prefix in PROF.items():
    for k, row in res["kernels"].items():
        if k.startswith(prefix):
            res["by_prof_name"][pn] = dict(row, kernel=k)
            break
# algorithmic bytes per launch of the SAME process (bench.py --profile-all line of the FETCH_SIZE run)
bj = os.path.join(root, "bench_under_pmc.json")
if os.path.isfile(bj) and os.path.getsize(bj):
    b = json.load(open(bj))
    pr = b.get("roofline", {}).get("process")
    nm = b.get("roofline", {}).get("kernel")
    if pr and nm in res["by_prof_name"]:
        row = res["by_prof_name"][nm]
        row["algorithmic_bytes_per_launch_same_process"] = pr["algorithmic_bytes_per_launch"]
        row["launches_hip_events"] = pr["launches"]
        row["traffic_over_algorithmic"] = row["hbm_bytes_per_launch"] / pr["algorithmic_bytes_per_launch"]
json.dump(res, open(out_json, "w"), indent=1)
for k, row in list(res["kernels"].items())[:8]:
    print(k[:70], {a: (round(b, 3) if isinstance(b, float) and b < 100 else b) for a, b in row.items()})
print(res.get("calibration_maxpool2x2"))

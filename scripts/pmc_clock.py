"""Per-kernel clock and matrix-pipe occupancy from scripts/pmc_craft.sh output (p1 = SQ counters, p2 = GRBM_GUI_ACTIVE)."""
import csv, sys, collections
root = sys.argv[1]
def load(sub, names):
    kt = {}
    for r in csv.DictReader(open(f"{root}/{sub}/{sub}_kernel_trace.csv")):
        kt[r["Dispatch_Id"]] = (r["Kernel_Name"][:48], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = set()
    for r in csv.DictReader(open(f"{root}/{sub}/{sub}_counter_collection.csv")):
        n, d = kt[r["Dispatch_Id"]]
        if r["Counter_Name"] in names:
            out[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"]) not in seen:
            seen.add(r["Dispatch_Id"]); out[n]["ns"] += d
    return out
p2 = load("p2", {"GRBM_GUI_ACTIVE"})
p1 = load("p1", {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"})
for n in sorted(p2, key=lambda k: -p2[k]["ns"])[:6]:
    cyc = p2[n]["GRBM_GUI_ACTIVE"] / 8          # summed over 8 XCDs
    ghz = cyc / p2[n]["ns"]
    busy = p1[n]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (p1[n]["ns"] * ghz) if n in p1 else float("nan")
    print(f"{n:50s} {p2[n]['ns']/1e6:8.3f} ms  {ghz:5.2f} GHz  matrix pipe busy {100*busy:5.1f} %")

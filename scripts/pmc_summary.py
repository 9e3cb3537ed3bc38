"""Aggregate rocprofv3 counter_collection CSVs per kernel name (sum over dispatches)."""
import csv, sys, collections, glob, os
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in sorted(glob.glob(os.path.join(root, "p*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(f, k)].add(r["Dispatch_Id"])
names = sorted({c for k in agg for c in agg[k]})
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:
    print(k)
    for c in names:
        if c in v:
            print(f"    {c:28s} {v[c]:.4g}")
    if "SQ_BUSY_CYCLES" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        print(f"    mfma_busy/busy(x4 simd?)     {v['SQ_VALU_MFMA_BUSY_CYCLES']/v['SQ_BUSY_CYCLES']:.3f}")
    if "SQ_WAVE_CYCLES" in v:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in v: print(f"    {c}/WAVE_CYCLES   {v[c]/v['SQ_WAVE_CYCLES']:.3f}")

#!/bin/bash
# Matrix-pipe utilisation / clock / wait PMC pass over the CRAFT-only probe (kernel-trace only, one rocprofv3 run per
# counter set, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  usage: pmc_layers.sh <tag> [N H W]
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcl_$1
mkdir -p $OUT
CMD="python $REPO/scripts/perf_craft.py ${2:-8} ${3:-1536} ${4:-1536} 1"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -o p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p2 -o p2 -- $CMD > $OUT/p2.log 2>&1
python3 - $OUT <<'PY'
import csv, sys, glob, os, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); ns = collections.defaultdict(float); nd = collections.defaultdict(set)
for f in glob.glob(os.path.join(root, "p*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            ns[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); nd[k].add(r["Dispatch_Id"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    if not v.get("GRBM_GUI_ACTIVE"): continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    line = f"{k:50s} n={len(nd[k]):3d} ms/launch={ns[k]/1e6/len(nd[k]):7.3f} clk={cyc/ns[k]:.2f}GHz mfma_busy={v['SQ_VALU_MFMA_BUSY_CYCLES']/(cyc*1024):.3f} waves/simd={v['SQ_WAVE_CYCLES']*4/(cyc*1024):.2f}"
    if v.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in v:
        wc = v["SQ_WAVE_CYCLES"]
        line += f" | wait_inst={v['SQ_WAIT_INST_ANY']/wc:.2f} act_valu={v.get('SQ_ACTIVE_INST_VALU',0)/wc:.2f} wait_lds={v.get('SQ_WAIT_INST_LDS',0)/wc:.2f} valu/mfma={v.get('SQ_INSTS_VALU',0)/max(v.get('SQ_INSTS_MFMA',1),1):.2f}"
    print(line)
PY

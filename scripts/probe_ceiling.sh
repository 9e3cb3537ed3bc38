#!/bin/bash
# MFMA ceiling probe (scripts/probes/probe_mfma_ceiling.hip): plain timing run, then one rocprofv3 --pmc pass (kernel-trace
# only) for the matrix-pipe busy counter and the clock of every variant.  usage: probe_ceiling.sh <tag>
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/probe_$1
mkdir -p $OUT
BIN=$REPO/scripts/probes/probe_mfma_ceiling
[ -x $BIN ] || hipcc -O3 --offload-arch=gfx950 -o $BIN $REPO/scripts/probes/probe_mfma_ceiling.hip
$BIN 6000 > $OUT/timing.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -o p1 -- $BIN 3000 > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p2 -o p2 -- $BIN 3000 > $OUT/p2.log 2>&1
python3 - $OUT > $OUT/pmc.txt <<'PY'
import csv, sys, glob, os
root = sys.argv[1]
labels = [l[:60].strip() + " [" + l[60:72].strip() + "]" for l in open(os.path.join(root, "timing.txt")) if "TF/s issued" in l]
def chunks(sub):
    disp = {}
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_probe" not in r["Kernel_Name"]: continue
            d = disp.setdefault(int(r["Dispatch_Id"]), {"ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "kn": r["Kernel_Name"]})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(disp)
    out = []
    for i in range(0, len(ids) - 3, 4):   # a run<> call = 1 warm-up + 3 timed launches
        out.append(min((disp[j] for j in ids[i + 1:i + 4]), key=lambda d: d["ns"]))
    return out
c1, c2 = chunks("p1"), chunks("p2")
for i, v in enumerate(c1):
    if not v.get("GRBM_GUI_ACTIVE"): continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    lab = labels[i] if i < len(labels) else v["kn"][:60]
    line = f"{lab:74s} ms={v['ns']/1e6:7.3f} clk={cyc/v['ns']:.2f}GHz mfma_busy={v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(cyc*1024):.3f} waves/simd={v.get('SQ_WAVE_CYCLES',0)*4/(cyc*1024):.2f}"
    if i < len(c2) and c2[i].get("SQ_WAVE_CYCLES", 0) == 0 and "SQ_WAIT_INST_ANY" in c2[i]:
        w = c2[i]; wc = v.get("SQ_WAVE_CYCLES") * (w["ns"] / v["ns"]) if v.get("SQ_WAVE_CYCLES") else 0
        if wc:
            line += f" | wait_inst={w['SQ_WAIT_INST_ANY']/wc:.2f} wait_any={w.get('SQ_WAIT_ANY',0)/wc:.2f} act_valu={w.get('SQ_ACTIVE_INST_VALU',0)/wc:.2f} valu/mfma={w.get('SQ_INSTS_VALU',0)/max(w.get('SQ_INSTS_MFMA',1),1):.2f}"
    print(line)
PY
cat $OUT/timing.txt; echo; cat $OUT/pmc.txt

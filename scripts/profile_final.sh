#!/bin/bash
# Round artefacts: (1) the default bench line (live PMC traffic passes included), (2) rocprofv3 --kernel-trace --stats of the
# SAME command with --profile-all (+ its bench line), (3) per-layer tables of the detector and the recogniser, (4) matrix-pipe
# busy / clock PMC passes over a CRAFT-only probe (separate runs, kernel-trace only).   usage: profile_final.sh <tag>
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final_$1
mkdir -p $OUT
python $REPO/bench.py --steps 10 --warmup 3 2> $OUT/bench.err | tail -1 > $OUT/bench.json
cp $REPO/gpurun_out/bench_full.json $OUT/bench_full.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py --profile-all --no-live-traffic > $OUT/trace.log 2>&1
grep '^\[bench full\] ' $OUT/trace.log | tail -1 | sed 's/^\[bench full\] //' > $OUT/bench_under_rocprof.json
KOCR_PROF_LAYERS=1 python $REPO/scripts/perf_craft.py 8 1536 1536 5 > $OUT/craft_layers.txt 2>&1
KOCR_PROF_LAYERS=1 python $REPO/scripts/perf_crnn.py 512 > $OUT/crnn_layers.txt 2>&1
# the detector on a page size no level of which tiles
KOCR_PROF_LAYERS=1 python $REPO/scripts/perf_craft.py 8 1500 2000 5 > $OUT/craft_layers_1500x2000.txt 2>&1
CMD="python $REPO/scripts/perf_craft.py 8 1536 1536 1"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p1 -o p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/p4 -o p4 -- $CMD > $OUT/p4.log 2>&1
cd $REPO
python scripts/pmc_clock.py $OUT > $OUT/pmc_mfma_clock.txt 2>&1
python - $OUT >> $OUT/pmc_mfma_clock.txt <<'PY'
import csv, collections, sys
root = sys.argv[1]
print("\nper-kernel counter ratios (p1: per SQ_WAVE_CYCLES; p4: per MFMA instruction; SQ_INSTS_VALU counts the MFMAs too)")
for sub in ("p1", "p4"):
    kt = {}
    for r in csv.DictReader(open(f"{root}/{sub}/{sub}_kernel_trace.csv")):
        kt[r["Dispatch_Id"]] = r["Kernel_Name"][:52]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f"{root}/{sub}/{sub}_counter_collection.csv")):
        agg[kt[r["Dispatch_Id"]]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(agg.items()):
        if "conv_w43" in k or "conv_hs" in k or "conv_ds" in k or "conv_first" in k:
            den = max(v.get("SQ_WAVE_CYCLES", v.get("SQ_INSTS_MFMA", 1)), 1)
            print(sub, k, {a: round(b / den, 3) for a, b in sorted(v.items())})
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p4
# round 6: HBM traffic of every dispatch of one forward (FETCH_SIZE / WRITE_SIZE, separate passes, kernel-trace only)
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
python $REPO/scripts/pmc_traffic_dispatch.py $OUT > $OUT/traffic_per_dispatch.txt 2>&1
rm -rf $OUT/fetch $OUT/write $OUT/fetch.log $OUT/write.log
cd $REPO
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
rm -rf $OUT/trace
ls $OUT; head -c 600 $OUT/bench.json

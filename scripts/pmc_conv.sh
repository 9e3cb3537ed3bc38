#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one convolution shape (scripts/perf_conv.py index), separate rocprofv3 runs.
# usage: pmc_conv.sh <tag> <shape index> [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; IDX=$2; shift 2
OUT=$REPO/gpurun_out/pmcc_$TAG
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  env "$@" rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- python $REPO/scripts/perf_conv.py $IDX > $OUT/$c.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(root, c, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("void conv_"):
                agg[r["Kernel_Name"][:44]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(r["Kernel_Name"][:44], c)].add(r["Dispatch_Id"])
for k, v in agg.items():
    nf = len(n[(k, "FETCH_SIZE")]) or 1
    print(f"{k:46s} launches={nf} fetch/launch={v['FETCH_SIZE']*1024*2/nf/1e9:.3f} GB write/launch={v['WRITE_SIZE']*1024/nf/1e9:.3f} GB")
PY

#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes only (quick traffic check of the bench command)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_bench_$1
mkdir -p $OUT
CMD="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-mode"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1

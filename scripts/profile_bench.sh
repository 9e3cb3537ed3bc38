#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command (run on the GPU box through gpurun).
set -e
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1 || true
ls -R $OUT | head -30
find $OUT -name "*kernel_stats*" | head

#!/bin/bash
# HBM-traffic PMC passes over the bench command (separate rocprofv3 runs, kernel-trace only, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2).
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_bench_$1
mkdir -p $OUT
CMD="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-mode"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/mfma -o mfma -- $CMD > $OUT/mfma.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
ls $OUT/*/

#!/bin/bash
# Round artefacts: (1) the default bench line, (2) rocprofv3 --kernel-trace --stats of the SAME default
# command (+ its bench line), (3) HBM-traffic / matrix-pipe PMC passes (separate runs, kernel-trace only).
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final_$1
mkdir -p $OUT
python $REPO/bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py --profile-all --no-live-traffic > $OUT/trace.log 2>&1
grep '^{"metric"' $OUT/trace.log | tail -1 > $OUT/bench_under_rocprof.json
CMD="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-mode --no-extra --profile-all --no-live-traffic"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
grep '^{"metric"' $OUT/fetch.log | tail -1 > $OUT/bench_under_pmc.json
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/mfma -o mfma -- $CMD > $OUT/mfma.log 2>&1
ls $OUT

#!/bin/bash
# PMC passes (each in its own rocprofv3 run, kernel-trace only) over a short CRAFT-only probe.
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$1
mkdir -p $OUT
CMD="python $REPO/scripts/perf_craft.py 4 768 768 1"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p1 -o p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p3 -o p3 -- $CMD > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/p4 -o p4 -- $CMD > $OUT/p4.log 2>&1
find $OUT -name "*.csv" | head -20
tail -3 $OUT/p1.log

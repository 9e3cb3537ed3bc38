#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c4; mkdir -p $O
timeout 600 python scripts/debug_heavy.py > $O/debug_heavy.txt 2>&1
timeout 900 python -m pytest tests/test_range_gpu.py -q -m gpu -s > $O/t_range.log 2>&1; echo "range rc $?" >> $O/summary.txt
cat $O/debug_heavy.txt; tail -n 8 $O/t_range.log; grep -E "^(lognormal|one_outlier|99pct|ordinary|heavy)" $O/t_range.log

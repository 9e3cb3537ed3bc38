#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c3; mkdir -p $O
timeout 900 python -m pytest tests/test_range_gpu.py -q -m gpu -s > $O/t_range.log 2>&1; echo "range rc $?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_cells_gpu.py -q -m gpu -k "cells or fp32_class" > $O/t_conv.log 2>&1; echo "conv rc $?" >> $O/summary.txt
timeout 1500 python -m pytest tests/test_fallback_paths_gpu.py -q -m gpu > $O/t_fallback.log 2>&1; echo "fallback rc $?" >> $O/summary.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/summary.txt
for f in $O/t_range.log $O/t_conv.log $O/t_fallback.log; do tail -n 6 $f; done; cat $O/summary.txt; grep -E "^(lognormal|one_outlier|99pct|ordinary|heavy)" $O/t_range.log; head -c 1500 $O/bench.json

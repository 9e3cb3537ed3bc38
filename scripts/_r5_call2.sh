#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c2; mkdir -p $O
timeout 900 python -m pytest tests/test_cells_gpu.py tests/test_conv_gpu.py -q -m gpu -k "cells or fp32_class" > $O/t_conv.log 2>&1; echo "conv rc $?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_crnn_gpu.py tests/test_craft_gpu.py -q -m gpu > $O/t_nets.log 2>&1; echo "nets rc $?" >> $O/summary.txt
timeout 1200 python -m pytest tests/test_pipeline_gpu.py tests/test_baseline_sizes_gpu.py -q -m gpu > $O/t_pipe.log 2>&1; echo "pipe rc $?" >> $O/summary.txt
KOCR_PROF_LAYERS=1 timeout 300 python scripts/perf_crnn.py 512 > $O/crnn_cells.txt 2>&1
KOCR_PROF_LAYERS=1 timeout 300 python scripts/perf_crnn.py 705 > $O/crnn_cells705.txt 2>&1
KOCR_PROF_LAYERS=1 timeout 300 python scripts/perf_craft.py 8 1500 2000 3 > $O/craft_ragged.txt 2>&1
for f in $O/t_conv.log $O/t_nets.log $O/t_pipe.log; do tail -n 5 $f; done; cat $O/summary.txt; head -n 12 $O/crnn_cells.txt

#!/bin/bash
# round 5, call 1: ragged + cell-grid kernels: unit tests, the recogniser, CRAFT at odd sizes, per-layer tables
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c1; mkdir -p $O
timeout 600 python -m pytest tests/test_cells_gpu.py tests/test_conv_gpu.py -x -q -m gpu -k "cells or fp32_class" > $O/t_conv.log 2>&1; echo "conv rc $?" >> $O/summary.txt
timeout 600 python -m pytest tests/test_crnn_gpu.py tests/test_craft_gpu.py -x -q -m gpu > $O/t_nets.log 2>&1; echo "nets rc $?" >> $O/summary.txt
timeout 600 python -m pytest tests/test_pipeline_gpu.py tests/test_baseline_sizes_gpu.py -x -q -m gpu -k "order_independence or cfg3 or end_to_end" > $O/t_pipe.log 2>&1; echo "pipe rc $?" >> $O/summary.txt
KOCR_PROF_LAYERS=1 timeout 300 python scripts/perf_crnn.py 512 > $O/crnn_cells.txt 2>&1
KOCR_CELLS=0 KOCR_PROF_LAYERS=1 timeout 300 python scripts/perf_crnn.py 512 > $O/crnn_dense.txt 2>&1
KOCR_PROF_LAYERS=1 timeout 300 python scripts/perf_crnn.py 705 > $O/crnn_cells705.txt 2>&1
KOCR_PROF_LAYERS=1 timeout 300 python scripts/perf_craft.py 8 1500 2000 3 > $O/craft_ragged.txt 2>&1
KOCR_W43RAG=0 KOCR_PROF_LAYERS=1 timeout 300 python scripts/perf_craft.py 8 1500 2000 3 > $O/craft_ragged_off.txt 2>&1
KOCR_PROF_LAYERS=1 timeout 300 python scripts/perf_craft.py 8 1536 1536 3 > $O/craft_1536.txt 2>&1
tail -3 $O/t_conv.log $O/t_nets.log $O/t_pipe.log; cat $O/summary.txt; head -12 $O/crnn_cells.txt; head -12 $O/crnn_dense.txt; head -3 $O/craft_ragged.txt $O/craft_ragged_off.txt $O/craft_1536.txt

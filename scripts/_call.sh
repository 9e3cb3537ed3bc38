mkdir -p gpurun_out/c17
( KOCR_W43B=1 timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_craft_gpu.py -m gpu -q -x ) > gpurun_out/c17/conv.log 2>&1
tail -5 gpurun_out/c17/conv.log
for b in 0 1; do echo "== W43B $b"; KOCR_W43B=$b timeout 120 python scripts/perf_conv.py 8 9 2>&1 | grep conv; done
( KOCR_W43B=1 KOCR_PROF_LAYERS=1 timeout 300 python scripts/perf_craft.py 8 1536 1536 ) > gpurun_out/c17/craft_b.log 2>&1
head -14 gpurun_out/c17/craft_b.log

mkdir -p gpurun_out/c14
( timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_craft_gpu.py -m gpu -q -s ) > gpurun_out/c14/conv.log 2>&1
grep -E "passed|failed|Error|error" gpurun_out/c14/conv.log | tail -5
( KOCR_PROF_LAYERS=1 timeout 300 python scripts/perf_craft.py 8 1536 1536 ) > gpurun_out/c14/craft.log 2>&1
head -3 gpurun_out/c14/craft.log; grep -E "slice5|upconv1.conv.3" gpurun_out/c14/craft.log

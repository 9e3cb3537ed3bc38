mkdir -p gpurun_out/c3
( time python -m pytest tests/test_baseline_sizes_gpu.py tests/test_pipeline_gpu.py -m gpu -q -s -x ) > gpurun_out/c3/new_tests.log 2>&1
( time python bench.py ) > gpurun_out/c3/bench.json 2> gpurun_out/c3/bench.err
grep -E "cfg|e2e|mixed|passed|failed|Error|error" gpurun_out/c3/new_tests.log | head -30; grep calibration gpurun_out/c3/bench.err; wc -l gpurun_out/c3/bench.json

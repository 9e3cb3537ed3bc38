mkdir -p gpurun_out/c13
( time python -m pytest tests/test_craft_gpu.py tests/test_pipeline_gpu.py tests/test_baseline_sizes_gpu.py tests/test_split_modes_gpu.py -m gpu -q -x ) > gpurun_out/c13/gpu.log 2>&1
( time python bench.py --no-cpu-baseline ) > gpurun_out/c13/bench.json 2> gpurun_out/c13/bench.err
tail -4 gpurun_out/c13/gpu.log; cut -c1-300 gpurun_out/c13/bench.json; tail -3 gpurun_out/c13/bench.err

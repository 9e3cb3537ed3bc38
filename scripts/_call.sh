mkdir -p gpurun_out/c4
( time python -m pytest tests -m gpu -q -x ) > gpurun_out/c4/gpu.log 2>&1
( time python bench.py ) > gpurun_out/c4/bench.json 2> gpurun_out/c4/bench.err
tail -15 gpurun_out/c4/gpu.log; grep calibration gpurun_out/c4/bench.err | tail -4; wc -l gpurun_out/c4/bench.json

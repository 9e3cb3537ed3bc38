mkdir -p gpurun_out/c1
ls -la keras-ocr_amd | head -3 > gpurun_out/c1/symlink.txt 2>&1
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/c1/gpu_bf16.log 2>&1
( time KOCR_SPLIT=f16 python -m pytest tests -m gpu -q ) > gpurun_out/c1/gpu_f16.log 2>&1
python bench.py > gpurun_out/c1/bench.json 2> gpurun_out/c1/bench.err
tail -3 gpurun_out/c1/gpu_bf16.log gpurun_out/c1/gpu_f16.log; cat gpurun_out/c1/bench.json | cut -c1-600

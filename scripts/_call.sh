mkdir -p gpurun_out/c16
( time KOCR_SPLIT=f16 python -m pytest tests -m gpu -q -x ) > gpurun_out/c16/gpu_f16.log 2>&1
( time python bench.py --no-cpu-baseline --no-extra ) > gpurun_out/c16/bench.json 2> gpurun_out/c16/bench.err
tail -4 gpurun_out/c16/gpu_f16.log; python -c "
import json;d=json.loads(open('gpurun_out/c16/bench.json').readline());print(d['value'],d['alt_split_mode'])"

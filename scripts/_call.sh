mkdir -p gpurun_out/c2
( time python bench.py ) > gpurun_out/c2/bench.json 2> gpurun_out/c2/bench.err
python bench.py --gpus 2 > gpurun_out/c2/bench2.out 2>&1; echo "rc=$?" >> gpurun_out/c2/bench2.out
tail -5 gpurun_out/c2/bench.err; tail -3 gpurun_out/c2/bench2.out; cut -c1-300 gpurun_out/c2/bench.json

mkdir -p gpurun_out/c18
( KOCR_W43_MIN_COUT=64 KOCR_PROF_LAYERS=1 timeout 300 python scripts/perf_craft.py 8 1536 1536 ) > gpurun_out/c18/craft64.log 2>&1
head -3 gpurun_out/c18/craft64.log; grep -E "slice1.3|upconv3.conv.3" gpurun_out/c18/craft64.log

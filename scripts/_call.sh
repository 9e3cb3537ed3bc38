mkdir -p gpurun_out/c19
( KOCR_WS_MIN_COUT=32 KOCR_PROF_LAYERS=1 timeout 300 python scripts/perf_craft.py 8 1536 1536 ) > gpurun_out/c19/craft32.log 2>&1
head -3 gpurun_out/c19/craft32.log; grep -E "upconv4.conv.3|conv_cls" gpurun_out/c19/craft32.log
( KOCR_WS_MIN_COUT=16 KOCR_PROF_LAYERS=1 timeout 300 python scripts/perf_craft.py 8 1536 1536 ) > gpurun_out/c19/craft16.log 2>&1
head -3 gpurun_out/c19/craft16.log; grep -E "upconv4.conv.3|conv_cls" gpurun_out/c19/craft16.log

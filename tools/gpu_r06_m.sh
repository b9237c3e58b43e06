#!/bin/bash
# round 6, last session: native score sort (sd_sort_scores_desc_device) and the rounds' N-sized set-up in front of the neighbour lists
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06m; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 900 python -m pytest tests/test_gpu_glue.py tests/test_gpu_parity2d.py tests/test_gpu_fullsize_parity.py tests/test_gpu_fullsize.py tests/test_gpu_heads.py -m gpu -q -x 2>&1 | tail -8 ) > $O/tests.log 2>&1
tail -4 $O/tests.log
timeout 200 python tools/time_predict_sections.py > $O/sections.log 2>&1; head -24 $O/sections.log
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err ) > $O/bench_time.log 2>&1
cut -c1-300 $O/bench.json; tail -3 $O/bench_time.log

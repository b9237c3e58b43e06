#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
SD_PAIR_LANES=64 timeout 200 python tools/time_nms2d_bench.py 4 > $O/s6_nms2d_64.log 2>&1
SD_PAIR_LANES=32 timeout 200 python tools/time_nms2d_bench.py 4 > $O/s6_nms2d_32.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_parity2d.py tests/test_gpu_fullsize_parity.py tests/test_gpu_bigparity.py -m gpu -q -x -k "not 3d" 2>&1 | tail -12 ) > $O/s6_tests.log 2>&1
tail -5 $O/s6_nms2d_64.log; tail -5 $O/s6_nms2d_32.log; tail -8 $O/s6_tests.log

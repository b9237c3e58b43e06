#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
SD_PAIR_LANES=6464 timeout 200 python tools/time_nms2d_bench.py 4 > $O/s7_nms2d_32bit.log 2>&1
SD_PAIR_LANES=64 timeout 200 python tools/time_nms2d_bench.py 4 > $O/s7_nms2d_16bit.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_parity2d.py tests/test_gpu_fullsize_parity.py tests/test_gpu_beam_prep.py -m gpu -q -x -k "not 3d" 2>&1 | tail -12 ) > $O/s7_tests.log 2>&1
tail -4 $O/s7_nms2d_32bit.log; tail -4 $O/s7_nms2d_16bit.log; tail -6 $O/s7_tests.log

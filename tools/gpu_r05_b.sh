#!/bin/bash
# round 5: short pair lists spread over all waves (beam kernel, general path): parity + NMS timing on the bench set
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 400 python -m pytest tests/test_gpu_parity2d.py tests/test_gpu_fullsize_parity.py tests/test_gpu_beam_prep.py -m gpu -q -x 2>&1 | tail -4 ) > $O/spread_tests.log 2>&1
SD_TRACE=1 timeout 120 python tools/time_nms2d_bench.py 4 2>&1 | grep -v amdgpu > $O/spread_nms2d.txt
timeout 120 python tools/check_defer.py 2>&1 | grep -v amdgpu > $O/spread_check_defer.txt
cat $O/spread_tests.log | tail -5; grep "^rep\|tail batch\|round 1" $O/spread_nms2d.txt | tail -8; tail -12 $O/spread_check_defer.txt

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
for v in 1 2 3; do SD_SPLIT_EXACT=$v timeout 200 python tools/time_nms3d_bench.py 4 > $O/s9_nms3d_split$v.log 2>&1; done
( time timeout 900 python -m pytest tests/test_gpu_parity3d.py tests/test_gpu_fullsize_parity.py -m gpu -q -x -k "3d or 3D" 2>&1 | tail -8 ) > $O/s9_tests.log 2>&1
for v in 1 2 3; do tail -3 $O/s9_nms3d_split$v.log | cut -c1-250; done; tail -5 $O/s9_tests.log

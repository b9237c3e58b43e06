#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
timeout 100 python tools/check_defer.py > $O/s21_check_defer.log 2>&1; grep -v "^/opt" $O/s21_check_defer.log | tail -10 | cut -c1-260
export SD_TEST_OPTIONS="nms2d_defer_undecided=2"
timeout 60 python -m pytest tests/test_gpu_parity2d.py -m gpu -q -x 2>&1 | tail -2 | cut -c1-200
timeout 60 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "2d or 2D" 2>&1 | tail -2 | cut -c1-200

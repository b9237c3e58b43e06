#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 600 python -m pytest tests/test_gpu_unet_parity.py -m gpu -q -x -s -k "base48 or depth4" 2>&1 | tail -15 ) > $O/s13_tests.log 2>&1
grep -v "^$" $O/s13_tests.log | tail -12

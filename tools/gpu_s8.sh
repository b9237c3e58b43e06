#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 600 python -m pytest tests/test_gpu_multi.py -m gpu -q -x 2>&1 | tail -15 ) > $O/s8_multi.log 2>&1
tail -15 $O/s8_multi.log

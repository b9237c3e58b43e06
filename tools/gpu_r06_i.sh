#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06i; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 900 python -m pytest tests/test_gpu_parity3d.py -m gpu -q -x -s -k "cartesian" 2>&1 | grep -a "cartesian(\|passed\|failed\|Error\|assert" | tail -12 ) > $O/tests.log 2>&1
cat $O/tests.log

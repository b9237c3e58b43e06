#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06i; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 1500 python -m pytest tests/test_gpu_parity3d.py -m gpu -q -rf -k "flag_combinations" 2>&1 | tail -12 ) > $O/tests.log 2>&1
cat $O/tests.log

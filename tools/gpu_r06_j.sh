#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06j; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 600 python -m pytest tests/test_gpu_parity2d.py tests/test_gpu_parity3d.py -m gpu -q -rf --durations=6 -k "extreme or threshold_edges" > $O/tests4_full.log 2>&1 )
grep -a -n "FAILED\|passed\|failed\|^E  \|Kernel Name\|s call\|aborting" $O/tests4_full.log | cut -c1-400 | head -40

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06j; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 300 python -m pytest tests/test_gpu_parity3d.py -m gpu -q -rf --durations=4 -k "label_extreme" > $O/tests5_full.log 2>&1 )
grep -a -n "FAILED\|passed\|failed\|^E  \|Kernel Name\|s call\|aborting" $O/tests5_full.log | cut -c1-400 | head -40

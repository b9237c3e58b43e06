#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06j; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 400 python -m pytest tests/test_gpu_fullsize_parity.py -m gpu -q -rf --durations=4 -k "3d_end_to_end" > $O/tests6_full.log 2>&1 )
grep -a -n "FAILED\|passed\|failed\|^E  \|Kernel Name\|s call\|aborting" $O/tests6_full.log | cut -c1-400 | head -40

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06j; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 600 python -m pytest tests/test_gpu_parity2d.py -m gpu -q -rf --durations=8 -k "many_rays" > $O/tests3_full.log 2>&1 )
grep -a -n "FAILED\|passed\|failed\|^E  \|Kernel Name\|s call\|aborting" $O/tests3_full.log | cut -c1-300 | head -30

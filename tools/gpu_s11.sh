#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
timeout 600 python tools/diag_stage3x.py > $O/s11_diag.log 2>&1
grep -v "^/opt" $O/s11_diag.log | tail -30

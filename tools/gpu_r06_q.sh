#!/bin/bash
# round 6: results under contention (P processes sharing the device)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R; ulimit -c 0
timeout 300 python tools/contention_check.py 1 4 2d > $O/contention_2d_p1.txt 2>&1; tail -4 $O/contention_2d_p1.txt
timeout 600 python tools/contention_check.py 8 12 2d > $O/contention_2d_p8.txt 2>&1; tail -12 $O/contention_2d_p8.txt
timeout 600 python tools/contention_check.py 6 6 3d > $O/contention_3d_p6.txt 2>&1; tail -12 $O/contention_3d_p6.txt

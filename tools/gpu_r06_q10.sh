#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R; ulimit -c 0
SD_OPTS="nms2d_strict=1" timeout 800 python tools/contention_check.py 8 100 2d > "$O/contention10_2d_p8_strict.txt" 2>&1; grep -v "^pid" "$O/contention10_2d_p8_strict.txt" | tail -6 | cut -c1-260
timeout 600 python tools/contention_check.py 8 20 3d > "$O/contention10_3d_p8.txt" 2>&1; grep -v "^pid" "$O/contention10_3d_p8.txt" | tail -6 | cut -c1-260

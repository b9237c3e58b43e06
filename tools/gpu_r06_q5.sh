#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R; ulimit -c 0
timeout 500 python tools/contention_band.py 8 40 > $O/contention_band_p8.txt 2>&1; tail -14 $O/contention_band_p8.txt | cut -c1-500
SD_OPTS="nms2d_strict=1" timeout 500 python tools/contention_check.py 8 20 2d > "$O/contention5_2d_p8_strict.txt" 2>&1; grep -v "^pid" "$O/contention5_2d_p8_strict.txt" | tail -8 | cut -c1-330

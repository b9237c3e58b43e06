#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R; ulimit -c 0
for cfg in "" "nms2d_defer_undecided=0"; do
  SD_OPTS="$cfg" timeout 500 python tools/contention_check.py 8 40 2d > "$O/contention4_2d_p8_$cfg.txt" 2>&1; grep -v "^pid" "$O/contention4_2d_p8_$cfg.txt" | tail -12 | cut -c1-330
done

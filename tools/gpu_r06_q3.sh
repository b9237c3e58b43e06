#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R; ulimit -c 0
for cfg in "nms2d_side_stream=0" ""; do
  SD_OPTS="$cfg" timeout 500 python tools/contention_check.py 8 40 2d > "$O/contention3_2d_p8_$cfg.txt" 2>&1; grep -v "^pid" "$O/contention3_2d_p8_$cfg.txt" | tail -9 | cut -c1-260
done

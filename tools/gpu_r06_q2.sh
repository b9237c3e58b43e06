#!/bin/bash
# round 6: which switch of the 2D NMS makes its result depend on the load of the device?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R; ulimit -c 0
for cfg in "" "nms2d_strict=1" "nms2d_defer_undecided=0" "nms2d_neighbours_single_pass=0"; do
  SD_OPTS="$cfg" timeout 400 python tools/contention_check.py 8 25 2d > "$O/contention_2d_p8_$cfg.txt" 2>&1; grep -v "^pid" "$O/contention_2d_p8_$cfg.txt" | tail -9 | cut -c1-260
done

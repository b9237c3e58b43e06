#!/bin/bash
# round 5: which greedy rounds of the 2D NMS hand their undecided pairs to the tail batch's sweep launch (options nms2d_defer_undecided / nms2d_defer_max)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R; ulimit -c 0
rm -f $O/defer_scan.txt
for cfg in "2 16384" "1 65536" "1 131072" "2 65536" "1 16384"; do
  set -- $cfg
  echo "== nms2d_defer_undecided=$1 nms2d_defer_max=$2" >> $O/defer_scan.txt
  SD_DEFER_FROM=$1 SD_DEFER_MAX=$2 timeout 120 python tools/time_nms2d_bench.py 4 2>&1 | grep "^rep" >> $O/defer_scan.txt
done
( time timeout 300 python -m pytest tests/test_gpu_parity2d.py -m gpu -q -x 2>&1 | tail -4 ) >> $O/defer_scan.txt 2>&1
cat $O/defer_scan.txt

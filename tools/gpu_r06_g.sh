#!/bin/bash
# late-round pair sweeps in 16- / 8-lane workgroups: parity + timing A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06g; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 900 python -m pytest tests/test_gpu_parity2d.py -m gpu -q -x 2>&1 | tail -5 ) > $O/tests.log 2>&1
for L in 0 16 8 16 0; do
  echo "== nms2d_late_pair_lanes $L"; SD_OPTS="nms2d_late_pair_lanes=$L" timeout 120 python tools/time_nms2d_bench.py 8 2>&1 | grep -v amdgpu.ids
done > $O/late_lanes.txt 2>&1
for L in 0 16; do
  echo "== nms2d_late_pair_lanes $L (trace)"; SD_TRACE=1 SD_OPTS="nms2d_late_pair_lanes=$L" timeout 120 python tools/time_nms2d_bench.py 2 2>&1 | grep -v amdgpu.ids
done > $O/late_lanes_trace.txt 2>&1
for L in 0 16; do
  echo "== strict, nms2d_late_pair_lanes $L"; SD_OPTS="nms2d_strict=1,nms2d_late_pair_lanes=$L" timeout 120 python tools/time_nms2d_bench.py 5 2>&1 | grep -v amdgpu.ids
done > $O/late_lanes_strict.txt 2>&1
tail -3 $O/tests.log; cat $O/late_lanes.txt

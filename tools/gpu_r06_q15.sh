#!/bin/bash
# a stand-alone persistent float kernel, launched twice on the same input, in 8 processes on one device; modes: 0 registers only,
# 1 + a gather from global memory per step, 2 + an exchange through LDS per step, 3 both
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R; ulimit -c 0
for M in 1 2 3; do
  for k in 1 2 3 4 5 6 7 8; do timeout 300 tools/bin/probe_timeslice 60 3000 $M > $O/timeslice_m${M}_$k.txt 2>&1 & done
  wait
  cat $O/timeslice_m${M}_*.txt | cut -c1-220
done

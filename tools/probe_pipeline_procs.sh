#!/bin/bash
# Is the two-stream pipeline of predict_instances_sharded result-identical to the serial loop?  Separate processes, ONE model:
# the first process calibrates and saves the head parameters, the others load them.  Two more processes calibrate on their own:
# their parameter hashes show whether the calibration itself is repeatable across processes.
# usage (GPU box): tools/probe_pipeline_procs.sh <tag>   -> gpurun_out/<tag>_*.log
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; O=$R/gpurun_out; ulimit -c 0
for dim in 2d 3d; do
  export SD_HEADS=/tmp/heads_$dim.pt; rm -f $SD_HEADS
  runs="serial pipe serial pipe pipe"; own="serial serial"
  [ $dim = 3d ] && runs="serial pipe pipe" && own="serial"
  for run in $runs; do
    echo "== $dim-$run" >> $O/${tag}_$dim.log
    timeout 300 python $R/tools/probe_pipeline.py $dim-$run >> $O/${tag}_$dim.log 2>&1
  done
  unset SD_HEADS
  for run in $own; do
    echo "== $dim-$run, own calibration" >> $O/${tag}_$dim.log
    timeout 300 python $R/tools/probe_pipeline.py $dim-$run >> $O/${tag}_$dim.log 2>&1
  done
done
grep -h "==\|sha1\|rep" $O/${tag}_2d.log $O/${tag}_3d.log | cut -c1-200

#!/bin/bash
# usage (on the GPU box): tools/gpu_pmc.sh <out-subdir> <kernel-substring> "<counters pass 1>" ["<counters pass 2>" ...] -- <command...>
# One rocprofv3 --pmc pass per counter group (never combined with trace domains other than kernel-trace), summarised per kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
sub=$1; shift; flt=$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
O=$R/gpurun_out/$sub; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for P in "${groups[@]}"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmc_$i -o p -- "$@" > $O/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name 'p_counter_collection.csv' | head -1)
  python $R/tools/pmc_multi.py ${f%_counter_collection.csv} "$flt" > $O/pmc_$i.md 2>&1
done
cat $O/pmc_*.md

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R; ulimit -c 0
for P in 2 4; do
  timeout 600 python tools/contention_check.py $P $((300 / P)) 2d > "$O/contention12_2d_p$P.txt" 2>&1
  echo "P=$P: $(grep -o 'keep [0-9a-f]* survivors [0-9]*' $O/contention12_2d_p$P.txt | sort | uniq -c | tr '\n' ';')"
done

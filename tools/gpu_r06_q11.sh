#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R; ulimit -c 0
for L in 0 1; do
  SD_LIB=stardist_amd/csrc/libstardist_hip_probe.so timeout 400 python tools/inprocess_load_check.py 150 $L > $O/inprocess_load_$L.txt 2>&1
  echo "load $L: rounds whose second evaluation disagrees: $(grep -c 'pairs with a different flag' $O/inprocess_load_$L.txt | tr -d '\n') of $(grep -c '^probe round' $O/inprocess_load_$L.txt), with mismatches: $(grep '^probe round' $O/inprocess_load_$L.txt | grep -vc ': 0 pairs')"; tail -1 $O/inprocess_load_$L.txt | cut -c1-200
done

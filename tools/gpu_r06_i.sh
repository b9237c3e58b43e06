#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06i; mkdir -p $O; cd $R/tools; ulimit -c 0
for v in "" oldnms3d "" oldnms3d; do STARDIST_AMD_PROBE_LIB=$v timeout 400 python ab_sharded3d.py 2>&1 | grep "RESULT\|rror" ; done > $O/ab_sharded3d.txt 2>&1
cat $O/ab_sharded3d.txt

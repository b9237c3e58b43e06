#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
timeout 200 python tools/ab_hull.py stardist_amd/csrc/libstardist_hip.so > $O/s15_hull_new.log 2>&1
grep "^R=" $O/s15_hull_new.log; tail -3 $O/s15_hull_new.log | cut -c1-300
timeout 200 python tools/time_predict_sections.py 2>&1 | grep -A12 "^3D"

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
timeout 200 python tools/pyprof_step.py 3d > $O/s14_pyprof3d.log 2>&1
timeout 200 python tools/pyprof_step.py 2d > $O/s14_pyprof2d.log 2>&1
grep -v "^/opt" $O/s14_pyprof3d.log | head -48 | cut -c1-170

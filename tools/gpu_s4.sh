#!/bin/bash
# session 4 of round 4: big == whole diagnostics, host-input section timing
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
timeout 200 python tools/time_predict_sections.py --host-input > $O/s4_sections_host.log 2>&1
timeout 400 python tools/diag_sharded.py 2d 128 192 256 > $O/s4_diag_2d.log 2>&1
timeout 400 python tools/diag_sharded.py 3d512 32 48 64 > $O/s4_diag_3d512.log 2>&1
timeout 400 python tools/diag_sharded.py 3d1024 32 > $O/s4_diag_3d1024.log 2>&1
cat $O/s4_sections_host.log; tail -12 $O/s4_diag_2d.log; tail -12 $O/s4_diag_3d512.log; tail -8 $O/s4_diag_3d1024.log

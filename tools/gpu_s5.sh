#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
timeout 200 python tools/time_predict_sections.py --host-input > $O/s5_sections_host.log 2>&1
timeout 600 python tools/diag_sharded.py 2d 128 192 256 384 > $O/s5_diag_2d.log 2>&1
( time timeout 600 python -m pytest tests/test_gpu_bigparity.py -m gpu -q -s -k 3d 2>&1 ) > $O/s5_bigparity3d.log 2>&1
grep -v "^frame" $O/s5_sections_host.log | head -14; grep -v "^frame" $O/s5_diag_2d.log | tail -12; tail -15 $O/s5_bigparity3d.log

#!/bin/bash
# round 6, last session: the 3D rounds with one read-back for survivors / undecided / stage-3 pairs and none for the hull count
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06l; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 900 python -m pytest tests/test_gpu_parity3d.py tests/test_gpu_fullsize_parity.py tests/test_gpu_lattice.py tests/test_gpu_relabel.py tests/test_gpu_bigparity.py -m gpu -q -x 2>&1 | tail -8 ) > $O/tests.log 2>&1
tail -4 $O/tests.log
SD_TRACE=1 timeout 120 python tools/time_nms3d_bench.py 3 > $O/nms3d_rounds_trace.txt 2>&1
grep "rep " $O/nms3d_rounds_trace.txt
timeout 200 python tools/time_predict_sections.py > $O/sections.log 2>&1; grep -A12 "^3D" $O/sections.log
timeout 200 python tools/time_nms3d.py 480 2 > $O/nms3d_1M.log 2>&1; tail -3 $O/nms3d_1M.log

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06j; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 900 python -m pytest tests/test_gpu_parity2d.py tests/test_gpu_beam_prep.py tests/test_gpu_lattice.py -m gpu -q -x -k "not 3d" 2>&1 | tail -3 ) > $O/tests2.log 2>&1
cat $O/tests2.log
timeout 120 python tools/time_nms2d_bench.py 8 2>&1 | grep -v amdgpu.ids > $O/nms2d_2.txt; tail -5 $O/nms2d_2.txt
SD_OPTS="nms2d_strict=1" timeout 120 python tools/time_nms2d_bench.py 5 2>&1 | grep "rep [34]"

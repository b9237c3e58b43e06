#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06h; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 1200 python -m pytest tests/test_gpu_lattice.py tests/test_gpu_parity3d.py tests/test_gpu_relabel.py -m gpu -q -x -rxXs 2>&1 | tail -15 ) > $O/tests.log 2>&1
cat $O/tests.log
timeout 300 python -m pytest -s -q tests/test_gpu_lattice.py -m gpu -k "cartesian_pair" 2>&1 | grep "cartesian\|passed\|failed" > $O/cartesian_pair_volumes.txt; cat $O/cartesian_pair_volumes.txt
SD_TRACE=0 timeout 120 python tools/time_nms3d_bench.py 3 2>&1 | grep -v amdgpu.ids | tail -4

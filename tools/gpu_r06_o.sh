#!/bin/bash
# round 6, last session: emission kernels with a workgroup per survivor
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06o; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 900 python -m pytest tests/test_gpu_parity2d.py tests/test_gpu_parity3d.py tests/test_gpu_lattice.py tests/test_gpu_fullsize_parity.py tests/test_gpu_bigparity.py -m gpu -q -x 2>&1 | tail -8 ) > $O/tests.log 2>&1
tail -4 $O/tests.log
timeout 200 python tools/time_predict_sections.py > $O/sections.log 2>&1; cat $O/sections.log | head -24

#!/bin/bash
# round 5: "nms3d_bounds_lean" (bounds-only launches at seven waves per CU): the 3D parity tests with it on (the build's default), then the A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05g; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 400 python -m pytest tests/test_gpu_parity3d.py tests/test_gpu_fullsize_parity.py tests/test_gpu_bigparity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "3d or 3D or dim" 2>&1 | tail -6 ) > $O/tests3d.log 2>&1
export SD_COMBOS="nms3d_bounds_lean=0;;nms3d_bounds_lean=0;"
( time timeout 200 python tools/time_nms3d_options.py 9 2>&1 | grep -v "^hiv:\|wave cycles\|^round\|^  \|^tail\|^ray\|^----" ) > $O/nms3d_lean.txt 2>&1
cut -c1-250 $O/tests3d.log; cut -c1-200 $O/nms3d_lean.txt

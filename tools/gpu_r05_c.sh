#!/bin/bash
# round 5, validation of the 3D NMS switches (nms3d_bounds_reuse, nms3d_defer_exact): parity tests, then the A/B on the bench set
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 500 python -m pytest tests/test_gpu_parity3d.py -m gpu -q -s -k "carried or reuse or tail_batch or neighbour_list or split_exact or volume_bounds or nuclei_survivors" 2>&1 | tail -25 ) > $O/tests3d.log 2>&1
( time timeout 300 python -m pytest tests/test_gpu_fullsize_parity.py -m gpu -q -k "nms3d_256 or 3d_end_to_end" 2>&1 | tail -8 ) > $O/tests3d_full.log 2>&1
( time timeout 300 python tools/time_nms3d_options.py 7 2>&1 | grep -v "^hiv:\|wave cycles" ) > $O/nms3d_options.txt 2>&1
cut -c1-260 $O/tests3d.log; cut -c1-260 $O/tests3d_full.log; cut -c1-260 $O/nms3d_options.txt

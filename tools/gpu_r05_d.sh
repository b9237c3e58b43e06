#!/bin/bash
# round 5: tail-batch threshold / split form with the exact volumes carried over (nms3d_defer_exact = 3, the new default)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05d; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 300 python -m pytest tests/test_gpu_parity3d.py -m gpu -q -s -k "carried" 2>&1 | tail -12 ) > $O/tests3d.log 2>&1
export SD_COMBOS=";nms3d_defer_exact=0;nms3d_tail_batch=16;nms3d_tail_batch=64;nms3d_tail_batch=128;nms3d_tail_batch=512;nms3d_split_exact=3;nms3d_split_exact=3,nms3d_tail_batch=64;nms3d_defer_exact=4,nms3d_tail_batch=64"
( time timeout 300 python tools/time_nms3d_options.py 7 2>&1 | grep -v "^hiv:\|wave cycles" ) > $O/nms3d_options.txt 2>&1
cut -c1-300 $O/tests3d.log; cut -c1-260 $O/nms3d_options.txt

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_parity2d.py -m gpu -q -s -k "area" 2>&1 | grep -v "^$" | tail -25 | cut -c1-250
SD_TRACE=1 timeout 200 python tools/time_nms2d_bench.py 2 2>&1 | tail -7 | cut -c1-250
for d in 6 3 2; do echo "tail div $d"; SD_NMS_TAIL_DIV=$d SD_NMS_TAIL_MAX=1000000 SD_LIB=tools/_ab/libstardist_hip_dbg.so timeout 200 python tools/time_nms2d_bench.py 3 2>&1 | grep "^rep [12]" | cut -c1-200; done

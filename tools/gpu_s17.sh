#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
SD_TRACE=1 timeout 200 python tools/time_nms3d_bench.py 2 2>&1 | tail -20 | cut -c1-200
timeout 400 python -m pytest tests/test_gpu_parity3d.py tests/test_gpu_bigparity.py -m gpu -x -q -k "3d or 3D" 2>&1 | tail -3
timeout 200 python tools/time_predict_sections.py 2>&1 | grep -A12 "^3D"

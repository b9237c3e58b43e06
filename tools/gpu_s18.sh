#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_parity2d.py -m gpu -q -x -k "nms2d" 2>&1 | tail -2 | cut -c1-250
timeout 200 python tools/time_nms2d_bench.py 4 2>&1 | tail -3 | cut -c1-250
timeout 200 python tools/time_predict_sections.py 2>&1 | grep -B1 -A10 "net_forward" | head -12

#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( SD_NMS_PAIR_SORT=0 timeout 200 python tools/time_nms2d_bench.py 4 2>&1 | tail -4 ) > gpurun_out/c5_nms_sort0.log 2>&1
( SD_NMS_PAIR_SORT=1 timeout 200 python tools/time_nms2d_bench.py 4 2>&1 | tail -4 ) > gpurun_out/c5_nms_sort1.log 2>&1
( timeout 300 python -m pytest tests/test_gpu_parity2d.py tests/test_gpu_fullsize_parity.py -q -x -k "not 3d" 2>&1 | tail -5 ) > gpurun_out/c5_parity2d.log 2>&1
cat gpurun_out/c5_nms_sort0.log gpurun_out/c5_nms_sort1.log; tail -3 gpurun_out/c5_parity2d.log

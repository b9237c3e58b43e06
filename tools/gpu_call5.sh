#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for m in 0 1 2 3 4 5; do
( SD_NMS_PAIR_KEY=$m timeout 200 python tools/time_nms2d_bench.py 3 2>&1 | tail -1 ) > gpurun_out/c5_nms_key$m.log 2>&1
echo "key $m: $(cat gpurun_out/c5_nms_key$m.log)"
done

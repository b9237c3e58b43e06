#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 200 python -m pytest tests/test_gpu_conv3x3.py -q -x 2>&1 | tail -8 ) > gpurun_out/c3_convtest.log 2>&1
( timeout 300 python tools/probe_hand_conv.py --reps 4 2>&1 | grep -v "^$" | tail -40 ) > gpurun_out/c3_probe.log 2>&1
( timeout 200 python tools/time_predict_sections.py --steps 5 2>&1 | tail -28 ) > gpurun_out/c3_sections.log 2>&1
tail -3 gpurun_out/c3_convtest.log; grep "network conv" gpurun_out/c3_probe.log

#!/bin/bash
# one-shot GPU session: conv kernel parity, per-layer A/B, bench A/B, section timing.  Logs under gpurun_out/
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_conv3x3.py -q -x 2>&1 | tail -25 ) > gpurun_out/c1_convtest.log 2>&1
echo "convtest rc=$?" >> gpurun_out/c1_convtest.log
( timeout 300 python -m pytest tests/test_gpu_unet_parity.py tests/test_gpu_heads.py tests/test_gpu_unet_ops.py -q 2>&1 | tail -15 ) > gpurun_out/c1_unettests.log 2>&1
( timeout 400 python tools/probe_hand_conv.py --reps 4 2>&1 | grep -v "^$" | tail -60 ) > gpurun_out/c1_probe.log 2>&1
( timeout 300 python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>&1 | tail -3 ) > gpurun_out/c1_bench_hand.json 2>&1
( timeout 120 python tools/time_predict_sections.py 2>&1 | tail -20 ) > gpurun_out/c1_sections.log 2>&1
tail -5 gpurun_out/c1_convtest.log; tail -3 gpurun_out/c1_unettests.log; tail -4 gpurun_out/c1_probe.log

#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_conv3x3.py -q -x 2>&1 | tail -12 ) > gpurun_out/c6_convtest.log 2>&1
( timeout 200 python -m pytest tests/test_gpu_unet_parity.py -q -x -s -k "bf16x6 or unet2d or unet3d" 2>&1 | grep -v "^$" | tail -14 ) > gpurun_out/c6_unet.log 2>&1
( timeout 200 python tools/probe_hand_conv.py --reps 4 --no-lib 2>&1 | grep "^2D\|^3D" | cut -c1-64,118-200 ) > gpurun_out/c6_probe.log 2>&1
( STARDIST_AMD_CONV=bf16x6 timeout 200 python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>&1 | grep '^{' ) > gpurun_out/c6_bench_bf16x6.json 2>&1
tail -4 gpurun_out/c6_convtest.log; tail -6 gpurun_out/c6_unet.log; cat gpurun_out/c6_probe.log; cut -c1-330 gpurun_out/c6_bench_bf16x6.json

#!/bin/bash
# round 6, first GPU call: split16 activations -- parity tests of the new forms, per-layer A/B (f32 tensors / split16, precomputed offsets on / off),
# phase profiles, a short bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 900 python -m pytest tests/test_gpu_split16.py tests/test_gpu_conv3x3.py tests/test_gpu_unet_ops.py tests/test_gpu_heads.py tests/test_gpu_unet_parity.py -m gpu -q -x 2>&1 | tail -25 ) > $O/tests_conv.log 2>&1
PROBE_F16_ONLY=1 timeout 300 python tools/probe_hand_conv.py --reps 5 > $O/layer_probe_exp1.txt 2>&1
PROBE_F16_ONLY=1 STARDIST_AMD_PROBE_LIB=exp0 timeout 300 python tools/probe_hand_conv.py --reps 5 > $O/layer_probe_exp0.txt 2>&1
timeout 120 tools/bin/cpp16_exp1 2 0 > $O/phases_f32.txt 2>&1
timeout 120 tools/bin/cpp16_exp1 2 1 > $O/phases_split16_exp1.txt 2>&1
timeout 120 tools/bin/cpp16_exp0 2 1 > $O/phases_split16_exp0.txt 2>&1
timeout 400 python bench.py --gpus 1 --steps 10 --warmup 3 --no-sharded --no-split-leg --no-cpu-baseline > $O/bench_short.json 2> $O/bench_short.err
STARDIST_AMD_SPLIT16=0 timeout 400 python bench.py --gpus 1 --steps 10 --warmup 3 --no-sharded --no-split-leg --no-cpu-baseline > $O/bench_short_f32tensors.json 2> $O/bench_short_f32tensors.err
tail -6 $O/tests_conv.log; tail -3 $O/layer_probe_exp1.txt; grep "network conv" $O/layer_probe_exp0.txt; cut -c1-700 $O/bench_short.json; tail -3 $O/bench_short.err; cut -c1-300 $O/bench_short_f32tensors.json

#!/bin/bash
# round 6: split16 -- the whole network-side GPU suite at HEAD, the power-roof evidence (MFMA on real data, kernel on zero data), a short bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06c; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 1200 python -m pytest tests/test_gpu_split16.py tests/test_gpu_conv3x3.py tests/test_gpu_unet_ops.py tests/test_gpu_heads.py tests/test_gpu_unet_parity.py tests/test_gpu_fullsize_parity.py -m gpu -q 2>&1 | tail -25 ) > $O/tests_conv.log 2>&1
tools/bin/probe_mfma_power > $O/mfma_power_roof.txt 2>&1
( echo "== real data"; tools/bin/cpp16_exp1 2 1 0 | grep kz; echo "== all-zero activations and weights (same instruction stream)"; tools/bin/cpp16_exp1 2 1 1 | grep kz ) > $O/conv_zero_vs_real.txt 2>&1
( for e in 1 7 11 15; do echo "== SD_CONV_EXP $e (bit 2: two of three A-operand LDS reads skipped, bit 3: B)"; tools/bin/cpp16_exp$e 2 1 0 | grep kz | cut -c1-90; done ) > $O/conv_lds_energy_probe.txt 2>&1
timeout 120 tools/bin/cpp16_exp1 2 1 > $O/phases_split16.txt 2>&1
timeout 120 tools/bin/cpp16_exp1 2 0 > $O/phases_f32.txt 2>&1
PROBE_F16_ONLY=1 timeout 300 python tools/probe_hand_conv.py --reps 5 > $O/layer_probe.txt 2>&1
tail -6 $O/tests_conv.log; cat $O/mfma_power_roof.txt

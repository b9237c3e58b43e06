#!/bin/bash
# session 1 of round 4: the split-fp16 convolution -- denormal probe, layer tests, network parity, layer timing
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
/opt/rocm/bin/hipcc --offload-arch=gfx950 tools/probe_mfma_f16_denorm.hip -o /tmp/probe_denorm 2>/dev/null && /tmp/probe_denorm > $O/s1_denorm.txt 2>&1
( time timeout 500 python -m pytest tests/test_gpu_conv3x3.py -m gpu -q -x 2>&1 | tail -15 ) > $O/s1_conv_tests.log 2>&1
( time timeout 400 python -m pytest tests/test_gpu_unet_parity.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -40 ) > $O/s1_unet_tests.log 2>&1
timeout 400 python tools/probe_hand_conv.py --reps 4 > $O/s1_layer_probe.txt 2>&1
cat $O/s1_denorm.txt; tail -5 $O/s1_conv_tests.log; tail -12 $O/s1_unet_tests.log; tail -30 $O/s1_layer_probe.txt

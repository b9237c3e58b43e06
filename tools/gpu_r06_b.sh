#!/bin/bash
# round 6: conv kernel iteration -- split16 tests, phase profile (split16), layer probe
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06b; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 900 python -m pytest tests/test_gpu_split16.py -m gpu -q -x 2>&1 | tail -25 ) > $O/tests_split16.log 2>&1
timeout 120 tools/bin/cpp16_exp1 2 1 > $O/phases_split16.txt 2>&1
PROBE_F16_ONLY=1 timeout 300 python tools/probe_hand_conv.py --reps 5 > $O/layer_probe.txt 2>&1
tail -4 $O/tests_split16.log; grep "network conv" $O/layer_probe.txt

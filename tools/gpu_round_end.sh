#!/bin/bash
# End-of-round measurement set on the GPU box: full GPU suite, smoke, the driver's bench command, the profile set of
# tools/profile_round.sh (kernel trace of the bench, PMC passes, per-forward convolution durations + traffic), section timing,
# the 1M-candidate 3D NMS, the per-layer convolution probe, the phase profile of the split-fp16 kernel, the network-vs-float64 log.
# usage: tools/gpu_round_end.sh <tag> <round tag for the profiles, e.g. r04>   -> gpurun_out/<tag>_*
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; rtag=$2; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > $O/${tag}_tests.log 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${tag}_smoke.log 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${tag}_bench.json 2> $O/${tag}_bench.err
timeout 1200 tools/profile_round.sh $rtag > $O/${tag}_profile_stdout.log 2>&1
timeout 200 python tools/time_predict_sections.py > $O/${tag}_sections.log 2>&1
timeout 200 python tools/time_predict_sections.py --host-input > $O/${tag}_sections_host.log 2>&1
timeout 200 python tools/time_nms3d.py 480 2 > $O/${tag}_nms3d_1M.log 2>&1
timeout 300 python tools/probe_hand_conv.py --reps 4 > $O/${tag}_conv_layer_probe.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Istardist_amd/csrc tools/conv_f16_phase_profile.hip -o /tmp/cpp16 2>/dev/null && ( /tmp/cpp16 2; /tmp/cpp16 1 ) > $O/${tag}_conv_f16_phases.txt 2>&1
timeout 300 python -m pytest -s -q tests/test_gpu_unet_parity.py -m gpu > $O/${tag}_unet_parity.log 2>&1
SD_TRACE=1 timeout 120 python tools/time_nms2d_bench.py 2 > $O/${tag}_nms2d_rounds_trace.txt 2>&1
SD_TRACE=1 timeout 120 python tools/time_nms3d_bench.py 2 > $O/${tag}_nms3d_rounds_trace.txt 2>&1
timeout 120 python tools/check_defer.py > $O/${tag}_nms2d_defer_undecided.txt 2>&1
timeout 200 python -m pytest -s -q tests/test_gpu_parity2d.py -m gpu -k area > $O/${tag}_area_enclosure_validation.txt 2>&1
timeout 120 python tools/ab_hull.py stardist_amd/csrc/libstardist_hip.so > $O/${tag}_hull_ab.txt 2>&1
tail -3 $O/${tag}_tests.log; cat $O/${tag}_smoke.log | tail -1; cut -c1-200 $O/${tag}_bench.json

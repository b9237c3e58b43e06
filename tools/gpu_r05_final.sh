#!/bin/bash
# End-of-round-5 measurement set on the GPU box, everything at HEAD: full GPU suite, smoke, the driver's bench command, the profile set of
# tools/profile_round.sh (kernel trace of the bench, PMC passes, per-forward convolution durations + traffic), the device timelines of one
# 2D / 3D step, section timing, NMS round traces, the 1M-candidate 3D NMS, the convolution layer probe and phase profile, the
# network-vs-float64 log, the area-enclosure validation against the exact sweep.
# usage: tools/gpu_r05_final.sh   -> gpurun_out/r05f/*
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05f; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > $O/tests.log 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/final_bench.json 2> $O/final_bench.err ) > $O/bench_time.log 2>&1
cd /tmp; export TMPDIR=/tmp
for W in 2d 3d; do
  rm -rf /tmp/tl_$W
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_$W -o p -- python $R/tools/step_timeline.py run $W 2 > $O/timeline_run_$W.log 2>&1
  python $R/tools/step_timeline.py report /tmp/tl_$W $W > $O/step_timeline_$W.txt 2>&1
done
cd $R
timeout 1500 tools/profile_round.sh r05 > $O/profile_stdout.log 2>&1
timeout 200 python tools/time_predict_sections.py > $O/sections.log 2>&1
timeout 200 python tools/time_predict_sections.py --host-input > $O/sections_host.log 2>&1
SD_TRACE=1 timeout 120 python tools/time_nms2d_bench.py 2 > $O/nms2d_rounds_trace.txt 2>&1
SD_TRACE=1 timeout 120 python tools/time_nms3d_bench.py 2 > $O/nms3d_rounds_trace.txt 2>&1
timeout 200 python tools/time_nms3d.py 480 2 > $O/nms3d_1M.log 2>&1
timeout 200 python -m pytest -s -q tests/test_gpu_parity2d.py -m gpu -k area > $O/area_enclosure_validation.txt 2>&1
timeout 120 python tools/check_defer.py > $O/nms2d_defer_undecided.txt 2>&1
timeout 300 python tools/probe_hand_conv.py --reps 4 > $O/conv_layer_probe.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Istardist_amd/csrc tools/conv_f16_phase_profile.hip -o /tmp/cpp16 2>/dev/null && ( /tmp/cpp16 2; /tmp/cpp16 1 ) > $O/conv_f16_phases.txt 2>&1
timeout 300 python -m pytest -s -q tests/test_gpu_unet_parity.py -m gpu > $O/unet_parity.log 2>&1
tail -3 $O/tests.log; tail -1 $O/smoke.log; cut -c1-300 $O/final_bench.json; tail -3 $O/bench_time.log

#!/bin/bash
# End-of-round-6 measurement set on the GPU box, everything at HEAD: full GPU suite, smoke, the driver's bench command, the profile set of
# tools/profile_round.sh (kernel trace of the bench, PMC passes, per-forward convolution durations + traffic), the device timelines of one
# 2D / 3D step, section timing, NMS round traces, the 1M-candidate 3D NMS, the convolution layer probe and phase profile, the
# network-vs-float64 log, the area-enclosure validation against the exact sweep.
# usage: tools/gpu_r05_final.sh   -> gpurun_out/r06f/*
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06f; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > $O/tests.log 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/final_bench.json 2> $O/final_bench.err ) > $O/bench_time.log 2>&1
cd /tmp; export TMPDIR=/tmp
for W in 2d 3d; do
  rm -rf /tmp/tl_$W
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_$W -o p -- python $R/tools/step_timeline.py run $W 2 > $O/timeline_run_$W.log 2>&1
  python $R/tools/step_timeline.py report /tmp/tl_$W $W > $O/step_timeline_$W.txt 2>&1
done
cd $R
timeout 1500 tools/profile_round.sh r06 > $O/profile_stdout.log 2>&1
timeout 200 python tools/time_predict_sections.py > $O/sections.log 2>&1
timeout 200 python tools/time_predict_sections.py --host-input > $O/sections_host.log 2>&1
SD_TRACE=1 timeout 120 python tools/time_nms2d_bench.py 2 > $O/nms2d_rounds_trace.txt 2>&1
SD_TRACE=1 timeout 120 python tools/time_nms3d_bench.py 2 > $O/nms3d_rounds_trace.txt 2>&1
timeout 200 python tools/time_nms3d.py 480 2 > $O/nms3d_1M.log 2>&1
timeout 200 python -m pytest -s -q tests/test_gpu_parity2d.py -m gpu -k area > $O/area_enclosure_validation.txt 2>&1
timeout 120 python tools/check_defer.py > $O/nms2d_defer_undecided.txt 2>&1
timeout 300 python tools/probe_hand_conv.py --reps 4 > $O/conv_layer_probe.txt 2>&1
# the convolution kernel: phase profile (f32 tensors / split16 tensors), the power probes (same instruction stream on zero data; two of three
# LDS operand reads skipped; the matrix pipe alone on zero / constant / changing operands)
( tools/bin/cpp16_exp1 2 1; tools/bin/cpp16_exp1 2 0 ) > $O/conv_f16_phases.txt 2>&1
( echo "== real data"; tools/bin/cpp16_exp1 2 1 0 | grep kz; echo "== all-zero activations and weights (the same instruction stream)"; tools/bin/cpp16_exp1 2 1 1 | grep kz ) > $O/conv_zero_vs_real.txt 2>&1
( for e in 1 7 11 15; do echo "== SD_CONV_EXP $e (bit 2: two of three A-operand LDS reads skipped, bit 3: B; wrong results, energy probe)"; tools/bin/cpp16_exp$e 2 1 0 | grep kz | cut -c1-120; done ) > $O/conv_lds_energy_probe.txt 2>&1
tools/bin/probe_mfma_power > $O/mfma_power_roof.txt 2>&1; tools/bin/probe_mfma_power >> $O/mfma_power_roof.txt 2>&1
tools/bin/probe_memtime > $O/memtime_calibration.txt 2>&1
timeout 200 python tools/time_conv_rows.py > $O/conv_rows_timing.txt 2>&1
timeout 300 python -m pytest -s -q tests/test_gpu_unet_parity.py -m gpu > $O/unet_parity.log 2>&1
tail -3 $O/tests.log; tail -1 $O/smoke.log; cut -c1-300 $O/final_bench.json; tail -3 $O/bench_time.log

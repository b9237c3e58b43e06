#!/bin/bash
# End-of-round-6 measurement set, second edition (the round kernels / score sort / 3D read-backs of the last session): everything of
# tools/gpu_r06_final.sh except the probes of the convolution kernel alone (its code did not change: profiles/r06_conv_power_*,
# r06_conv_f16_phases, r06_mfma_power_roof, r06_memtime_calibration, r06_conv_layer_probe, r06_conv_rows_timing stay as taken at f67e65c).
# usage: tools/gpu_r06_final2.sh   -> gpurun_out/r06f/*, gpurun_out/r06/*; then tools/copy_r06_profiles.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06f; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > $O/tests.log 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/final_bench.json 2> $O/final_bench.err ) > $O/bench_time.log 2>&1
tail -3 $O/tests.log; tail -1 $O/smoke.log; cut -c1-300 $O/final_bench.json; tail -3 $O/bench_time.log
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
timeout 300 python -m pytest -s -q tests/test_gpu_unet_parity.py -m gpu > $O/unet_parity.log 2>&1
head -12 $O/sections.log; tail -3 $O/profile_stdout.log | cut -c1-300

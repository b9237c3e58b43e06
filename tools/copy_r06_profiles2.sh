#!/bin/bash
# copies the summaries of tools/gpu_r06_final2.sh (no convolution-only probes) (gpurun_out/r06f, gpurun_out/r06) into profiles/ under their committed names
cd "$(dirname "$0")/.."; F=gpurun_out/r06f; P=gpurun_out/r06
cp $F/area_enclosure_validation.txt profiles/r06_area_enclosure_validation.txt
cp $P/kernel_stats.md profiles/r06_bench_kernel_stats.md
cp $P/bench_under_trace.json profiles/r06_bench_under_trace.json
cp $P/conv_forward.md profiles/r06_conv_forward.md
cp $F/final_bench.json profiles/r06_final_bench.json
cp $F/nms2d_defer_undecided.txt profiles/r06_nms2d_defer_undecided.txt
cp $F/nms2d_rounds_trace.txt profiles/r06_nms2d_rounds_trace.txt
cp $F/nms3d_1M.log profiles/r06_nms3d_1M_candidates.txt
cp $F/nms3d_rounds_trace.txt profiles/r06_nms3d_rounds_trace.txt
cp $P/pmc_conv_stalls.md profiles/r06_pmc_conv_stalls.md
cp $P/pmc_hbm_traffic.md profiles/r06_pmc_hbm_traffic.md
cp $P/pmc_mfma.md profiles/r06_pmc_mfma.md
( cat $F/sections.log; echo; echo "## host-array input"; cat $F/sections_host.log ) > profiles/r06_step_sections.txt
cp $F/step_timeline_2d.txt profiles/r06_step_timeline_2d.txt
cp $F/step_timeline_3d.txt profiles/r06_step_timeline_3d.txt
cp $F/unet_parity.log profiles/r06_unet_parity_vs_float64.txt
( tail -6 $F/tests.log; echo; cat $F/smoke.log | tail -2 ) > profiles/r06_final_gpu_suite.txt
cp $P/pair_kernel_traffic.json profiles/pair_kernel_traffic.json
cp $P/conv_kernel_traffic.json profiles/conv_kernel_traffic.json

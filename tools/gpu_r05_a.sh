#!/bin/bash
# round 5, first GPU call: the new legacy-NMS tests, the device timeline of one 2D / 3D step (kernels + copies), a short bench with the host-input and strict legs
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/tests_all.log 2>&1
cd /tmp; export TMPDIR=/tmp
for W in 2d 3d; do
  rm -rf /tmp/tl_$W
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_$W -o p -- python $R/tools/step_timeline.py run $W 2 > $O/timeline_run_$W.log 2>&1
  python $R/tools/step_timeline.py report /tmp/tl_$W $W > $O/step_timeline_$W.txt 2>&1; gzip -c /tmp/tl_$W/p_kernel_trace.csv > $O/kt_$W.csv.gz; gzip -c /tmp/tl_$W/p_memory_copy_trace.csv > $O/mc_$W.csv.gz
done
cd $R
timeout 400 python bench.py --gpus 1 --steps 10 --warmup 3 --no-sharded --no-split-leg --no-cpu-baseline > $O/bench_short.json 2> $O/bench_short.err
tail -4 $O/tests_all.log; head -3 $O/step_timeline_2d.txt | cut -c1-300; head -3 $O/step_timeline_3d.txt | cut -c1-300; cut -c1-600 $O/bench_short.json; tail -5 $O/bench_short.err

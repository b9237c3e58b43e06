#!/bin/bash
# end-of-round GPU session: profiles of the final state, full GPU suite, smoke, the driver's bench command
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
( time timeout 420 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 ) > gpurun_out/final_gputests.log 2>&1
( timeout 120 python __graft_entry__.py smoke 2>&1 | tail -3 ) > gpurun_out/final_smoke.log 2>&1
( timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep '^{' ) > gpurun_out/final_bench.json 2>gpurun_out/final_bench.err
bash tools/profile_round.sh r02c > gpurun_out/r02c_stdout.log 2>&1
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_2d; (cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_2d -o kt -- python bench.py --skip-3d --no-cpu-baseline --steps 5 --warmup 2 > $R/gpurun_out/r02c/bench2d_under_trace.log 2>&1)
db=$(find /tmp/prof_2d -name '*.db' | head -1)
python $R/tools/rocpd_summary.py $db --md --group k_conv3 14 > $R/gpurun_out/r02c/bench2d_kernel_stats.md 2>&1
grep '^{' $R/gpurun_out/r02c/bench2d_under_trace.log > $R/gpurun_out/r02c/bench2d_under_trace.json
tail -4 $R/gpurun_out/final_gputests.log; cat $R/gpurun_out/final_smoke.log; cut -c1-400 $R/gpurun_out/final_bench.json; head -4 $R/gpurun_out/r02c/bench2d_kernel_stats.md | cut -c1-200

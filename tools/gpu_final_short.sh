#!/bin/bash
# final measurement set of the round, in order of importance (GPU budget nearly spent): bench, kernel trace of the bench, GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
timeout 280 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/s20_bench.json 2> $O/s20_bench.err
cut -c1-400 $O/s20_bench.json
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_kt; (cd $R && timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python bench.py --no-cpu-baseline --no-sharded --no-split-leg --steps 5 --warmup 2 > $O/s20_bench_under_trace.log 2>&1)
db=$(find /tmp/prof_kt -name '*.db' | head -1)
python $R/tools/rocpd_summary.py $db --md > $O/s20_kernel_stats.md 2>&1
grep "^{" $O/s20_bench_under_trace.log > $O/s20_bench_under_trace.json
head -12 $O/s20_kernel_stats.md | cut -c1-70,160-230
cd $R
( time timeout 320 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 ) > $O/s20_tests.log 2>&1
tail -6 $O/s20_tests.log

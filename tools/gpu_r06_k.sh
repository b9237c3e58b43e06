#!/bin/bash
# round 6, last session: the round kernels with fewer atomics (k_round_triage per workgroup, k_round_emit staged in LDS, k_round_scan /
# k_round_decide3 four candidates per wave): parity suites, N = 2 plumbing with a 3D sharded input that fits two ranks on one device,
# bench, device timelines
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06k; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 900 python -m pytest tests/test_gpu_parity2d.py tests/test_gpu_parity3d.py tests/test_gpu_fullsize_parity.py tests/test_gpu_glue.py tests/test_gpu_bigparity.py -m gpu -q -x 2>&1 | tail -8 ) > $O/tests.log 2>&1
tail -4 $O/tests.log
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) > $O/bench_time.log 2>&1
cut -c1-400 $O/bench.json; tail -3 $O/bench_time.log
timeout 200 python tools/time_predict_sections.py > $O/sections.log 2>&1; head -20 $O/sections.log
cd /tmp; export TMPDIR=/tmp
for W in 2d 3d; do
  rm -rf /tmp/tl_$W
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_$W -o p -- python $R/tools/step_timeline.py run $W 2 > $O/timeline_run_$W.log 2>&1
  python $R/tools/step_timeline.py report /tmp/tl_$W $W > $O/step_timeline_$W.txt 2>&1
done
cd $R
grep -n "k_round\|busy" $O/step_timeline_2d.txt | head -20
export STARDIST_AMD_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-split-leg --sharded-size3d 768 --sharded-block3d 416 > $O/bench_2ranks_gloo.json 2> $O/bench_2ranks_gloo.err ) > $O/time_2.log 2>&1
echo rc=$? >> $O/time_2.log
grep -v "Gloo\|socket.cpp" $O/bench_2ranks_gloo.err | tail -c 1500; cut -c1-300 $O/bench_2ranks_gloo.json; cat $O/time_2.log

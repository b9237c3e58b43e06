#!/bin/bash
# round 5: the N > 1 form of bench.py on the one-GPU box (two ranks sharing the device, gloo collectives: every code path of the N > 1
# legs except the RCCL transport; the numbers mean nothing)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05e; mkdir -p $O; cd $R; ulimit -c 0
export STARDIST_AMD_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${NR:-2} --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus ${NR:-2} --steps 3 --warmup 1 --no-cpu-baseline --no-split-leg --skip-sharded-3d > $O/bench_${NR:-2}ranks_gloo.json 2> $O/bench_${NR:-2}ranks_gloo.err ) > $O/time.log 2>&1
echo rc=$? >> $O/time.log
tail -c 1500 $O/bench_${NR:-2}ranks_gloo.err; cut -c1-1200 $O/bench_${NR:-2}ranks_gloo.json; cat $O/time.log

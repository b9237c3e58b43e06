#!/bin/bash
# round 6: the N > 1 form of bench.py (default legs: 2D, 3D, sharded 16384^2 and 1024^3 through ShardedInput) on the one-GPU box -- NR ranks
# sharing the device, gloo collectives: every code path of the N > 1 legs except the RCCL transport; the numbers mean nothing
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06e2; mkdir -p $O; cd $R; ulimit -c 0
export STARDIST_AMD_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for NR in ${NRS:-2 4}; do
( time timeout 700 python -m torch.distributed.run --nnodes=1 --nproc-per-node $NR --master-addr 127.0.0.1 --master-port 2951$NR bench.py --gpus $NR --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${NR}ranks_gloo.json 2> $O/bench_${NR}ranks_gloo.err ) > $O/time_$NR.log 2>&1
echo rc=$? >> $O/time_$NR.log
tail -c 800 $O/bench_${NR}ranks_gloo.err; cut -c1-1500 $O/bench_${NR}ranks_gloo.json; cat $O/time_$NR.log
done

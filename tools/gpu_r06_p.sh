#!/bin/bash
# round 6: the whole bench at N = 8 on the one-GPU box (eight ranks sharing the device, gloo collectives; the 3D sharded input reduced to
# 512^3 in 8 blocks of 320^3 so that eight ranks fit one device: one block per rank) -- plumbing only, the numbers mean nothing
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06p; mkdir -p $O; cd $R; ulimit -c 0
export STARDIST_AMD_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29538 bench.py --gpus 8 --steps 2 --warmup 1 --no-cpu-baseline --no-split-leg --sharded-size3d 512 --sharded-block3d 320 > $O/bench_8ranks_gloo.json 2> $O/bench_8ranks_gloo.err ) > $O/time_8.log 2>&1
echo rc=$? >> $O/time_8.log
grep -v "Gloo\|socket.cpp\|amdgpu.ids" $O/bench_8ranks_gloo.err | tail -c 1500; cut -c1-300 $O/bench_8ranks_gloo.json; cat $O/time_8.log

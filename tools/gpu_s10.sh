#!/bin/bash
# plumbing check of bench.py's N > 1 legs on the one-GPU box: two ranks on one device over gloo (numbers meaningless)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
STARDIST_AMD_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --no-split-leg --sharded-size 8192 --sharded-size3d 512 --sharded-block3d 304 > $O/s10_bench2.json 2> $O/s10_bench2.err
tail -5 $O/s10_bench2.err; cut -c1-600 $O/s10_bench2.json

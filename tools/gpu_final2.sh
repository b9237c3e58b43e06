#!/bin/bash
# re-validation after the convolution refactor + split-bf16 kernel (3D NMS parity file unchanged since the last full run: left out)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
( time timeout 300 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity3d.py 2>&1 | tail -6 ) > gpurun_out/final2_gputests.log 2>&1
( timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep '^{' ) > gpurun_out/final2_bench.json 2>gpurun_out/final2_bench.err
tail -3 $R/gpurun_out/final2_gputests.log; cut -c1-300 $R/gpurun_out/final2_bench.json

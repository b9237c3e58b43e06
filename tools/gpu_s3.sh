#!/bin/bash
# session 3 of round 4: transposed epilogue of the fp16 kernel, full suite, full-size sharded goldens + parity tests, bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
free -g > $O/s3_host.txt; nproc >> $O/s3_host.txt
timeout 120 python tools/probe_h2d.py > $O/s3_h2d.txt 2>&1
timeout 300 python tools/probe_hand_conv.py --reps 4 > $O/s3_layer_probe.txt 2>&1
( time timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bigparity.py 2>&1 | tail -15 ) > $O/s3_tests.log 2>&1
( time timeout 1200 python tests/golden/make_sharded_golden.py 2d 3d ) > $O/s3_golden.log 2>&1
cp $O/sharded_fullsize.json tests/golden/sharded_fullsize.json 2>/dev/null
( time timeout 600 python -m pytest tests/test_gpu_bigparity.py -m gpu -q -s 2>&1 | tail -25 ) > $O/s3_bigparity.log 2>&1
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 > $O/s3_bench.json 2> $O/s3_bench.err
cat $O/s3_host.txt $O/s3_h2d.txt; tail -3 $O/s3_layer_probe.txt; tail -8 $O/s3_tests.log; tail -12 $O/s3_golden.log; tail -12 $O/s3_bigparity.log; tail -3 $O/s3_bench.err; cut -c1-300 $O/s3_bench.json

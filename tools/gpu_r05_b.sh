#!/bin/bash
# round 5: quick validation of the host-glue changes (end-to-end parity files + a short bench)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 400 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_fullsize.py tests/test_gpu_cli_multiclass.py -m gpu -q -x 2>&1 | tail -6 ) > $O/tests_subset.log 2>&1
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-sharded --no-split-leg --no-cpu-baseline > $O/bench_short.json 2> $O/bench_short.err
tail -4 $O/tests_subset.log; python -c "
import json; d=json.load(open('$O/bench_short.json'))
print(d['value'], d['ms_per_step'], d['stages_ms']); print(d['value_3d'], d['ms_per_step_3d'], d['stages_ms_3d']); print(d['value_host_input']['value'], d['value_host_input_3d']['value'], d['nms2d_strict']['value'], d['nms2d_strict']['same_result_as_default'])"

#!/bin/bash
# round 6: resnet-bn parity, bench with the structured stages
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06e; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 900 python -m pytest tests/test_gpu_unet_parity.py -m gpu -q -s -k "resnet3d or unet2d-bn or unet3d" 2>&1 | grep -E "vs float64|passed|failed|Error" | cut -c1-400 ) > $O/tests_unet.log 2>&1
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-sharded --no-split-leg > $O/bench_short.json 2> $O/bench_short.err
tail -12 $O/tests_unet.log; python - <<PY
import json
d=json.loads(open("$O/bench_short.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("value_3d"), d.get("ms_per_step_3d"))
print(json.dumps(d.get("stages_ms"))); print(json.dumps(d.get("stages_ms_3d")))
print(json.dumps(d.get("cpu_baseline",{}).get("stage_ratios"))); print(json.dumps(d.get("cpu_baseline_3d",{}).get("stage_ratios")))
PY
tail -3 $O/bench_short.err

#!/bin/bash
# round 6: widened decision band (NEAR_W 0.1875) -- 2D parity tests, split16 suite, short bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 1500 python -m pytest tests/test_gpu_split16.py tests/test_gpu_parity2d.py tests/test_gpu_beam_prep.py tests/test_gpu_fullsize_parity.py tests/test_gpu_glue.py -m gpu -q 2>&1 | tail -15 ) > $O/tests.log 2>&1
timeout 400 python bench.py --gpus 1 --steps 10 --warmup 3 --no-sharded --no-cpu-baseline > $O/bench_short.json 2> $O/bench_short.err
tail -5 $O/tests.log; python - <<PY
import json
d=json.loads(open("$O/bench_short.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("value_3d"), d.get("stages_ms"))
print({k:v for k,v in d.items() if "strict" in k or "pairs" in k})
PY

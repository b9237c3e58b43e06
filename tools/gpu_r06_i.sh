#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06i; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 900 python -m pytest tests/test_gpu_glue.py tests/test_gpu_fullsize_parity.py tests/test_gpu_parity2d.py -m gpu -q -x 2>&1 | tail -12 ) > $O/tests.log 2>&1
cat $O/tests.log
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) > $O/bench_time.log 2>&1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06i/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], json.dumps(d.get("stages_ms", d.get("config", {}))))
PY

#!/bin/bash
# round 6: the whole GPU suite at HEAD + short bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06f; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $O/tests_all.log 2>&1
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-sharded --no-split-leg --no-cpu-baseline > $O/bench_short.json 2> $O/bench_short.err
tail -12 $O/tests_all.log; python - <<PY
import json
d=json.loads(open("$O/bench_short.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("value_3d"), d.get("ms_per_step_3d"))
print(json.dumps(d.get("stages_ms"))[:600]); print(json.dumps(d.get("stages_ms_3d"))[:600])
PY
tail -3 $O/bench_short.err

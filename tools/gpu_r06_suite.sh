#!/bin/bash
# the full GPU suite + smoke at HEAD (the last step of the round)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06f; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 1300 python -m pytest tests -m gpu -q -x --durations=12 2>&1 | tail -24 ) > $O/tests.log 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
tail -8 $O/tests.log; tail -1 $O/smoke.log

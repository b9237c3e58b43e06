#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > $O/s19_tests.log 2>&1
tail -12 $O/s19_tests.log

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 400 python -m pytest tests/test_gpu_glue.py -m gpu -q -x 2>&1 | tail -15 ) > $O/glue_tests.log 2>&1
cat $O/glue_tests.log | cut -c1-250

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R; ulimit -c 0
timeout 500 python tools/contention_trace.py 8 25 > $O/contention_trace_p8.txt 2>&1; cat $O/contention_trace_p8.txt | cut -c1-200 | head -70

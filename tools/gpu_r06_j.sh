#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06j; mkdir -p $O; cd $R; ulimit -c 0
timeout 120 python tools/time_clip_latency.py 2>&1 | grep -v amdgpu.ids > $O/clip_latency.txt; cat $O/clip_latency.txt

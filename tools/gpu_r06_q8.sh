#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R; ulimit -c 0
SD_OPTS="probe_tier=77" timeout 500 python tools/contention_trace.py 8 25 > $O/contention_trace_p8_selfcheck.txt 2>&1; grep "^probe\|x  probe\|keep crc\|round 1" $O/contention_trace_p8_selfcheck.txt | cut -c1-330 | head -60

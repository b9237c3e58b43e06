#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R; ulimit -c 0
timeout 500 python tools/contention_trace.py 8 25 > $O/contention_trace_p8_ld.txt 2>&1; grep -c "" $O/contention_trace_p8_ld.txt; grep "keep crc\|round 1" $O/contention_trace_p8_ld.txt | cut -c1-160 | head -30
SD_OPTS="" timeout 500 python tools/contention_check.py 8 40 2d > "$O/contention7_2d_p8.txt" 2>&1; grep -v "^pid" "$O/contention7_2d_p8.txt" | grep "DIFF\|processes" | cut -c1-200; grep -c "^  " "$O/contention7_2d_p8.txt"

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R; ulimit -c 0
timeout 300 python tools/contention_band.py 8 10 > $O/contention_band_p8_readlane.txt 2>&1; tail -3 $O/contention_band_p8_readlane.txt | cut -c1-200
SD_OPTS="probe_tier=77" timeout 500 python tools/contention_trace.py 8 40 > $O/contention_trace_p8_selfcheck_readlane.txt 2>&1; grep "^probe round\|x  probe\|keep crc\|x  round 1" $O/contention_trace_p8_selfcheck_readlane.txt | cut -c1-260 | head -40

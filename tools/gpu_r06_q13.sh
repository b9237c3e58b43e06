#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R; ulimit -c 0
SD_LIB=$R/stardist_amd/csrc/libstardist_hip_probe.so SD_OPTS="probe_tier=77" timeout 500 python tools/contention_trace.py 8 40 > $O/contention_trace_p8_selfcheck_atomicstore.txt 2>&1
echo "rounds with a disagreement: $(grep '^probe round' $O/contention_trace_p8_selfcheck_atomicstore.txt | grep -c 'pairs with a different flag')"; grep "x  probe\|keep crc" $O/contention_trace_p8_selfcheck_atomicstore.txt | cut -c1-200
grep '^probe round' $O/contention_trace_p8_selfcheck_atomicstore.txt | grep 'pairs with a different flag' | cut -c1-330 | head -8

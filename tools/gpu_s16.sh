#!/bin/bash
# kernel trace of the 3D step alone (3 warm + 10 timed predict_instances) -> gpurun_out/s16_kt3d.md
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; ulimit -c 0
rm -rf /tmp/kt3; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt3 -o kt -- python $R/tools/pyprof_step.py ${1:-3d} > $O/s16_run.log 2>&1
db=$(find /tmp/kt3 -name '*.db' | head -1)
python $R/tools/rocpd_summary.py $db --md > $O/s16_kt${1:-3d}.md 2>&1
head -34 $O/s16_kt${1:-3d}.md | cut -c1-60,150-260

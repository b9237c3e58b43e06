#!/bin/bash
# device timeline of one 2D / 3D step (kernels + copies)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06tl; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for W in ${1:-2d}; do
  rm -rf /tmp/tl_$W
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_$W -o p -- python $R/tools/step_timeline.py run $W 2 > $O/timeline_run_$W.log 2>&1
  python $R/tools/step_timeline.py report /tmp/tl_$W $W > $O/step_timeline_$W.txt 2>&1
done
grep -n "k_build32\|k_neighbours\|k_poly_props\|k_prepare\|k_cell\|k_round_triage" $O/step_timeline_2d.txt | head -12; tail -30 $O/step_timeline_2d.txt | head -30

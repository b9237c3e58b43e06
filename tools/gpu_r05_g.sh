#!/bin/bash
# round 5, after "nms3d_bounds_lean": the driver's bench command, the 3D section timing / round trace / device timeline at HEAD
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05h; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/final_bench.json 2> $O/final_bench.err ) > $O/bench_time.log 2>&1
timeout 120 python tools/time_predict_sections.py > $O/sections.log 2>&1
SD_TRACE=1 timeout 100 python tools/time_nms3d_bench.py 2 > $O/nms3d_rounds_trace.txt 2>&1
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tl_3d
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_3d -o p -- python $R/tools/step_timeline.py run 3d 2 > $O/timeline_run_3d.log 2>&1
python $R/tools/step_timeline.py report /tmp/tl_3d 3d > $O/step_timeline_3d.txt 2>&1
cd $R; cut -c1-200 $O/final_bench.json; tail -3 $O/bench_time.log; grep "3D:\|nms_3d" $O/sections.log

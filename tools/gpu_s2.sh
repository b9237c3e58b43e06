#!/bin/bash
# session 2 of round 4: full GPU suite, smoke, bench, phase profile of the split-fp16 kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
( time timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > $O/s2_tests.log 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/s2_smoke.log 2>&1
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 > $O/s2_bench.json 2> $O/s2_bench.err
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Istardist_amd/csrc tools/conv_f16_phase_profile.hip -o /tmp/cpp16 2>/dev/null
( /tmp/cpp16 2; /tmp/cpp16 1 ) > $O/s2_conv_f16_phases.txt 2>&1
timeout 200 python tools/time_predict_sections.py > $O/s2_sections.log 2>&1
tail -6 $O/s2_tests.log; tail -1 $O/s2_smoke.log; tail -5 $O/s2_bench.err; cut -c1-1500 $O/s2_bench.json; head -40 $O/s2_conv_f16_phases.txt; cat $O/s2_sections.log

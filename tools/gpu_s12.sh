#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; ulimit -c 0
rm -f $O/sharded_fullsize.json
( time timeout 1500 python tests/golden/make_sharded_golden.py 2d 3d ) > $O/s12_golden.log 2>&1
cp $O/sharded_fullsize.json tests/golden/sharded_fullsize.json
( time timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/s12_tests.log 2>&1
grep -E "^2D|^3D|host" $O/s12_golden.log; tail -5 $O/s12_tests.log

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06h; mkdir -p $O; cd $R; ulimit -c 0
timeout 300 python -m pytest -s -q tests/test_gpu_lattice.py -m gpu -k "cartesian_pair or shifted_copies" 2>&1 | grep -v "^QH\|^$\|qhull\|Qhull\|precision\|^  \|^-\|^The\|^See\|^If\|^ERR\|While\|^Use\|^To\|flat\|^A \|^e\.g\|joggle\|^on\|^One\|^Is\|^produce\|^last\|^Or\|^Options" | tail -30 > $O/cartesian_pair_volumes.txt; cat $O/cartesian_pair_volumes.txt
( time timeout 1200 python -m pytest tests/test_gpu_lattice.py tests/test_gpu_parity3d.py tests/test_gpu_relabel.py tests/test_gpu_fullsize_parity.py -m gpu -q -x 2>&1 | tail -6 ) > $O/tests.log 2>&1
cat $O/tests.log

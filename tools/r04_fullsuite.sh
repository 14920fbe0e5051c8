#!/bin/bash
# the whole -m gpu suite and smoke(), as the driver runs them at round end
T=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) | tee $O/smoke.log
(timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=12 2>&1 | tail -30) | tee $O/pytest_gpu_full.log

#!/bin/bash
# Round 3 GPU session: the -N 1 seed cache model and the interleaved seed policies on the device (+ the option-surface and paired suites).
#   gpurun --timeout 1200 -- 'bash tools/r03_session7.sh TAG'
T=${1:-r03q}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(timeout 900 python -m pytest -q -x -m gpu tests/test_seed_cache_model.py tests/test_cli_options.py tests/test_paired.py tests/test_gpu_align.py 2>&1 | tail -15) | tee $O/pytest.log

#!/bin/bash
# the unpaired command-line tests on the device with the final build (the last two GPU-minutes of the round)
T=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(timeout 95 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_cli_options.py tests/test_zz_effort_knobs.py tests/test_seed_cache_model.py tests/test_zz_bam_input.py tests/test_zz_mixed_inputs.py 2>&1 | tail -5) | tee $O/pytest.log

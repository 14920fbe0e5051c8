#!/bin/bash
T=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_stages.py tests/test_gpu_align.py 2>&1 | tail -8) | tee $O/pytest.log
(timeout 500 python bench.py --config local400 --steps 4 --warmup 2 --parity-only 2>$O/bench_local.err | tail -1) > $O/bench_local.json; tail -1 $O/bench_local.err | cut -c1-200
python3 -c "
import json
d=json.load(open('$O/bench_local.json')); c=d['config']; print('local400', round(d['value']), 'reads/s', c['kernel_ms_per_step'], 'depth', c.get('steps_in_flight'), 'parity', c.get('parity_identical'), c.get('parity_differing_sam_lines'), 'flagged', c.get('reads_overflowed')); print(c['worker_phase_us_per_read_profiled_pass']); print(c['backtrace_profile_per_read'], c['worker_counts_per_read'])"

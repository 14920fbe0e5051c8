#!/bin/bash
# stage tests (incl. the one-mm stage and the every-row check of the device layout) + the headline bench with the SAM comparison
T=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(timeout 1200 python -m pytest -q -x -m gpu tests/test_gpu_stages.py tests/test_rank_index.py tests/test_gpu_align.py 2>&1 | tail -8) | tee $O/pytest.log
(timeout 600 python bench.py --steps 4 --warmup 2 --parity-only 2>$O/bench_se150.err | tail -1) > $O/bench_se150.json; tail -1 $O/bench_se150.err | cut -c1-200
python3 -c "
import json
d=json.load(open('$O/bench_se150.json')); c=d['config']; print('se150', round(d['value']), 'reads/s', c['kernel_ms_per_step'], 'parity', c.get('parity_identical'), c.get('parity_differing_sam_lines'), 'flagged', c.get('reads_overflowed')); print(c['worker_phase_us_per_read_profiled_pass']); print(d['roofline']['fm_kernels'])"

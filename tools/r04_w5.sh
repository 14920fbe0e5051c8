#!/bin/bash
# the 5-waves-per-SIMD class of the worker: unpaired end-to-end tests on it, then the headline line with and without it
T=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(timeout 1200 python -m pytest -q -x -m gpu tests/test_gpu_align.py tests/test_row_sampler.py tests/test_work_counters.py tests/test_all_hits_limit.py tests/test_overhang.py tests/test_gpu_scale.py -k "not paired" 2>&1 | tail -6) | tee $O/pytest.log
for v in w5 w4; do
  if [ $v = w4 ]; then export BT2G_NO_W5=1; else unset BT2G_NO_W5; fi
  (timeout 400 python bench.py --steps 6 --warmup 2 --parity-only 2>$O/bench_$v.err | tail -1) > $O/bench_$v.json
  python3 -c "
import json
d=json.load(open('$O/bench_$v.json')); c=d['config']; print('$v', round(d['value']), 'reads/s', c['kernel_ms_per_step'], 'parity', c.get('parity_identical'), c.get('parity_differing_sam_lines'), 'flagged', c.get('reads_overflowed'))"
done
(timeout 300 python bench.py --config ecoli100 --steps 6 --warmup 2 --parity-only 2>$O/bench_ecoli.err | tail -1) > $O/bench_ecoli.json
python3 -c "
import json
d=json.load(open('$O/bench_ecoli.json')); c=d['config']; print('ecoli100 w4 (env still set):', round(d['value']), c['kernel_ms_per_step'], c.get('parity_identical'))"
unset BT2G_NO_W5
(timeout 300 python bench.py --config ecoli100 --steps 6 --warmup 2 --parity-only 2>$O/bench_ecoli5.err | tail -1) > $O/bench_ecoli5.json
python3 -c "
import json
d=json.load(open('$O/bench_ecoli5.json')); c=d['config']; print('ecoli100 w5:', round(d['value']), c['kernel_ms_per_step'], c.get('parity_identical'))"

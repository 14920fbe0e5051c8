#!/bin/bash
# stage tests + worker tests + pairs (incl. wide opposite-mate windows) + the lambda example + headline and paired bench lines
T=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(timeout 1500 python -m pytest -q -x -m gpu tests/test_gpu_stages.py tests/test_rank_index.py tests/test_gpu_align.py tests/test_paired.py tests/test_gpu_scale.py -k "not repeat_rich" 2>&1 | tail -8) | tee $O/pytest.log
(timeout 600 python bench.py --steps 4 --warmup 2 --parity-only 2>$O/bench_se150.err | tail -1) > $O/bench_se150.json; tail -1 $O/bench_se150.err | cut -c1-200
(timeout 600 python bench.py --config pe-vsens --steps 4 --warmup 2 --no-cpu-baseline 2>$O/bench_pe.err | tail -1) > $O/bench_pe.json; tail -1 $O/bench_pe.err | cut -c1-200
python3 -c "
import json
for f in ('se150','pe'):
    d=json.load(open('$O/bench_%s.json' % f)); c=d['config']; print(f, round(d['value']), 'reads/s', c['kernel_ms_per_step'], 'parity', c.get('parity_identical'), c.get('parity_differing_sam_lines'), 'flagged', c.get('reads_overflowed'))"

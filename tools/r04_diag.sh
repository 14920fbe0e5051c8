#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R; export TMPDIR=/tmp
(BT2G_LIB=$R/build/variants/libbt2g_diag.so timeout 300 python tools/r04_diag.py pe-vsens 400000 2>&1 | grep -v "^\[\|amdgpu.ids" | tee $O/pe.txt) | cut -c1-330
(timeout 500 python bench.py --config pe-vsens --steps 9 --warmup 3 --parity-only 2>$O/bench_pe.err | tail -1) > $O/bench_pe.json
python3 -c "
import json
d=json.load(open('$O/bench_pe.json')); c=d['config']; print('pe-vsens', round(d['value']), 'reads/s', c['kernel_ms_per_step'], 'depth', c.get('steps_in_flight'), 'parity', c.get('parity_identical'), c.get('parity_differing_sam_lines'), 'flagged', c.get('reads_overflowed'))"

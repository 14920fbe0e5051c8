#!/bin/bash
T=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
run() { # name args...
  local nm=$1; shift
  (timeout 400 python bench.py "$@" 2>$O/$nm.err | tail -1) > $O/$nm.json; tail -1 $O/$nm.err | cut -c1-200
  python3 -c "
import json
try:
    d=json.load(open('$O/$nm.json')); c=d['config']; print('$nm', round(d['value']), 'reads/s', 'ms/step', round(d['ms_per_step'],1), c['kernel_ms_per_step'], 'depth', c.get('steps_in_flight'), 'parity', c.get('parity_identical'), 'flagged', c.get('reads_overflowed'))
except Exception as e: print('$nm', 'no result', e)"
}
run pe_p3 --config pe-vsens --steps 9 --warmup 3 --pipeline 3 --no-cpu-baseline
run local_p3 --config local400 --steps 6 --warmup 3 --pipeline 3 --no-cpu-baseline

#!/bin/bash
# headline config with 1, 2 and 3 launches in flight
T=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
for d in 1 2 3; do
  (timeout 300 python bench.py --steps 9 --warmup 3 --pipeline $d --no-cpu-baseline 2>$O/b_$d.err | tail -1) > $O/b_$d.json
  python3 -c "
import json
d=json.load(open('$O/b_$d.json')); c=d['config']; print('in flight $d:', round(d['value']), 'reads/s', round(d['ms_per_step'],2), 'ms/step', c['kernel_ms_per_step'])" | tee -a $O/summary.txt
done

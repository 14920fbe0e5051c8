#!/bin/bash
T=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
for n in 400000 1600000; do
  (timeout 300 python bench.py --config pe-vsens --reads $n --steps 2 --warmup 1 --no-cpu-baseline 2>$O/pe_$n.err | tail -1) > $O/pe_$n.json
  python3 -c "
import json
d=json.load(open('$O/pe_$n.json')); c=d['config']; print('pe reads', $n, round(d['value']), c['kernel_ms_per_step'], c['worker_phase_us_per_read_profiled_pass']['whole_read'])"
done
for n in 200000 800000; do
  (timeout 300 python bench.py --config local400 --reads $n --steps 2 --warmup 1 --no-cpu-baseline 2>$O/lo_$n.err | tail -1) > $O/lo_$n.json
  python3 -c "
import json
d=json.load(open('$O/lo_$n.json')); c=d['config']; print('local reads', $n, round(d['value']), c['kernel_ms_per_step'], c['worker_phase_us_per_read_profiled_pass']['whole_read'])"
done

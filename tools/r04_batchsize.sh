#!/bin/bash
# device-side throughput of the headline workload at the batch sizes the driver uses, three launches in flight (as its three device threads keep)
T=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
for cfg in "262144 48 3" "524288 24 3" "1048576 12 3" "262144 48 1"; do
  set -- $cfg
  (timeout 300 python bench.py --reads $1 --steps $2 --warmup 3 --pipeline $3 --no-cpu-baseline 2>$O/b_$1_$3.err | tail -1) > $O/b_$1_$3.json
  python3 -c "
import json
d=json.load(open('$O/b_$1_$3.json')); c=d['config']; print('reads/launch $1 in flight $3:', round(d['value']), 'reads/s', d['ms_per_step'], 'ms/step', c['kernel_ms_per_step'])" | tee -a $O/summary.txt
done

#!/bin/bash
T=${1:-r04e}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
for m in 1040 1296 1552 784; do
  (BT2G_DBG_EXT=$m timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> $O/m$m.err | tail -1) > $O/m$m.json
  python3 -c "
import json
c=json.load(open('$O/m$m.json'))['config']; print('mode $m', c['kernel_ms_per_step'])"
done

#!/bin/bash
# bench.py against builds of libbt2g.so under build/variants (tools/build_variant.sh): tools/r04_variants.sh TAG v1 v2 ...
T=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "cur" ]; then unset BT2G_LIB; else export BT2G_LIB=$R/build/variants/libbt2g_$v.so; fi
  (timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2> $O/var_$v.err | tail -1) > $O/var_$v.json
  python3 - <<P
import json
try:
    j = json.load(open("$O/var_$v.json")); c = j["config"]
    print("$v", "reads/s %.0f" % j["value"], c["kernel_ms_per_step"], "flagged", c["reads_overflowed"], "aligned", c["fraction_aligned"], "whole_read", c["worker_phase_us_per_read_profiled_pass"]["whole_read"])
except Exception as e:
    print("$v", "no result", e)
P
done

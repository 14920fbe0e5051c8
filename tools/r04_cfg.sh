#!/bin/bash
# the other configurations, parity-only: tools/r04_cfg.sh TAG cfg1 cfg2 ...
T=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
for C in "$@"; do
  (timeout 500 python bench.py --config $C --steps 3 --warmup 1 --parity-only 2>$O/bench_$C.err | tail -1) > $O/bench_$C.json; tail -1 $O/bench_$C.err
  python3 - <<P
import json
try:
    d = json.loads(open("$O/bench_$C.json").read()); c = d["config"]
    print("$C", round(d["value"]), "reads/s", c["kernel_ms_per_step"], "parity", c.get("parity_identical"), c.get("parity_differing_sam_lines"), "flagged", c.get("reads_overflowed"), "aligned", c.get("fraction_aligned"), "cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"]))
    print("  ", c["worker_phase_us_per_read_profiled_pass"])
except Exception as e:
    print("$C", "no result", e)
P
done

#!/bin/bash
# headline bench with the in-run 1 M-read SAM comparison (reference run once): the quick check after a worker change
T=${1:-r03r}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(timeout 400 python bench.py --steps 5 --warmup 2 --parity-only 2>$O/bench.err | tail -1) > $O/bench.json; tail -2 $O/bench.err
python - <<P
import json
d = json.loads(open("$O/bench.json").read()); c = d["config"]
print(round(d["value"]), "reads/s", c["kernel_ms_per_step"], "parity", c.get("parity_identical"), c.get("parity_differing_sam_lines"), "flagged", c.get("reads_overflowed"), "aligned", c.get("fraction_aligned"))
print(c["worker_phase_us_per_read_profiled_pass"])
P

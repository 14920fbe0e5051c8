#!/bin/bash
# Round 3: the other two BASELINE.json configurations on the device (parity against the reference in the same run).
#   gpurun --timeout 1800 -- 'bash tools/r03_configs.sh TAG [extra bench args]'
T=${1:-r03cfg}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
for C in pe-vsens local400; do
  (timeout 1200 python bench.py --config $C --steps 3 --warmup 1 "$@" 2>$O/bench_$C.err | tail -1) > $O/bench_$C.json; tail -3 $O/bench_$C.err
done
python - <<P
import json
for w in ("pe-vsens", "local400"):
    try:
        d = json.loads(open("$O/bench_%s.json" % w).read()); c = d["config"]
        print(w, round(d["value"]), "reads/s", c["kernel_ms_per_step"], "parity", c.get("parity_identical"), c.get("parity_differing_sam_lines"), "flagged", c.get("reads_overflowed"), "aligned", c.get("fraction_aligned"))
        print("  cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"]), c["worker_phase_us_per_read_profiled_pass"])
    except Exception as e:
        print(w, "no result:", e)
P

#!/bin/bash
# Round 3 GPU session: the whole -m gpu suite, then the paired configuration with the worker's phase profile.
#   gpurun --timeout 1800 -- 'bash tools/r03_session6.sh TAG'
T=${1:-r03n}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(timeout 1000 python -m pytest -q -m gpu tests 2>&1 | tail -15) | tee $O/pytest_gpu_full.log
(timeout 420 python bench.py --config pe-vsens --steps 3 --warmup 1 --parity-only 2>$O/bench_pe-vsens.err | tail -1) > $O/bench_pe-vsens.json; tail -2 $O/bench_pe-vsens.err
python - <<P
import json
for w in ("bench_pe-vsens",):
    try:
        d = json.loads(open("$O/%s.json" % w).read()); c = d["config"]
        print(w, round(d["value"]), "reads/s", c["kernel_ms_per_step"], "parity", c.get("parity_identical"), c.get("parity_differing_sam_lines"), "flagged", c.get("reads_overflowed"), "aligned", c.get("fraction_aligned"))
        print("  cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"]), c["worker_phase_us_per_read_profiled_pass"])
        print("  bt", c.get("backtrace_profile_per_read"), c.get("worker_counts_per_read"))
    except Exception as e:
        print(w, "no result:", e)
P

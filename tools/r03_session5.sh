#!/bin/bash
# Round 3 GPU session: whole-path parity tests (incl. --local and pairs), the headline line, --local on 400-bp reads.
#   gpurun --timeout 1500 -- 'bash tools/r03_session5.sh TAG'
T=${1:-r03l}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(timeout 700 python -m pytest -q -x -m gpu tests/test_gpu_align.py tests/test_paired.py tests/test_gpu_stages.py 2>&1 | tail -15) | tee $O/pytest.log
(timeout 400 python bench.py --steps 5 --warmup 2 --parity-only 2>$O/bench.err | tail -1) > $O/bench.json; tail -2 $O/bench.err
(timeout 420 python bench.py --config local400 --steps 3 --warmup 1 --parity-only 2>$O/bench_local400.err | tail -1) > $O/bench_local400.json; tail -2 $O/bench_local400.err
python - <<P
import json
for w in ("bench", "bench_local400"):
    try:
        d = json.loads(open("$O/%s.json" % w).read()); c = d["config"]
        print(w, round(d["value"]), "reads/s", c["kernel_ms_per_step"], "parity", c.get("parity_identical"), c.get("parity_differing_sam_lines"), "flagged", c.get("reads_overflowed"), "aligned", c.get("fraction_aligned"))
        print("  cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"]), c["worker_phase_us_per_read_profiled_pass"])
        print("  bt", c.get("backtrace_profile_per_read"))
    except Exception as e:
        print(w, "no result:", e)
P

#!/bin/bash
# quick device check after a worker change: optional pytest selection, then the headline line with the 1 M-read SAM comparison
#   gpurun --timeout 900 -- 'bash tools/r04_quick.sh TAG "tests/test_gpu_stages.py tests/test_gpu_align.py"'
T=${1:-r04q}; TESTS=$2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
if [ -n "$TESTS" ]; then (timeout 900 python -m pytest -q -x -m gpu $TESTS 2>&1 | tail -6) | tee $O/pytest.log; fi
(timeout 400 python bench.py --steps 5 --warmup 2 --parity-only 2>$O/bench.err | tail -1) > $O/bench.json; tail -2 $O/bench.err
python3 - <<P
import json
d = json.loads(open("$O/bench.json").read()); c = d["config"]
print("default", round(d["value"]), "reads/s", c["kernel_ms_per_step"], "parity", c.get("parity_identical"), c.get("parity_differing_sam_lines"), "flagged", c.get("reads_overflowed"), "aligned", c.get("fraction_aligned"))
print(c["worker_phase_us_per_read_profiled_pass"])
print(c["backtrace_profile_per_read"], c["worker_counts_per_read"], "dp_gcups", d["roofline"]["dp_gcups"], d["roofline"]["dp_cells_note"])
P

#!/bin/bash
# Round 4, session 2: the register-only row sampler on the device: its own parity test + the whole-path tests, then the headline line with
# the 1 M-read SAM comparison.
#   gpurun --timeout 1500 -- 'bash tools/r04_session2.sh TAG'
T=${1:-r04b}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(timeout 900 python -m pytest -q -x -m gpu tests/test_row_sampler.py tests/test_gpu_align.py tests/test_gpu_stages.py 2>&1 | tail -8) | tee $O/pytest.log
(timeout 400 python bench.py --steps 5 --warmup 2 --parity-only 2>$O/bench.err | tail -1) > $O/bench.json; tail -2 $O/bench.err
python3 - <<P
import json
d = json.loads(open("$O/bench.json").read()); c = d["config"]
print("default", round(d["value"]), "reads/s", c["kernel_ms_per_step"], "parity", c.get("parity_identical"), c.get("parity_differing_sam_lines"), "flagged", c.get("reads_overflowed"), "aligned", c.get("fraction_aligned"))
print(c["worker_phase_us_per_read_profiled_pass"])
P

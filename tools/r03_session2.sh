#!/bin/bash
# Round 3 GPU session: stage + whole-path parity tests on the device, then the headline bench (reference run once: SAM parity).
#   gpurun --timeout 1800 -- 'bash tools/r03_session2.sh TAG [pytest -k expression]'
T=${1:-r03b}; K=${2:-}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
if [ -n "$K" ]; then
  (timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_stages.py tests/test_gpu_align.py tests/test_gpu_pack.py -k "$K" 2>&1 | tail -15) | tee $O/pytest.log
else
  (timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_stages.py tests/test_gpu_align.py tests/test_gpu_pack.py 2>&1 | tail -15) | tee $O/pytest.log
fi
(timeout 900 python bench.py --steps 5 --warmup 2 --parity-only 2>$O/bench.err | tail -1) > $O/bench.json; tail -4 $O/bench.err
python - <<P
import json
d = json.loads(open("$O/bench.json").read()); c = d["config"]
print(round(d["value"]), "reads/s", c["kernel_ms_per_step"], "parity", c.get("parity_identical"), c.get("parity_differing_sam_lines"), "flagged", c.get("reads_overflowed"), "aligned", c.get("fraction_aligned"))
print(c["worker_phase_us_per_read_profiled_pass"])
print("roofline frac", d["roofline"]["frac"], "fm", d["roofline"]["fm_kernels"]["frac"], d["roofline"]["fm_kernels"]["ms_per_launch_sum"])
P

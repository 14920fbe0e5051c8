#!/bin/bash
# Rebuild the worker kernels with another register budget ON THE GPU BOX and check them: tools/wpe_try.sh TAG WPE
T=${1:-wpe}; W=${2:-4}
O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
cd $GRAFT_REPO_ROOT
touch bowtie2_amd/csrc/bt2g_align_kernel.hip
make -C bowtie2_amd/csrc WPE=$W > $O/make.log 2>&1 || { tail -5 $O/make.log; exit 1; }
(timeout 900 python -m pytest tests/test_gpu_align.py -x -q -m gpu -k "golden_sam or determinism" 2>&1 | tail -5) | tee $O/pytest.log
(timeout 900 python bench.py 2>$O/bench.err | tail -1) > $O/bench.json
python - <<P
import json
d=json.loads(open("$O/bench.json").read()); c=d["config"]
print("WPE=$W", round(d["value"]), c["kernel_ms_per_step"], "parity", c.get("parity_identical"), c.get("parity_differing_sam_lines"))
print(c["worker_phase_us_per_read_profiled_pass"])
P

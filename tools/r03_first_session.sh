#!/bin/bash
# First GPU session of the round after round 2 ("Plan for the next round's first GPU hour"), one gpurun call, ~25 GPU-minutes:
#   gpurun --timeout 2400 -- 'bash tools/r03_first_session.sh r03a'
# 1. the -m gpu twins of the tests added when round 2's GPU minutes were spent (BAM input, option variants, effort knobs, mixed inputs)
# 2. the default bench line (headline configuration, CPU baseline, 1 M-read SAM parity)
# 3. the occupancy probe: the worker kernels rebuilt for 5 and 6 waves per SIMD (102 / 85 VGPRs), short bench with the 1 M-read parity check each
T=${1:-r03a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(timeout 900 python -m pytest -q -m gpu tests/test_zz_bam_input.py tests/test_zz_effort_knobs.py tests/test_zz_mixed_inputs.py 2>&1 | tail -25) | tee $O/pytest_new_gpu_tests.log
(timeout 600 python bench.py 2>$O/bench.err | tail -1) > $O/bench_wpe4.json; cut -c1-400 $O/bench_wpe4.json
for W in 5 6; do
  touch bowtie2_amd/csrc/bt2g_align_kernel.hip
  make -C bowtie2_amd/csrc WPE=$W > $O/make_wpe$W.log 2>&1 || { tail -5 $O/make_wpe$W.log; continue; }
  (timeout 300 python -m pytest tests/test_gpu_align.py -x -q -m gpu -k "golden_sam or determinism" 2>&1 | tail -3) | tee $O/pytest_wpe$W.log
  (timeout 600 python bench.py --steps 5 --warmup 2 2>$O/bench_wpe$W.err | tail -1) > $O/bench_wpe$W.json
done
python - <<P
import json
for w in (4, 5, 6):
    try:
        d = json.loads(open("$O/bench_wpe%d.json" % w).read()); c = d["config"]
        print("WPE=%d" % w, round(d["value"]), "reads/s", c["kernel_ms_per_step"], "parity", c.get("parity_identical"), "flagged", c.get("reads_overflowed"))
    except Exception as e:
        print("WPE=%d" % w, "no result:", e)
P

#!/bin/bash
# Round 3, first GPU session (one gpurun call):
#   gpurun --timeout 2400 -- 'bash tools/r03_session1.sh r03a'
# 1. the -m gpu twins of the tests added at the end of round 2 (never run on a device yet)
# 2. the headline bench line at hg38 scale (3 100 Mbp .bt2l, CPU baseline, 1 M-read SAM parity) with the corrected roofline accounting
# 3. occupancy probes of the worker kernel: 5 and 6 waves per SIMD at the shipped LDS footprint, and with the LDS footprint cut
#    (BT2G_PROBE_SMALL: capacities for 150-bp unpaired reads only) so that LDS admits 20 / 24 waves per CU
T=${1:-r03a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(timeout 900 python -m pytest -q -m gpu tests/test_zz_bam_input.py tests/test_zz_effort_knobs.py tests/test_zz_mixed_inputs.py 2>&1 | tail -25) | tee $O/pytest_new_gpu_tests.log
(timeout 900 python bench.py --steps 5 --warmup 2 2>$O/bench.err | tail -1) > $O/bench_wpe4.json; cut -c1-600 $O/bench_wpe4.json; tail -5 $O/bench.err
probe() {  # name, make arguments
  local n=$1; shift
  touch bowtie2_amd/csrc/bt2g_align_kernel.hip bowtie2_amd/csrc/bt2g_capi.hip bowtie2_amd/csrc/bt2g_kernels.hip
  make -C bowtie2_amd/csrc "$@" > $O/make_$n.log 2>&1 || { tail -5 $O/make_$n.log; return; }
  (timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>$O/bench_$n.err | tail -1) > $O/bench_$n.json
}
probe wpe5 WPE=5
probe wpe6 WPE=6
probe wpe5_small WPE=5 EXTRA=-DBT2G_PROBE_SMALL=32
probe wpe6_small WPE=6 EXTRA=-DBT2G_PROBE_SMALL=16
probe wpe4_small WPE=4 EXTRA=-DBT2G_PROBE_SMALL=16
python - <<P
import json
for w in ("wpe4", "wpe5", "wpe6", "wpe5_small", "wpe6_small", "wpe4_small"):
    try:
        d = json.loads(open("$O/bench_%s.json" % w).read()); c = d["config"]
        print(w, round(d["value"]), "reads/s", c["kernel_ms_per_step"], "parity", c.get("parity_identical"), "flagged", c.get("reads_overflowed"), "aligned", c.get("fraction_aligned"))
    except Exception as e:
        print(w, "no result:", e)
P

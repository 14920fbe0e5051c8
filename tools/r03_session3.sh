#!/bin/bash
# Round 3 GPU session: default bench (profile with the backtrace sub-timers) + occupancy probes built WITHOUT VGPR->AGPR spilling.
# (The first probes, profiles/r03a_occupancy_probes.txt, were not what they claimed: at a 96- or 80-register budget the compiler spilled
# into AGPRs, the unified allocation grew to 138-181 registers and the kernel ran at 2-3 waves per SIMD.)
T=${1:-r03d}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(timeout 900 python bench.py --steps 4 --warmup 1 --parity-only 2>$O/bench.err | tail -1) > $O/bench_wpe4.json; tail -2 $O/bench.err
probe() {  # name, make arguments
  local n=$1; shift
  touch bowtie2_amd/csrc/bt2g_align_kernel.hip bowtie2_amd/csrc/bt2g_capi.hip bowtie2_amd/csrc/bt2g_kernels.hip bowtie2_amd/csrc/bt2g_rankidx.hip
  make -C bowtie2_amd/csrc "$@" > $O/make_$n.log 2>&1 || { tail -5 $O/make_$n.log; return; }
  (timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>$O/bench_$n.err | tail -1) > $O/bench_$n.json
}
probe wpe5_small_noagpr WPE=5 "EXTRA=-DBT2G_PROBE_SMALL=16 -mllvm -amdgpu-spill-vgpr-to-agpr=0"
probe wpe6_small_noagpr WPE=6 "EXTRA=-DBT2G_PROBE_SMALL=16 -mllvm -amdgpu-spill-vgpr-to-agpr=0"
probe wpe4_small_noagpr WPE=4 "EXTRA=-DBT2G_PROBE_SMALL=16 -mllvm -amdgpu-spill-vgpr-to-agpr=0"
python - <<P
import json
for w in ("wpe4", "wpe4_small_noagpr", "wpe5_small_noagpr", "wpe6_small_noagpr"):
    try:
        d = json.loads(open("$O/bench_%s.json" % w).read()); c = d["config"]
        print(w, round(d["value"]), "reads/s", c["kernel_ms_per_step"], "parity", c.get("parity_identical"), "flagged", c.get("reads_overflowed"), "aligned", c.get("fraction_aligned"))
        if w == "wpe4": print(c["worker_phase_us_per_read_profiled_pass"]); print(c["backtrace_profile_per_read"], c["worker_counts_per_read"])
    except Exception as e:
        print(w, "no result:", e)
P

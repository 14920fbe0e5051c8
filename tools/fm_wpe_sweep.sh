#!/bin/bash
# Sweep the register budget of the 1-mismatch FM kernels on the GPU box: rebuild bt2g_kernels.o with each value, time the kernels.
T=${1:-sweep}; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
cd $GRAFT_REPO_ROOT
for w in "$@"; do   # (the sweep over all five kernels is recorded in profiles/r02_fm_wpe_sweep.txt; what is left to vary is the 1-mismatch pair)
  touch bowtie2_amd/csrc/bt2g_kernels.hip
  make -C bowtie2_amd/csrc EXTRA=-DBT2G_MM1_WPE=$w > $O/make_$w.log 2>&1 || { tail -5 $O/make_$w.log; continue; }
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_fmwpe$w.json
  python - <<P
import json
d=json.loads(open("$O/bench_fmwpe$w.json").read())
print("FMWPE=$w", round(d["value"]), d["config"]["kernel_ms_per_step"])
P
done

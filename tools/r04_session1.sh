#!/bin/bash
# Round 4, session 1: variants of the worker build side by side (round-3 build, leaf fills + inlined extend_seeds at 4 and 3 waves per
# SIMD), then the parity line and the traffic counters of the new default build.
#   gpurun --timeout 1200 -- 'bash tools/r04_session1.sh TAG'
T=${1:-r04a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
for v in base leaf leaf_w3; do
  export BT2G_LIB=$R/build/variants/libbt2g_$v.so
  (timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2> $O/var_$v.err | tail -1) > $O/var_$v.json
  python3 - <<P
import json
try:
    j = json.load(open("$O/var_$v.json")); c = j["config"]
    print("$v", "reads/s %.0f" % j["value"], c["kernel_ms_per_step"], "flagged", c["reads_overflowed"], "aligned", c["fraction_aligned"])
    print("   ", c["worker_phase_us_per_read_profiled_pass"])
except Exception as e:
    print("$v", "no result", e)
P
done
unset BT2G_LIB
(timeout 400 python bench.py --steps 5 --warmup 2 --parity-only 2>$O/bench.err | tail -1) > $O/bench.json; tail -2 $O/bench.err
python3 - <<P
import json
d = json.loads(open("$O/bench.json").read()); c = d["config"]
print("default", round(d["value"]), "reads/s", c["kernel_ms_per_step"], "parity", c.get("parity_identical"), c.get("parity_differing_sam_lines"), "flagged", c.get("reads_overflowed"), "aligned", c.get("fraction_aligned"))
P
P=$O/pmc; mkdir -p $P; cd /tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --reads 200000 --no-cpu-baseline"
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/b -- $CMD > $P/b.json 2> $P/b.err
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/c -- $CMD > $P/c.json 2> $P/c.err
timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $P/a -- $CMD > $P/a.json 2> $P/a.err
cd $R
python tools/pmc_summary.py $P 200000 2 > $P/pmc_summary.txt 2>&1; grep "k_align_reads" $P/pmc_summary.txt | cut -c1-200
find $P -name "*.csv" -size +1M -delete

#!/bin/bash
# Round 4, final evidence on the final code, per configuration: rocprofv3 kernel trace, the PMC passes (each in its own run), the bench line
# with the reference timed beside it.   gpurun --timeout 1500 -- 'bash tools/r04_final.sh TAG "se150:2000000:full ecoli100:200000:full"'
#   item = config : reads per launch of the PMC passes (0 = no PMC passes) : full | parity (cpu_baseline from two reference runs | one)
T=${1:-r04zz}; ITEMS=$2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
for item in $ITEMS; do
  C=${item%%:*}; rest=${item#*:}; N=${rest%%:*}; MODE=${rest#*:}
  P=$O/$C; mkdir -p $P
  if [ "$N" != "0" ]; then
    cd /tmp
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python $R/bench.py --config $C --steps 3 --warmup 1 --no-cpu-baseline > $P/bench_under_rocprof.json 2> $P/rocprof.err
    f=$(find $P/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $P/kernel_stats.csv && head -7 "$f" | cut -c1-150
    find $P/trace -name "*.csv" -size +1M -delete
    CMD="python $R/bench.py --config $C --steps 1 --warmup 1 --reads $N --pipeline 1 --no-cpu-baseline"
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $P/a -- $CMD > $P/a.json 2> $P/a.err
    timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/b -- $CMD > $P/b.json 2> $P/b.err
    timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/c -- $CMD > $P/c.json 2> $P/c.err
    cd $R
    python tools/pmc_summary.py $P $N 2 > $P/pmc_summary.txt 2>&1; grep "k_align" $P/pmc_summary.txt | cut -c1-160 | head -12
    find $P -name "*.csv" -size +1M -delete
    if [ -s $P/pmc_traffic.json ]; then
      if [ $C = se150 ]; then cp $P/pmc_traffic.json $R/profiles/${T}_pmc_traffic.json; else cp $P/pmc_traffic.json $R/profiles/${T}_pmc_traffic_$C.json; fi
    fi
  fi
  EXTRA=""; [ "$MODE" = "parity" ] && EXTRA="--parity-only"
  STEPS="--steps 9 --warmup 3"; [ $C = se150 ] && STEPS="--steps 25 --warmup 5"
  (timeout 600 python bench.py --config $C $STEPS $EXTRA 2>$P/bench.err | tail -1) > $P/bench.json; tail -1 $P/bench.err | cut -c1-160
  python3 - <<PY
import json
try:
    d = json.loads(open("$P/bench.json").read()); c = d["config"]
    print("$C", round(d["value"]), "reads/s", c["kernel_ms_per_step"], "parity", c.get("parity_identical"), c.get("parity_differing_sam_lines"), "flagged", c.get("reads_overflowed"), "aligned", c.get("fraction_aligned"))
    print("  cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"]), c["worker_phase_us_per_read_profiled_pass"])
    r = d["roofline"]; print("  roofline", r["frac"], "traffic", r["traffic"], r.get("traffic_over_algorithmic"), "fm", r["fm_kernels"]["frac"], r["fm_kernels"].get("physical_frac"), r["instruction_issue"])
except Exception as e:
    print("$C", "no result:", e)
PY
done

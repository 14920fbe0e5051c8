#!/bin/bash
# Round 3, final evidence on the final code: smoke(), rocprofv3 kernel trace + the PMC passes (each in its own run) for the three
# configurations, then the three bench lines with the reference timed beside them.
#   gpurun --timeout 1500 -- 'bash tools/r03_final.sh TAG'
T=${1:-r03z}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) | tee $O/smoke.log
prof_cfg() {   # name, reads per launch for the PMC passes, extra bench args
  local C=$1 N=$2; shift 2
  local P=$O/$C; mkdir -p $P; cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python $R/bench.py --config $C --steps 3 --warmup 1 --no-cpu-baseline "$@" > $P/bench_under_rocprof.json 2> $P/rocprof.err
  f=$(find $P/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $P/kernel_stats.csv && head -8 "$f" | cut -c1-160
  find $P/trace -name "*.csv" -size +1M -delete
  local CMD="python $R/bench.py --config $C --steps 1 --warmup 1 --reads $N --no-cpu-baseline"
  timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $P/a -- $CMD > $P/a.json 2> $P/a.err
  timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/b -- $CMD > $P/b.json 2> $P/b.err
  timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/c -- $CMD > $P/c.json 2> $P/c.err
  cd $R
  python tools/pmc_summary.py $P $N 2 > $P/pmc_summary.txt 2>&1; head -3 $P/pmc_summary.txt | cut -c1-160
  find $P -name "*.csv" -size +1M -delete
  if [ -s $P/pmc_traffic.json ]; then
    if [ $C = se150 ]; then cp $P/pmc_traffic.json $R/profiles/${T}_pmc_traffic.json; else cp $P/pmc_traffic.json $R/profiles/${T}_pmc_traffic_$C.json; fi
  fi
}
prof_cfg se150 200000
prof_cfg pe-vsens 40000
prof_cfg local400 20000
(timeout 480 python bench.py --steps 25 --warmup 5 2>$O/bench_se150.err | tail -1) > $O/bench_se150.json; tail -2 $O/bench_se150.err
(timeout 300 python bench.py --config pe-vsens --steps 5 --warmup 1 2>$O/bench_pe-vsens.err | tail -1) > $O/bench_pe-vsens.json; tail -1 $O/bench_pe-vsens.err
(timeout 300 python bench.py --config local400 --steps 5 --warmup 1 2>$O/bench_local400.err | tail -1) > $O/bench_local400.json; tail -1 $O/bench_local400.err
python - <<P
import json
for w in ("bench_se150", "bench_pe-vsens", "bench_local400"):
    try:
        d = json.loads(open("$O/%s.json" % w).read()); c = d["config"]
        print(w, round(d["value"]), "reads/s", c["kernel_ms_per_step"], "parity", c.get("parity_identical"), c.get("parity_differing_sam_lines"), "flagged", c.get("reads_overflowed"), "aligned", c.get("fraction_aligned"))
        print("  cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"]), c["worker_phase_us_per_read_profiled_pass"])
        print("  roofline", d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"].get("traffic_over_algorithmic"), d["roofline"]["fm_kernels"]["frac"], d["roofline"]["instruction_issue"])
    except Exception as e:
        print(w, "no result:", e)
P

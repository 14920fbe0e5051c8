#!/bin/bash
# One GPU-box session:   gpurun --timeout N -- 'bash tools/gpu_session.sh TAG step [step ...]'      (results under gpurun_out/TAG/)
#   tests[:FILES]      pytest -m gpu (whole suite, or the listed files separated by commas)
#   smoke              __graft_entry__.smoke()
#   bench[:NAME[:ENV=V,ENV=V...]]   bench.py $BENCH_ARGS (default: --steps 6 --warmup 2 --parity-only) under the given environment -> bench_NAME.json
#   full[:NAME]        the default bench.py line (cpu_baseline median of 3, parity, e2e) -> bench_NAME.json
#   prof[:NAME]        rocprofv3 --kernel-trace --stats of a short bench -> kernel_stats_NAME.csv
#   pmc[:NAME]         the --pmc passes (each its own run, no tracing domains besides the kernel trace) -> pmc_NAME/
#   cfg:CONFIG         bench.py --config CONFIG --parity-only -> bench_CONFIG.json
T=${1:-r05}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
summ() { python3 - "$1" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]
    print(sys.argv[1].split("/")[-1], round(d["value"]), d["unit"], "ms/step", d["ms_per_step"], c.get("kernel_ms_per_step"), "parity", c.get("parity_identical"), c.get("parity_differing_sam_lines"),
          "flagged", c.get("reads_overflowed"), "frac", d.get("roofline", {}).get("frac"), "e2e", d.get("e2e"))
    print("   phases", c.get("worker_phase_us_per_read_profiled_pass")); print("   bt", c.get("backtrace_profile_per_read"))
except Exception as e: print(sys.argv[1], "unreadable:", e)
P
}
for step in "$@"; do
  IFS=: read -r what a b <<< "$step"
  case $what in
  tests) (timeout ${PYTEST_TIMEOUT:-1500} python -m pytest -x -q -m gpu $([ -n "$a" ] && echo "${a//,/ }" || echo tests) 2>&1 | tail -8) | tee $O/pytest_${b:-gpu}.log ;;
  smoke) (timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -4) | tee $O/smoke.log ;;
  bench) n=${a:-default}; (env ${b//,/ } timeout ${BENCH_TIMEOUT:-600} python bench.py ${BENCH_ARGS:---steps 6 --warmup 2 --parity-only} 2>$O/bench_$n.err | tail -1) > $O/bench_$n.json; summ $O/bench_$n.json | tee -a $O/summary.txt ;;
  full) n=${a:-full}; (timeout ${BENCH_TIMEOUT:-1500} python bench.py 2>$O/bench_$n.err | tail -1) > $O/bench_$n.json; summ $O/bench_$n.json | tee -a $O/summary.txt ;;
  cfg) (timeout ${BENCH_TIMEOUT:-900} python bench.py --config $a --parity-only ${BENCH_ARGS} 2>$O/bench_$a.err | tail -1) > $O/bench_$a.json; summ $O/bench_$a.json | tee -a $O/summary.txt ;;
  prof) n=${a:-default}; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $O/bench_under_rocprof_$n.json 2> $O/rocprof_$n.err
    cd $R; f=$(find $O/prof_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" && cp "$f" $O/kernel_stats_$n.csv; rm -rf $O/prof_$n ;;
  pmc) n=${a:-default}; P=$O/pmc_$n; mkdir -p $P; cd /tmp
    CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS}"
    timeout 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $P/a -- $CMD > $P/a.json 2> $P/a.err
    timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/b -- $CMD > $P/b.json 2> $P/b.err
    timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/c -- $CMD > $P/c.json 2> $P/c.err
    cd $R; python tools/pmc_summary.py $P ${PMC_READS:-2000000} 2 > $P/summary.txt; head -60 $P/summary.txt; find $P -name "*.csv" -size +2M -delete ;;
  esac
done

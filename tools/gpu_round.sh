#!/bin/bash
# One GPU-box session of a round: usage  tools/gpu_round.sh TAG [tests|bench|prof|pmc|pe ...]
#   tests  pytest -m gpu (optionally only $PYTEST_ARGS)        bench  default bench.py (headline config, CPU baseline + parity)
#   prof   rocprofv3 --kernel-trace --stats of a short bench    pmc    the --pmc passes (each in its own run)
T=${1:-r02}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
export TMPDIR=/tmp
for what in "$@"; do
case $what in
tests)
  (timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -x -q -m gpu ${PYTEST_ARGS} 2>&1 | tail -15) | tee $O/pytest_gpu.log ;;
smoke)
  (timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -4) | tee $O/smoke.log ;;
bench)
  (timeout ${BENCH_TIMEOUT:-900} python bench.py ${BENCH_ARGS} 2>$O/bench.err | tail -1) > $O/bench.json; tail -12 $O/bench.err; cut -c1-1500 $O/bench.json ;;
prof)
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $O/bench_under_rocprof.json 2> $O/rocprof.err
  cd $R; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" && cp "$f" $O/kernel_stats.csv
  find $O/prof -name "*kernel_trace.csv" -size +5M -delete ;;
pmc)
  P=$O/pmc; mkdir -p $P; cd /tmp
  CMD="python $R/bench.py --steps 1 --warmup 1 --reads 200000 --no-cpu-baseline ${BENCH_ARGS}"
  timeout 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $P/a -- $CMD > $P/a.json 2> $P/a.err
  timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/b -- $CMD > $P/b.json 2> $P/b.err
  timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/c -- $CMD > $P/c.json 2> $P/c.err
  cd $R
  python tools/pmc_summary.py $P 200000 2 > $P/summary.txt; cat $P/summary.txt | head -60
  find $P -name "*.csv" -size +2M -delete ;;
pe)
  (timeout 600 python bench.py --paired --reads 400000 --steps 3 --warmup 1 ${BENCH_ARGS} 2>$O/bench_pe.err | tail -1) > $O/bench_pe.json; cut -c1-800 $O/bench_pe.json ;;
esac
done

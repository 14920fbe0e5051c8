#!/bin/bash
# end-of-round evidence: smoke, default bench (with cpu baseline), rocprofv3 kernel trace of the same bench command
T=${1:-r01e}
mkdir -p gpurun_out/$T
cd $GRAFT_REPO_ROOT
(timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) | tee gpurun_out/$T/smoke.log
(timeout 300 python bench.py 2>gpurun_out/$T/bench.err | tail -1) > gpurun_out/$T/bench.json; cut -c1-700 gpurun_out/$T/bench.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/prof -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$T/bench_under_rocprof.json 2> $R/gpurun_out/$T/rocprof.err
cd $R
f=$(find gpurun_out/$T/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
find gpurun_out/$T/prof -name "*kernel_trace.csv" -size +5M -delete

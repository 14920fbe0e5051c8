#!/bin/bash
# full GPU check: tests, smoke, bench (+cpu baseline), rocprofv3 kernel trace
mkdir -p gpurun_out/r01c
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5) | tee gpurun_out/r01c/pytest_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) | tee gpurun_out/r01c/smoke.log
(timeout 600 python bench.py 2>gpurun_out/r01c/bench.err | tail -1) > gpurun_out/r01c/bench.json; cat gpurun_out/r01c/bench.json | cut -c1-600
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01c/prof -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01c/bench_under_rocprof.json 2> $R/gpurun_out/r01c/rocprof.err
cd $R; find gpurun_out/r01c/prof -name "*kernel_stats*" | head -3
f=$(find gpurun_out/r01c/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
find gpurun_out/r01c/prof -name "*kernel_trace.csv" -size +5M -delete

#!/bin/bash
# rocprofv3 kernel trace of a short headline run: per-kernel times
T=${1:-r04t}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/bench_under_rocprof.json 2> $O/rocprof.err
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
find $O/trace -name "*.csv" -size +1M -delete
python3 - <<P
import csv
for r in csv.DictReader(open("$O/kernel_stats.csv")):
    n = r["Name"]
    if "bt2g::k_" in n and "build" not in n:
        print(n.split("(")[0][:70], r["Calls"], round(float(r["AverageNs"]) / 1e6, 3))
P

#!/bin/bash
# Where do k_align_reads' wave cycles go?  Two counter passes of a short bench (each its own run, kernel trace only):
#   gpurun -- 'bash tools/pmc_stall_probe.sh TAG'   -> gpurun_out/TAG/stall_{a,b}.txt
T=${1:-stall}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC\?_[A-Z_0-9]*" | sort -u > $O/counters_sq.txt
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS}"
pass() { n=$1; shift
  timeout 500 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/p_$n -- $CMD > $O/p_$n.json 2> $O/p_$n.err
  python3 - $O/p_$n "$@" > $O/stall_$n.txt <<'P'
import csv, glob, sys, collections
d = sys.argv[1]; acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
for k, v in acc.items():
    if "align" in k or "one_mm_scan" in k: print(k, len(nd[k]), {a: b / len(nd[k]) for a, b in v.items()})
P
  find $O/p_$n -name "*.csv" -size +1M -delete
}
pass a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_WAIT_INST_LDS
pass b SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES
pass c SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM
cat $O/stall_a.txt $O/stall_b.txt $O/stall_c.txt

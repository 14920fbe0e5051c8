T=r06ab; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|TA_[A-Z_0-9]*" | sort -u > $O/counters_tcc.txt
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --pipeline 1"
pass() { n=$1; shift
  timeout 500 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/p_$n -- $CMD > $O/p_$n.json 2> $O/p_$n.err
  python3 - $O/p_$n > $O/tcc_$n.txt <<'P'
import csv, glob, sys, collections
d = sys.argv[1]; acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
for k, v in acc.items():
    if "align" in k or "one_mm_scan" in k or "seed_search" in k: print(k, len(nd[k]), {a: round(b / len(nd[k]) / 2e6, 2) for a, b in v.items()})
P
  find $O/p_$n -name "*.csv" -size +1M -delete
}
pass a TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum
pass b TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_ATOMIC_sum
pass c TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum
cat $O/tcc_*.txt

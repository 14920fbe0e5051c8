#!/bin/bash
# Is the LDS what a CU's waves queue for?  (k_align_reads keeps its control state in LDS and reads / writes it from all 64 lanes at one address.)
#   gpurun -- 'bash tools/pmc_lds_probe.sh TAG [bench args]'   -> gpurun_out/TAG/lds.txt (per read)
T=${1:-lds}; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --pipeline 1 --e2e-reads 0 $*"
pass() { n=$1; shift
  timeout 500 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/p_$n -- $CMD > $O/p_$n.json 2> $O/p_$n.err
  python3 - $O/p_$n >> $O/lds.txt <<'P'
import csv, glob, sys, collections
d = sys.argv[1]; acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
for k, v in acc.items():
    if "k_align" in k: print(k, len(nd[k]), {a: round(b / len(nd[k]) / 2e6, 1) for a, b in v.items()})
P
  find $O/p_$n -name "*.csv" -size +1M -delete
}
rm -f $O/lds.txt
pass a SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC
pass b SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES
cat $O/lds.txt

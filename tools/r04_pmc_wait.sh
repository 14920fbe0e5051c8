#!/bin/bash
# where the waves of the worker kernel spend their cycles: issue / wait counters of the SQ (one more --pmc pass, its own run)
T=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --reads 2000000 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $O/d -- $CMD > $O/d.json 2> $O/d.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VALU --output-format csv -d $O/e -- $CMD > $O/e.json 2> $O/e.err
cd $R
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); nd = collections.defaultdict(set)
for f in glob.glob("$O/[de]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_align_reads" not in k: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); nd[r["Counter_Name"]].add(r["Dispatch_Id"])
with open("$O/wait_summary.csv", "w") as out:
    out.write("counter,dispatches,sum,per_read\n")
    for c, v in sorted(agg.items()):
        line = "%s,%d,%.0f,%.1f" % (c, len(nd[c]), v, v / (2000000.0 * max(1, len(nd[c]))))
        out.write(line + "\n"); print(line)
PY
find $O -name "*.csv" -size +1M -delete
tail -3 $O/d.err $O/e.err | cut -c1-200

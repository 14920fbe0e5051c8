#!/bin/bash
# PC sampling of the worker kernel (rocprofv3 host-trap sampling, beta): where the instruction stream of k_align_reads spends its time.
#   gpurun --timeout 600 -- 'bash tools/r04_pcsamp.sh TAG [bench args]'
T=${1:-r04p}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method ${PCS_METHOD:-stochastic} --pc-sampling-unit ${PCS_UNIT:-cycles} --pc-sampling-interval ${PCS_INTERVAL:-4194304} --kernel-trace --output-format csv -d $O/pcs -- \
  python $R/bench.py --steps 2 --warmup 1 --reads 500000 --no-cpu-baseline "$@" > $O/bench.json 2> $O/rocprof.err
echo "rc=$?"; tail -3 $O/rocprof.err
find $O/pcs -type f | head -20; find $O/pcs -type f -name "*.csv" -exec ls -la {} \;
f=$(find $O/pcs -name "*pc_sampling*.csv" | head -1)
if [ -n "$f" ]; then
  head -3 "$f"
  python3 - "$f" "$O" <<'P'
import csv, sys, collections, gzip
f, O = sys.argv[1], sys.argv[2]
cnt = collections.Counter(); n = 0
rd = csv.DictReader(open(f))
cols = rd.fieldnames
print(cols)
for r in rd:
    n += 1
    cnt[(r.get("Dispatch_Id", ""), r.get("Instruction", ""), r.get("Instruction_Comment", ""))] += 1
print("samples", n)
with open(O + "/pc_hist.csv", "w") as out:
    out.write("count,dispatch,instruction,comment\n")
    for (d, i, c), v in cnt.most_common():
        out.write("%d,%s,\"%s\",\"%s\"\n" % (v, d, i, c))
P
  gzip -c "$f" > $O/pc_samples.csv.gz; ls -la $O/pc_samples.csv.gz
  [ $(stat -c %s $O/pc_samples.csv.gz) -gt 40000000 ] && rm $O/pc_samples.csv.gz
fi
f2=$(find $O/pcs -name "*kernel_trace.csv" | head -1); [ -n "$f2" ] && cp "$f2" $O/kernel_trace.csv
find $O/pcs -name "*.csv" -size +1M -delete

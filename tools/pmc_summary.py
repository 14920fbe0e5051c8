#!/usr/bin/env python3
"""Sum the rocprofv3 --pmc passes of tools/gpu_round.sh per kernel; writes <dir>/summary.csv and <dir>/pmc_traffic.json
(the file bench.py's roofline.traffic reads once it is copied to profiles/).  usage: pmc_summary.py DIR READS_PER_LAUNCH LAUNCHES"""
import collections, csv, glob, json, os, sys
O, reads, launches = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
agg = collections.defaultdict(float); nd = collections.defaultdict(set)
for p in "abcd":
    for f in glob.glob(O + "/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "bt2g" not in k:
                continue
            agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); nd[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
with open(O + "/summary.csv", "w") as out:
    out.write("kernel,counter,dispatches,sum,per_read\n")
    for (k, c), v in sorted(agg.items()):
        out.write("%s,%s,%d,%.0f,%.2f\n" % (k, c, len(nd[(k, c)]), v, v / (reads * max(1, len(nd[(k, c)])))))
kern = collections.defaultdict(dict)
for (k, c), v in agg.items():
    short = k.split("<")[0].split("::")[-1]
    kern[short][c] = kern[short].get(c, 0) + v
    kern[short]["dispatches_" + c] = len(nd[(k, c)])
launches = max([len(v) for (k, c), v in nd.items() if "k_align" in k] or [launches])      # dispatches actually profiled
import hashlib
try:
    sha = hashlib.sha256(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bowtie2_amd", "libbt2g.so"), "rb").read()).hexdigest()
except OSError:
    sha = None
json.dump({"lib_sha256": sha, "lib_sha256_note": "the libbt2g.so these passes ran on: bench.py quotes this file only when it loads the same library",
           "reads_per_launch": reads, "launches": launches, "unit": "FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them", "kernels": kern},
          open(O + "/pmc_traffic.json", "w"), indent=1)
print(open(O + "/summary.csv").read())

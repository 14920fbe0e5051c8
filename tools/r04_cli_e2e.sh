#!/bin/bash
# End to end through the drop-in binary (process start, index load + transcoding, FASTQ parse, device, SAM text out) on the headline
# workload, file to file: 8 M and 24 M of the bench's reads.  `-t` prints the wall time of the search after the index load, the rate that follows from
# it, and the seconds each host stage's thread / the device stage's threads spent (the stages overlap).
#   gpurun --timeout 1200 -- 'bash tools/r04_cli_e2e.sh TAG'
T=${1:-r04e}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(timeout 300 python bench.py --steps 4 --warmup 2 --parity-only 2>$O/bench.err | tail -1) > $O/bench.json
C=/tmp/bt2_amd_bench
cat $C/sample.fq $C/sample.fq $C/sample.fq $C/sample.fq > /tmp/e2e_4m.fq
cat /tmp/e2e_4m.fq /tmp/e2e_4m.fq > /tmp/e2e_8m.fq; rm /tmp/e2e_4m.fq
cat /tmp/e2e_8m.fq /tmp/e2e_8m.fq /tmp/e2e_8m.fq > /tmp/e2e_24m.fq
python3 - "$O" <<'P' | tee $O/e2e.txt
import json, subprocess, sys, time, re
O = sys.argv[1]
res = json.load(open(O + "/bench.json"))["value"]
print("resident-batch rate (bench.py, 2 M reads per launch): %d reads/s" % res)
B = "/tmp/bt2_amd_bench/hg38like_3100mbp_s2_bt2l"
for sz, extra in (("8m", ["-S", "/dev/null"]), ("8m", ["-S", "/dev/null"]), ("24m", ["-S", "/dev/null"]), ("24m", ["-S", "/tmp/e2e.sam"]), ("24m", ["--batch", "524288", "-S", "/dev/null"])):
    t0 = time.time()
    p = subprocess.run(["bowtie2_amd/bin/bowtie2-align-l", "--sensitive", "-t", "-p", "16"] + extra + ["-x", B, "-U", "/tmp/e2e_%s.fq" % sz], stderr=subprocess.PIPE, text=True, timeout=600)
    w = time.time() - t0
    print("%s %s: %.2f s wall (process start and index load included), rc %d" % (sz, " ".join(extra), w, p.returncode))
    print("\n".join(l for l in p.stderr.splitlines() if "bt2g" in l))
    m = re.search(r"-> (\d+) reads/s after the load", p.stderr)
    if m: print("   = %.2f x the resident-batch rate" % (int(m.group(1)) / res))
P
rm -f /tmp/e2e_8m.fq /tmp/e2e_24m.fq /tmp/e2e.sam

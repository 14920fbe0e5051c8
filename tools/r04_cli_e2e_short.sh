#!/bin/bash
# tools/r04_cli_e2e.sh cut to what the last minutes of a round allow: the resident-batch rate, then 24 M reads through the drop-in binary twice
T=${1:-r04e}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(timeout 200 python bench.py --steps 4 --warmup 2 --parity-only 2>$O/bench.err | tail -1) > $O/bench.json
C=/tmp/bt2_amd_bench
for i in 1 2 3 4 5 6 7 8; do cat $C/sample.fq $C/sample.fq $C/sample.fq; done > /tmp/e2e_24m.fq
python3 - "$O" <<'P' | tee $O/e2e.txt
import json, subprocess, sys, time, re
O = sys.argv[1]
res = json.load(open(O + "/bench.json"))["value"]
print("resident-batch rate (bench.py, 2 M reads per launch): %d reads/s" % res)
B = "/tmp/bt2_amd_bench/hg38like_3100mbp_s2_bt2l"
for extra in (["-S", "/dev/null"], ["-S", "/tmp/e2e.sam"]):
    t0 = time.time()
    p = subprocess.run(["bowtie2_amd/bin/bowtie2-align-l", "--sensitive", "-t", "-p", "16"] + extra + ["-x", B, "-U", "/tmp/e2e_24m.fq"], stderr=subprocess.PIPE, text=True, timeout=300)
    print("24m %s: %.2f s wall (process start and index load included), rc %d" % (" ".join(extra), time.time() - t0, p.returncode))
    print("\n".join(l for l in p.stderr.splitlines() if "bt2g" in l))
    m = re.search(r"-> (\d+) reads/s after the load", p.stderr)
    if m: print("   = %.2f x the resident-batch rate" % (int(m.group(1)) / res))
P
rm -f /tmp/e2e_24m.fq /tmp/e2e.sam

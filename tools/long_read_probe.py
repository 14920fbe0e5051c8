#!/usr/bin/env python3
"""Throughput of the worker's long-read class next to the reference binary: the 240 example long reads (513 ... 1 999 bp, tests/golden/example)
N times over (distinct names) against phage lambda, --sensitive and --local; the SAM of the two must be identical.
Usage (GPU box):  python tools/long_read_probe.py [copies]  > gpurun_out/<tag>/long_reads.log 2>&1"""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bt2test import ref_bin          # noqa: E402
from test_long_reads import workload  # noqa: E402

BIN = os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-align-s")


def main():
    copies = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    base, fq = workload()
    recs = open(fq).read().split("\n")
    big = fq + ".x%d" % copies
    with open(big, "w") as f:
        for k in range(copies):
            for i in range(0, len(recs) - 3, 4):
                f.write("%s_%d\n%s\n+\n%s\n" % (recs[i], k, recs[i + 1], recs[i + 3]))
    n = copies * (len(recs) // 4)
    threads = str(min(16, os.cpu_count() or 1))
    body = lambda t: [l for l in t.splitlines() if not l.startswith("@PG")]
    # (the first process on a fresh box pays for the device's first large allocation -- seconds: an untimed run of the binary comes first,
    # and every mode is run twice)
    subprocess.run([BIN, "-p", threads, "-x", base, "-U", big], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for args in (["--sensitive"], ["--local"]):
        t0 = time.time()
        r = subprocess.run([ref_bin("bowtie2-align-s"), "-p", threads, "--reorder"] + args + ["-x", base, "-U", big], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        t_ref = time.time() - t0
        for run in (1, 2):
            t0 = time.time()
            p = subprocess.run([BIN, "-t", "-p", threads, "--reorder"] + args + ["-x", base, "-U", big], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            t_our = time.time() - t0
            m = re.search(r"search ([\d.]+) s wall, (\d+) reads -> (\d+) reads/s after the load", p.stderr)
            warns = sum(1 for l in p.stderr.splitlines() if l.startswith("Warning: read"))
            print("%-12s run %d, %d reads: reference -p %s %.2f s wall = %.0f reads/s; this build %.2f s wall (search %s s = %s reads/s after the load), rc %d, %d reads flagged, SAM identical: %s"
                  % (" ".join(args), run, n, threads, t_ref, n / t_ref, t_our, m.group(1) if m else "?", m.group(3) if m else "?", p.returncode, warns, body(r.stdout) == body(p.stdout)), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Pairs whose mates are long enough for the 16-bit end-to-end kernel (minimum score below -254: mates of 424 bp and more at the default threshold)
against the reference binary: phage lambda, mates of 430-510 bp, fragments of 800-1 150 bp, substitutions and indels, a few unrelated mates.
usage: tools/long_pairs_check.py [pairs] [seed]   -- prints one line per option set; exit status 1 if any SAM differs or a read was flagged."""
import os, random, subprocess, sys, gzip, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bt2test import ref_bin, write_fastq, revcomp, build_index
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
fa = os.path.join(ROOT, "tests", "golden", "example", "lambda_virus.fa")
g = "".join(l.strip() for l in open(fa) if not l.startswith(">")).upper()
def mut(s):
    o = []
    for c in s:
        r = rnd.random()
        if r < 0.02: o.append(rnd.choice("ACGT"))
        elif r < 0.023: continue
        elif r < 0.026: o.append(c); o.append(rnd.choice("ACGT"))
        else: o.append(c)
    return "".join(o)
r1, r2 = [], []
for i in range(n):
    fl = rnd.randrange(800, 1150); p = rnd.randrange(0, len(g) - fl)
    l1, l2 = rnd.randrange(430, 511), rnd.randrange(430, 511)
    a, b_ = g[p:p + l1], revcomp(g[p + fl - l2:p + fl])
    if rnd.random() < 0.5: a, b_ = b_, a
    if rnd.random() < 0.05: b_ = "".join(rnd.choice("ACGT") for _ in b_)
    a, b_ = mut(a), mut(b_)
    r1.append(("p%d" % i, a, "".join(rnd.choice("IIIH?5") for _ in a))); r2.append(("p%d" % i, b_, "".join(rnd.choice("IIIH?5") for _ in b_)))
d = tempfile.mkdtemp(prefix="longpairs")
write_fastq(d + "/1.fq", r1); write_fastq(d + "/2.fq", r2)
base = d + "/lambda"; build_index(fa, base)
ours = os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-align-s")
bad = 0
for args in (["-X", "1200"], ["--very-sensitive", "-X", "1200", "--no-mixed"], ["-X", "1200", "-k", "3", "--ff"], ["-X", "700"]):
    cmd = ["-x", base, "-1", d + "/1.fq", "-2", d + "/2.fq", "--reorder", "-p", "4"] + args
    want = subprocess.run([ref_bin("bowtie2-align-s")] + cmd, capture_output=True, text=True)
    got = subprocess.run([ours] + cmd, capture_output=True, text=True)
    w = [l for l in want.stdout.splitlines() if not l.startswith("@PG")]; o = [l for l in got.stdout.splitlines() if not l.startswith("@PG")]
    nd = sum(1 for x, y in zip(w, o) if x != y) + abs(len(w) - len(o))
    flagged = got.stderr.count("capacity site")
    import collections
    sites = collections.Counter(l.rsplit("capacity site", 1)[1].strip() for l in got.stderr.splitlines() if "capacity site" in l)
    if sites: print("   capacity sites:", dict(sites))
    print("long pairs %-45s rc %d/%d  SAM lines %d, differing %d, reads flagged %d" % (" ".join(args), want.returncode, got.returncode, len(w), nd, flagged))
    if nd or flagged or got.returncode != want.returncode: bad = 1
sys.exit(bad)

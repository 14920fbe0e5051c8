"""--overhang (gReportOverhangs): reads hanging off either end of a reference sequence are aligned against N padding and
come back soft-clipped.  Differential against the reference binaries on reads built to overhang by 0-15 bp, with
mismatches and indels, both strands, both index widths.  CPU: the host-compiled worker; GPU: the product binary."""
import os
import random
import subprocess

import pytest

from bt2test import CACHE_DIR, build_index, have_ref, ref_bin, revcomp, write_fasta, write_fastq, build_hostsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HS = os.path.join(ROOT, "tests", "hostsim")
OPTION_SETS = (["--overhang"], ["--overhang", "--local"], ["--overhang", "-k", "3"], ["--overhang", "-N", "1", "-L", "14"],
               ["--overhang", "--very-sensitive", "--n-ceil", "L,0,0.3"],
               ["--overhang", "--score-min", "L,0,-0.2", "--rdg", "10,4", "--rfg", "10,4"], ["--overhang", "-a", "--local", "--ma", "3"])


def overhang_case():
    rnd = random.Random(11)
    refs = [("r%d" % i, "".join(rnd.choice("ACGT") for _ in range(rnd.randrange(300, 1500)))) for i in range(4)]
    reads = []
    for i in range(600):
        _, s = refs[rnd.randrange(4)]
        L = rnd.randrange(30, 120)
        ov = rnd.randrange(0, 16)
        junk = "".join(rnd.choice("ACGT") for _ in range(ov))
        seq = junk + s[:L - ov] if rnd.random() < 0.5 else s[len(s) - (L - ov):] + junk
        seq = "".join(c if rnd.random() > 0.02 else rnd.choice("ACGT") for c in seq)
        if rnd.random() < 0.1:
            p = rnd.randrange(1, len(seq) - 1)
            seq = seq[:p] + seq[p + 1:]
        if rnd.random() < 0.1:
            p = rnd.randrange(1, len(seq) - 1)
            seq = seq[:p] + rnd.choice("ACGT") + seq[p:]
        if rnd.random() < 0.5:
            seq = revcomp(seq)
        reads.append(("q%d" % i, seq, "".join(rnd.choice("IIIH?5") for _ in seq)))
    return refs, reads


def check(exe_s, exe_l):
    refs, reads = overhang_case()
    d = os.path.join(CACHE_DIR, "overhang")
    os.makedirs(d, exist_ok=True)
    fa, fq = os.path.join(d, "o.fa"), os.path.join(d, "o.fq")
    write_fasta(fa, refs)
    write_fastq(fq, reads)
    clipped = 0
    for large, exe in ((False, exe_s), (True, exe_l)):
        base = os.path.join(d, "o" + ("l" if large else "s"))
        build_index(fa, base, large)
        ref_exe = ref_bin("bowtie2-align-l" if large else "bowtie2-align-s")
        for args in OPTION_SETS:
            rs = os.path.join(d, "ref.sam")
            subprocess.check_call([ref_exe] + args + ["-x", base, "-U", fq, "-p", "4", "--reorder", "-S", rs], stderr=subprocess.DEVNULL)
            want = [l.rstrip("\n") for l in open(rs) if not l.startswith("@PG")]
            p = subprocess.run([exe] + args + ["-x", base, "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            assert p.returncode == 0 and "Warning" not in p.stderr, p.stderr[-800:]
            got = [l for l in p.stdout.splitlines() if not l.startswith("@PG")]
            assert len(got) == len(want), args
            bad = [i for i in range(len(got)) if got[i] != want[i]]
            assert not bad, (large, args, len(bad), want[bad[0]], got[bad[0]])
            clipped += sum(1 for l in want if not l.startswith("@") and "S" in l.split("\t")[5])
    assert clipped > 3000      # the case really exercises the clipping


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
def test_overhang_hostsim():
    exe = os.path.join(HS, "hostsim")
    build_hostsim(exe)
    check(exe, exe)


@pytest.mark.gpu
@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
def test_overhang_gpu():
    b = os.path.join(ROOT, "bowtie2_amd", "bin")
    check(os.path.join(b, "bowtie2-align-s"), os.path.join(b, "bowtie2-align-l"))

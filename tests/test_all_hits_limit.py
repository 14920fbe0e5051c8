"""-a and -k above 64.  The reference has no ceiling on -k (aln_sink.cpp:33-326); this build's was 64 alignments per
result record, and a read with more was flagged.  Such batches now run in the worker's many-alignments class (BT2G_CLASS_BIG_K: 1000 alignments
per read, the extension list of maxIters = 400 + 20 (k - 1) rows): the fuzzer's -a case whose reads r6 / r59 have 74 / 70 alignments
(tests/golden/all_hits/README.md) must come out as the reference wrote it, with nothing flagged; -k 65 ... -k 1000 and -a on a genome made of
a 1 500-copy repeat family must equal the reference binary's output; -k 1001 is refused."""
import os
import subprocess

import pytest

from bt2test import CACHE_DIR, build_index, build_hostsim, ref_bin, write_fasta, write_fastq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "all_hits")
HS = os.path.join(ROOT, "tests", "hostsim")
BIN = os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-align-s")


def body(text):
    return [l for l in text.splitlines() if not l.startswith("@PG")]


def check_fuzzer_case(exe, tmp_path):
    base = str(tmp_path / "g")
    build_index(os.path.join(GOLD, "genome.fa"), base, False)
    p = subprocess.run([exe, "-a", "-x", base, "-U", os.path.join(GOLD, "reads.fq")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-800:]
    assert not [l for l in p.stderr.splitlines() if l.startswith("Warning: read")], p.stderr[-800:]
    want = body(open(os.path.join(GOLD, "reference_a.sam")).read())
    got = body(p.stdout)
    assert sum(1 for l in want if l.startswith("r6\t")) == 74 and sum(1 for l in want if l.startswith("r59\t")) == 70
    assert got == want


def family_workload():
    """A 260 kbp genome that is mostly one 120-bp family in ~1 500 copies at 0-6 % divergence, and reads from it: hundreds to a thousand-odd
    alignments per read."""
    import random
    d = os.path.join(CACHE_DIR, "many_alns_family")
    base, fq = os.path.join(d, "idx"), os.path.join(d, "reads.fq")
    if not os.path.exists(base + ".rev.2.bt2"):
        os.makedirs(d, exist_ok=True)
        rnd = random.Random(4242)
        cons = "".join(rnd.choice("ACGT") for _ in range(120))
        parts = []
        for _ in range(1500):
            div = rnd.random() * 0.06
            parts.append("".join((rnd.choice("ACGT") if rnd.random() < div else c) for c in cons))
            parts.append("".join(rnd.choice("ACGT") for _ in range(rnd.randint(20, 90))))
        g = "".join(parts)
        write_fasta(os.path.join(d, "genome.fa"), [("fam", g)])
        build_index(os.path.join(d, "genome.fa"), base, False)
        reads = []
        for i in range(24):
            s = rnd.randint(0, 40)
            seq = "".join((rnd.choice("ACGT") if rnd.random() < 0.01 else c) for c in cons[s:s + 70])
            reads.append(("f%d" % i, seq, "I" * len(seq)))
        for i in range(8):      # and a few unique reads
            s = rnd.randint(0, len(g) - 200)
            reads.append(("u%d" % i, g[s:s + 100], "I" * 100))
        write_fastq(fq, reads)
    return base, fq


def check_family(exe, args, product):
    base, fq = family_workload()
    want = subprocess.run([ref_bin("bowtie2-align-s")] + args + ["-x", base, "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True).stdout
    p = subprocess.run([exe] + args + ["-x", base, "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
    w, g = body(want), body(p.stdout)
    per_read = {}
    for l in w:
        if not l.startswith("@"):
            per_read.setdefault(l.split("\t", 1)[0], []).append(l)
    assert max(len(v) for v in per_read.values()) > 64, "the workload must exercise records above 64 alignments"
    # -a: a read with more alignments than a record holds (1000) is flagged -- warning, and exit status 1 from the product binary -- never cut silently
    over = sorted(n for n, v in per_read.items() if len(v) > 1000)
    flagged = sorted(l.split()[2].rstrip(":") for l in p.stderr.splitlines() if l.startswith("Warning: read"))
    assert flagged == over, (flagged, over, p.stderr[-500:])
    assert p.returncode == (1 if over and product else 0), p.stderr[-800:]
    if "-a" in args:
        assert over and len(over) < len(per_read), "the -a case wants reads on both sides of the ceiling"
    keep = lambda lines: [l for l in lines if l.startswith("@") or l.split("\t", 1)[0] not in over]
    assert keep(g) == keep(w), "%s: %d vs %d lines" % (" ".join(args), len(g), len(w))


@pytest.fixture(scope="module")
def hostsim():
    exe = os.path.join(HS, "hostsim_allhits")
    build_hostsim(exe)
    return exe


def test_all_hits_fuzzer_case_hostsim(hostsim, tmp_path):
    check_fuzzer_case(hostsim, tmp_path)


@pytest.mark.parametrize("args", [["-k", "65"], ["-k", "200"], ["-k", "1000"], ["-a"], ["-k", "300", "--local"]])
def test_many_alignments_hostsim(hostsim, args):
    check_family(hostsim, args, False)


def test_k_above_the_ceiling_is_refused(hostsim):
    base, fq = family_workload()
    p = subprocess.run([hostsim, "-k", "1001", "-x", base, "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode != 0 and "-k above 1000" in p.stderr


@pytest.mark.gpu
def test_all_hits_fuzzer_case_gpu(tmp_path):
    check_fuzzer_case(BIN, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["-k", "65"], ["-k", "200"], ["-k", "1000"], ["-a"], ["-k", "300", "--local"]])
def test_many_alignments_gpu(args):
    check_family(BIN, args, True)


@pytest.mark.gpu
def test_many_alignments_pairs_gpu(tmp_path):
    """pairs with -k 100: the golden paired reads against the tiny index equal the reference binary's output"""
    gold = os.path.join(ROOT, "tests", "golden")
    args = ["-k", "100", "-x", os.path.join(gold, "tiny_s"), "-1", os.path.join(gold, "pe_reads_1.fq"), "-2", os.path.join(gold, "pe_reads_2.fq")]
    want = subprocess.run([ref_bin("bowtie2-align-s")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True).stdout
    p = subprocess.run([BIN] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-500:]
    assert body(p.stdout) == body(want)

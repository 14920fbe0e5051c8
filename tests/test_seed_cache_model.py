"""The reference keeps the seed hits of one seeding round in a pool of 16 KB pages (--seed-cache-sz MB, 20 by default): every SA range is stored
with one slot per row, and a read whose seeds hit more rows than the pool holds has the range being stored cut, that seed dropped and later
new seeds dropped too (AlignmentCache::addOnTheFlyImpl, aligner_cache.cpp:53-105; searchAllSeeds, aligner_seed.cpp:672-690).  SAM parity
needs exactly that, so the worker replays the page accounting (CacheModel; cache_filter for -N 0, cache_account_mm1 for -N 1 where a seed
owns one range per reference string).  Here: a genome with a 30 000-copy tandem family and a pool of 1 MB, so that the pool does run out --
the reference's SAM changes with --seed-cache-sz, and ours must change the same way (CPU twin; GPU binary under -m gpu)."""
import os
import random
import subprocess

import pytest

from bt2test import CACHE_DIR, build_index, have_ref, ref_bin, write_fasta, write_fastq, build_hostsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-align-s")
OPTS = [["-N", "1"], ["-N", "1", "-L", "14", "-i", "C,5,0", "-k", "3"], ["-N", "1", "--local"], ["-N", "0", "-L", "16", "-i", "C,4,0"]]


def workload(large):
    d = os.path.join(CACHE_DIR, "tandem_%s" % ("l" if large else "s"))
    os.makedirs(d, exist_ok=True)
    fa, fq, base = os.path.join(d, "t.fa"), os.path.join(d, "t.fq"), os.path.join(d, "t")
    if not os.path.exists(fq):
        rng = random.Random(12)
        rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
        unit = rnd(41)
        copies = []
        for _ in range(30000):
            u = list(unit)
            for k in range(len(u)):
                if rng.random() < 0.01:
                    u[k] = rng.choice("ACGT")
            copies.append("".join(u))
        g = rnd(150000) + "".join(copies) + rnd(150000)
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        reads = []
        for n in range(300):
            p = rng.randrange(0, 150000 - 120) if n % 4 == 0 else rng.randrange(150000, 150000 + 30000 * 41 - 120)
            s = list(g[p:p + 90])
            for k in range(90):
                if rng.random() < 0.02:
                    s[k] = rng.choice("ACGT")
            s = "".join(s)
            if rng.random() < 0.5:
                s = "".join(comp[c] for c in reversed(s))
            reads.append(("q%d" % n, s, "I" * 90))
        rc = lambda t: "".join(comp[c] for c in reversed(t))
        m1, m2 = [], []
        for n in range(150):          # pairs: both mates in the tandem family, so that they hit the same reference strings and share the pool
            p = rng.randrange(100000, len(g) - 100000 - 400)
            frag = g[p:p + rng.randrange(200, 320)]
            mut = lambda t: "".join(rng.choice("ACGT") if rng.random() < 0.02 else ch for ch in t)
            m1.append(("p%d/1" % n, mut(frag[:80]), "I" * 80))
            m2.append(("p%d/2" % n, mut(rc(frag[-80:])), "I" * 80))
        write_fasta(fa, [("chrT", g)])
        write_fastq(os.path.join(d, "t_1.fq"), m1)
        write_fastq(os.path.join(d, "t_2.fq"), m2)
        write_fastq(fq, reads)
        build_index(fa, base, large)
    return base, fq


def body(text):
    return [l for l in text.splitlines() if not l.startswith("@PG")]


def check(exe, large):
    base, fq = workload(large)
    ref = ref_bin("bowtie2-align-l" if large else "bowtie2-align-s")
    changed = 0
    for opts in OPTS:
        sams = []
        for sz in ([], ["--seed-cache-sz", "1"]):
            r = subprocess.run([ref] + opts + sz + ["-x", base, "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
            o = subprocess.run([exe] + opts + sz + ["-x", base, "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
            assert r.returncode == 0 and o.returncode == 0, (opts, sz, r.stderr[-300:], o.stderr[-300:])
            assert "Warning" not in o.stderr, o.stderr[-300:]
            assert body(o.stdout) == body(r.stdout), (large, opts, sz)
            sams.append(body(r.stdout))
        changed += sams[0] != sams[1]
    assert changed >= 2        # the small pool did run out: otherwise this test shows nothing
    # pairs: the two mates' seeds of a round share the pool, and a mate's seed may find its reference strings already stored (cut or not)
    # by the other mate -- in the order the reference's lockstep search of the two seed policies discovered them
    d = os.path.dirname(fq)
    changed = 0
    for opts in ([["-N", "1"], ["-N", "1", "-L", "14", "-i", "C,5,0"], ["-N", "0", "-L", "16", "-i", "C,4,0"]]):
        sams = []
        for sz in ([], ["--seed-cache-sz", "1"]):
            io = ["-x", base, "-1", os.path.join(d, "t_1.fq"), "-2", os.path.join(d, "t_2.fq")]
            r = subprocess.run([ref] + opts + sz + io, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
            o = subprocess.run([exe] + opts + sz + io, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
            assert r.returncode == 0 and o.returncode == 0, (opts, sz, r.stderr[-300:], o.stderr[-300:])
            assert body(o.stdout) == body(r.stdout), (large, "pairs", opts, sz)
            sams.append(body(r.stdout))
        changed += sams[0] != sams[1]
    assert changed >= 2


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("large", [False, True])
def test_pool_exhaustion_host_twin(tmp_path, large):
    exe = str(tmp_path / "hostsim")
    build_hostsim(exe)
    check(exe, large)


@pytest.mark.gpu
@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("large", [False, True])
def test_pool_exhaustion_gpu(large):
    check(EXE, large)

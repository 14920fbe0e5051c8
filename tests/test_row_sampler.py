"""The row sampler (RowSampler + Random1toN, aligner_sw_driver.h:179-256, random_util.h:32-219) decides which rows of the seed-hit ranges are
extended, and every draw consumes the read's RNG: SAM parity needs the draws replayed exactly.  The worker replays them without any list in
memory (Aligner::sample_rows_fast: swap lists as sparse overrides of the identity, seen lists as a set, converted lists as "i + number of seen
values with value - rank <= i", all in one table held in lane registers).  The three list forms only occur for ranges of particular sizes:
  * < 128 rows: swap list from the first draw;
  * >= 128 rows: seen list, converted to a swap list of the unseen rows once max(16, 10 %) rows have been drawn from the range.
Here: a genome with repeat families of 60, 150, 400, 1 500 and 6 000 copies (1-3 % divergence) so that the seeds of one read own ranges of
all those sizes at once, reads that start inside a copy or straddle its edge, and option sets that change how many rows are drawn (-D/-R,
-k, -L/-i, --local; -k 40 needs more draws than the register table holds and takes the arena path).  SAM must equal the reference's, byte for byte -- CPU twin here, the device binary under -m gpu."""
import os
import random
import subprocess

import pytest

from bt2test import CACHE_DIR, build_index, have_ref, ref_bin, write_fasta, write_fastq, build_hostsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-align-s")
OPTS = [["--sensitive"], ["--very-sensitive"], ["-k", "4"], ["-k", "20", "--very-fast"], ["-L", "16", "-i", "C,6,0", "-D", "30", "-R", "3"], ["--local"], ["-k", "40", "--very-fast"]]      # -k 40: more draws than the register table holds -> the arena path


def workload(large):
    d = os.path.join(CACHE_DIR, "rowsamp_%s" % ("l" if large else "s"))
    os.makedirs(d, exist_ok=True)
    fa, fq, base = os.path.join(d, "g.fa"), os.path.join(d, "r.fq"), os.path.join(d, "g")
    if not os.path.exists(fq):
        rng = random.Random(77)
        rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
        fams = [(rnd(180), 60, 0.01), (rnd(220), 150, 0.02), (rnd(160), 400, 0.02), (rnd(200), 1500, 0.03), (rnd(140), 6000, 0.03)]
        pieces, spots = [], []
        order = []
        for fi, (cons, copies, div) in enumerate(fams):
            order += [fi] * copies
        rng.shuffle(order)
        pos = 0
        for fi in order:
            cons, _, div = fams[fi]
            spacer = rnd(rng.randrange(20, 120))
            copy = "".join(rng.choice("ACGT") if rng.random() < div else ch for ch in cons)
            pieces.append(spacer); pos += len(spacer)
            spots.append((pos, len(copy)))
            pieces.append(copy); pos += len(copy)
        g = "".join(pieces) + rnd(5000)
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        reads = []
        for n in range(700):
            p0, ln = spots[rng.randrange(len(spots))]
            L = rng.choice([60, 100, 150])
            p = max(0, min(len(g) - L, p0 + rng.randrange(-L // 2, ln - L // 2)))
            s = list(g[p:p + L])
            for k in range(L):
                if rng.random() < 0.015:
                    s[k] = rng.choice("ACGT")
            if rng.random() < 0.15:
                k = rng.randrange(10, L - 10)
                s = s[:k] + ([rng.choice("ACGT")] if rng.random() < 0.5 else []) + s[k + (0 if rng.random() < 0.5 else 1):]
            s = "".join(s)
            if rng.random() < 0.5:
                s = "".join(comp[c] for c in reversed(s))
            reads.append(("q%d" % n, s, "".join(rng.choice("I5+") for _ in s)))
        write_fasta(fa, [("chrR", g)])
        write_fastq(fq, reads)
        build_index(fa, base, large)
    return base, fq


def body(text):
    return [l for l in text.splitlines() if not l.startswith("@PG")]


def check(exe, large, jobs=1):
    """jobs > 1 (the CPU twin): the option sets are independent processes and go through a thread pool."""
    from concurrent.futures import ThreadPoolExecutor
    base, fq = workload(large)
    ref = ref_bin("bowtie2-align-l" if large else "bowtie2-align-s")

    def one(opts):
        r = subprocess.run([ref] + opts + ["-x", base, "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1800)
        o = subprocess.run([exe] + opts + ["-x", base, "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1800)
        return r, o

    if jobs > 1:
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            done = list(ex.map(one, OPTS))
    else:
        done = [one(o) for o in OPTS]
    for opts, (r, o) in zip(OPTS, done):
        assert r.returncode == 0, (opts, r.stderr[-300:])
        assert o.returncode == 0 and "Warning" not in o.stderr, (opts, o.stderr[-300:])
        a, b = body(r.stdout), body(o.stdout)
        ndiff = sum(1 for x, y in zip(a, b) if x != y) + abs(len(a) - len(b))
        assert ndiff == 0, (large, opts, ndiff, [(x, y) for x, y in zip(a, b) if x != y][:2])


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("large", [False, True])
def test_row_sampler_host_twin(tmp_path, large):
    exe = str(tmp_path / "hostsim")
    build_hostsim(exe)
    check(exe, large, jobs=min(8, os.cpu_count() or 1))


@pytest.mark.gpu
@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("large", [False, True])
def test_row_sampler_gpu(large):
    check(EXE, large)

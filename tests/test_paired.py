"""Paired-end alignment (-1/-2): extendSeedsPaired, opposite-mate DP, concordant / discordant / unpaired reporting,
mate fields of the SAM records and the paired summary.  Differential against the reference binaries on pairs drawn from
a repeat-rich synthetic genome: proper pairs (fragment ~300 +- 40), over-long fragments, wrong orientation, a junk mate,
mates on different sequences; both strands, indels, mixed read lengths.  CPU: host-compiled worker; GPU: product binary."""
import os
import random
import subprocess

import pytest

from bt2test import CACHE_DIR, build_index, have_ref, ref_bin, revcomp, write_fasta, write_fastq, build_hostsim
from test_gpu_align import repeat_genome

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HS = os.path.join(ROOT, "tests", "hostsim")
OPTION_SETS = ([], ["--local"], ["-k", "3"], ["--no-mixed"], ["--no-discordant"], ["-I", "200", "-X", "350"], ["--ff"],
               ["--very-sensitive", "--dovetail"], ["--no-contain", "--no-overlap"], ["-N", "1", "-L", "18"],
               ["--very-sensitive-local", "-k", "4"], ["--rf", "-X", "700"], ["--no-unal", "--xeq", "-3", "5"], ["--passthrough"],
               # opposite-mate windows wider than the 1 100 columns a launch holds by default (bt2g_align_params::max_dp_cols): -X + mate + 2 x gaps
               ["--local", "-X", "1000"], ["-X", "1500", "--dovetail"])


def make_pairs(n, seed):
    rnd = random.Random(seed)
    refs, _ = repeat_genome()

    def mut(s, sub=0.01, indel=0.002):
        out = []
        for c in s:
            r = rnd.random()
            if r < sub:
                out.append(rnd.choice("ACGT"))
            elif r < sub + indel:
                continue
            elif r < sub + 2 * indel:
                out.append(c)
                out.append(rnd.choice("ACGT"))
            else:
                out.append(c)
        return "".join(out)

    r1, r2 = [], []
    for i in range(n):
        _, s = refs[rnd.randrange(len(refs))]
        L1, L2 = rnd.randrange(40, 151), rnd.randrange(40, 151)
        kind = rnd.random()
        frag = max(int(rnd.gauss(300, 40)), max(L1, L2) + 5)
        if kind < 0.05:
            frag = rnd.randrange(600, 1500)
        p = rnd.randrange(0, len(s) - frag - 1)
        f = s[p:p + frag]
        m1, m2 = f[:L1], revcomp(f[-L2:])
        if kind > 0.95:
            m2 = revcomp(m2)
        if 0.90 < kind <= 0.95:
            m2 = "".join(rnd.choice("ACGT") for _ in range(L2))
        if 0.85 < kind <= 0.90:
            _, s2 = refs[rnd.randrange(len(refs))]
            q = rnd.randrange(0, len(s2) - L2 - 1)
            m2 = revcomp(s2[q:q + L2])
        if rnd.random() < 0.5:
            m1, m2 = m2, m1
        m1, m2 = mut(m1), mut(m2)
        r1.append(("p%d/1" % i, m1, "".join(rnd.choice("IIIIHH?5") for _ in m1)))
        r2.append(("p%d/2" % i, m2, "".join(rnd.choice("IIIIHH?5") for _ in m2)))
    return refs, r1, r2


def check(exe_s, exe_l, n, option_sets, extra=(), jobs=1):
    """jobs > 1 (the CPU twin): the (index width, option set) runs are independent processes and go through a thread pool."""
    from concurrent.futures import ThreadPoolExecutor
    refs, r1, r2 = make_pairs(n, 9)
    d = os.path.join(CACHE_DIR, "paired")
    os.makedirs(d, exist_ok=True)
    fa, f1, f2 = os.path.join(d, "g.fa"), os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")
    write_fasta(fa, refs)
    write_fastq(f1, r1)
    write_fastq(f2, r2)
    todo = []
    for large, exe in ((False, exe_s), (True, exe_l)):
        base = os.path.join(d, "g" + ("l" if large else "s"))
        build_index(fa, base, large)
        for k, args in enumerate(option_sets):
            todo.append((large, exe, base, args, os.path.join(d, "ref_%d_%d.sam" % (large, k))))

    def one(item):
        large, exe, base, args, rs = item
        ref_exe = ref_bin("bowtie2-align-l" if large else "bowtie2-align-s")
        pr = subprocess.run([ref_exe] + args + ["-x", base, "-1", f1, "-2", f2, "-p", "8" if jobs == 1 else "2", "--reorder", "-S", rs], stderr=subprocess.PIPE, text=True)
        want = [l.rstrip("\n") for l in open(rs) if not l.startswith("@PG")]
        os.remove(rs)
        p = subprocess.run([exe] + list(extra) + args + ["-x", base, "-1", f1, "-2", f2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        return pr, want, p

    if jobs > 1:
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            done = list(ex.map(one, todo))
    else:
        done = [one(t) for t in todo]
    kinds = set()
    for (large, exe, base, args, rs), (pr, want, p) in zip(todo, done):
        assert p.returncode == 0 and "Warning" not in p.stderr, p.stderr[-800:]
        got = [l for l in p.stdout.splitlines() if not l.startswith("@PG")]
        assert len(got) == len(want), (large, args)
        bad = [i for i in range(len(got)) if got[i] != want[i]]
        assert not bad, (large, args, len(bad), want[bad[0]], got[bad[0]])
        keep = lambda txt: [l for l in txt.splitlines() if not l.startswith("Warning") and "amdgpu.ids" not in l and not l.startswith("[bt2g]")]
        assert keep(p.stderr) == keep(pr.stderr), (large, args)      # the paired alignment summary
        kinds.update(l.rsplit("YT:Z:", 1)[1][:2] for l in want if "YT:Z:" in l)
    assert {"CP", "DP", "UP"} <= kinds
    # --interleaved: the same pairs from one file; -s/-u count pairs
    fi = os.path.join(d, "inter.fq")
    l1, l2 = open(f1).read().splitlines(), open(f2).read().splitlines()
    open(fi, "w").write("".join("\n".join(l1[i:i + 4] + l2[i:i + 4]) + "\n" for i in range(0, len(l1), 4)))
    base = os.path.join(d, "gs")
    for args in ([], ["-s", "10", "-u", "50"], ["--local", "-k", "2"]):
        a = subprocess.run([ref_bin("bowtie2-align-s")] + args + ["-x", base, "--interleaved", fi, "--reorder", "-p", "4"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        b = subprocess.run([exe_s] + list(extra) + args + ["-x", base, "--interleaved", fi], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        body = lambda t: [l for l in t.splitlines() if not l.startswith("@PG")]
        assert body(a.stdout) == body(b.stdout), args


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
def test_paired_hostsim():
    exe = os.path.join(HS, "hostsim")
    build_hostsim(exe)
    check(exe, exe, 500, OPTION_SETS, jobs=min(8, os.cpu_count() or 1))


@pytest.mark.gpu
@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
def test_paired_gpu():
    b = os.path.join(ROOT, "bowtie2_amd", "bin")
    check(os.path.join(b, "bowtie2-align-s"), os.path.join(b, "bowtie2-align-l"), 1000, OPTION_SETS, extra=("-p", "4"))

#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container (needs oracle/_ref, i.e. /root/reference):  python tests/golden/make_golden.py

Outputs (committed):
  tiny.fa, tiny_{s,l}.*.bt2[l]   a ~6 kbp, 2-reference genome indexed by the reference's own
                                 bowtie2-build-{s,l} with --ftabchars 5 --offrate 3 (few KB per file)
  fm_golden_{s,l}.json           answers of the reference's Ebwt / SeedAligner classes (via
                                 oracle/ref_shim.cpp) on that index: rank4, mapLF1, getOffset,
                                 joinedToTextOff, exactSweep, exact seed rounds
  dp_golden.json                 end-to-end u8 DP problems with sha256 of the reference's filled
                                 H|E|F matrices (alignNucleotidesEnd2EndSseU8) and best score
  rng_golden.json                RandomSource streams
"""
import ctypes as C
import json
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from bt2test import *  # noqa


def build_tiny():
    refs = synth_genome(n_refs=2, total=6000, seed=11)
    fa = os.path.join(HERE, "tiny.fa")
    write_fasta(fa, refs)
    for large in (False, True):
        exe = ref_bin("bowtie2-build-l" if large else "bowtie2-build-s")
        base = os.path.join(HERE, "tiny_l" if large else "tiny_s")
        subprocess.check_call([exe, "-q", "--ftabchars", "5", "--offrate", "3", fa, base], stdout=subprocess.DEVNULL)
    return refs


def fm_golden(large, refs):
    R = refshim(large)
    base = os.path.join(HERE, "tiny_l" if large else "tiny_s")
    h = R.ref_open(base.encode())
    assert h
    n = R.ref_len(h)
    rnd = random.Random(99)
    g = {"len": n, "zoff": [R.ref_zoff(h, 0), R.ref_zoff(h, 1)]}
    rows = sorted(set([0, 1, n - 1, n, g["zoff"][0], g["zoff"][0] + 1, g["zoff"][1], g["zoff"][1] + 1, 191, 192, 193, 383, 384, 385]
                      + [rnd.randrange(0, n + 1) for _ in range(300)]))
    rows = [r for r in rows if 0 <= r <= n]
    a = (u64 * 4)()
    g["rows"] = rows
    g["rank4"] = []
    g["lf1c"] = []
    for d in (0, 1):
        rr, ll = [], []
        for row in rows:
            R.ref_rank4(h, d, row, a)
            rr.append(list(a))
            ll.append([(lambda v: -1 if v == 2**64 - 1 else v)(R.ref_map_lf1c(h, d, row, c)) for c in range(4)])
        g["rank4"].append(rr)
        g["lf1c"].append(ll)
    g["get_offset"] = [R.ref_get_offset(h, row) for row in rows]
    jo = []
    for _ in range(200):
        off = rnd.randrange(0, n)
        q = rnd.choice([1, 10, 22, 60])
        if off + q > n:
            continue
        for rej in (0, 1):
            t = [u64(), u64(), u64()]
            s = C.c_int()
            R.ref_joined_to_text_off(h, q, off, C.byref(t[0]), C.byref(t[1]), C.byref(t[2]), rej, C.byref(s))
            ti = -1 if t[0].value == 2**64 - 1 else t[0].value
            jo.append([q, off, rej, ti, t[1].value if ti >= 0 else 0, t[2].value if ti >= 0 else 0, s.value])
    g["joined"] = jo
    reads = (synth_reads(refs, 60, 50, seed=21) + synth_reads(refs, 30, 80, seed=22, sub=0, ins=0, dele=0)
             + synth_reads(refs, 30, 40, seed=23, n_rate=0.02) + synth_reads(refs, 10, 8, seed=24)
             + synth_reads(refs, 10, 4, seed=25))
    o = (u64 * 10)()
    sweeps, seeds = [], []
    out = (u64 * (5 * 2 * 64))()
    bw = u64()
    for nm, s, q in reads:
        R.ref_exact_sweep(h, s.encode(), q.encode(), o)
        sweeps.append(list(o))
        per = []
        for (sl, iv, off) in ((12, 6, 0), (10, 5, 2), (20, 9, 0)):
            if off > 0 and sl + off > len(s):
                continue
            ns = R.ref_seed_round(h, s.encode(), q.encode(), sl, iv, off, out, 5 * 2 * 64, C.byref(bw))
            per.append([sl, iv, off, ns, list(out[:ns * 2 * 5]), bw.value])
        seeds.append(per)
    g["reads"] = [[s, q] for _, s, q in reads]
    g["sweeps"] = sweeps
    g["seeds"] = seeds
    R.ref_close(h)
    return g


def dp_golden(refs):
    R = refshim(False)
    h = R.ref_open(os.path.join(HERE, "tiny_s").encode())
    rnd = random.Random(5)
    g = refs[0][1].replace("N", "A")
    probs = []
    for t in range(40):
        rows = rnd.choice([5, 9, 17, 30, 33, 50, 64, 65, 100, 128, 150, 200, 250])
        maxgap = rnd.choice([0, 3, 15])
        cols = rows + 4 * maxgap if maxgap else rows + rnd.randint(0, 5)
        pos = rnd.randrange(0, len(g) - cols - 2)
        window = g[pos:pos + cols + 1]
        rd = list(window[2 * maxgap:2 * maxgap + rows])
        for i in range(rows):
            r = rnd.random()
            if r < 0.03:
                rd[i] = rnd.choice("ACGT")
            elif r < 0.04:
                rd[i] = "N"
        if rnd.random() < 0.5 and rows > 40:
            k = rnd.randrange(10, rows - 10)
            del rd[k]
            rd.append(rnd.choice("ACGT"))
        if rnd.random() < 0.3 and rows > 40:
            k = rnd.randrange(10, rows - 10)
            rd.insert(k, rnd.choice("ACGT"))
            rd = rd[:rows]
        rd = "".join(rd)
        qu = "".join(rnd.choice("GGG?5-I#") for _ in range(rows))
        w = list(window)
        if rnd.random() < 0.3:
            w[rnd.randrange(len(w))] = "N"
        w = "".join(w)
        rf = bytes(1 << "ACGTN".index(c) for c in w)
        H = C.create_string_buffer(rows * cols)
        E = C.create_string_buffer(rows * cols)
        F = C.create_string_buffer(rows * cols)
        flag = C.c_int()
        best = R.ref_sw_fill_ee_u8(h, rd.encode(), qu.encode(), rf, cols, -250, H, E, F, C.byref(flag))
        assert flag.value == 0
        probs.append({"rd": rd, "qu": qu, "rf": w[:cols], "rows": rows, "cols": cols, "best": best,
                      "sha": sha(H.raw + E.raw + F.raw)})
    R.ref_close(h)
    return probs


def rng_golden():
    R = refshim(False)
    rnd = random.Random(77)
    out = []
    for seed in (0, 1, 12345, 0xffffffff, 0xc0000000, 0x9e3779b9):
        ops = bytes(rnd.randrange(5) for _ in range(300))
        buf = (C.c_uint32 * 300)()
        R.ref_rng_stream(seed, ops, 300, buf)
        out.append({"seed": seed, "ops": list(ops), "out": list(buf)})
    return out


def align_golden(refs):
    """End-to-end golden: the reference binary's SAM on a fixed read set against the tiny indexes."""
    rnd = random.Random(31)
    reads = (synth_reads(refs, 250, 60, seed=41) + synth_reads(refs, 120, 100, seed=42, sub=0.03, ins=0.004, dele=0.004)
             + synth_reads(refs, 60, 40, seed=43, n_rate=0.02) + synth_reads(refs, 40, 150, seed=44, sub=0.02)
             + synth_reads(refs, 30, 25, seed=45, len_jitter=10))
    reads += [("rand%d" % i, "".join(rnd.choice("ACGT") for _ in range(70)), "I" * 70) for i in range(20)]
    reads += [("polyA", "A" * 60, "I" * 60), ("oneN", "N", "I"), ("short", "ACG", "III")]
    rnd.shuffle(reads)
    fq = os.path.join(HERE, "align_reads.fq")
    write_fastq(fq, reads)
    for large in (False, True):
        exe = ref_bin("bowtie2-align-l" if large else "bowtie2-align-s")
        base = os.path.join(HERE, "tiny_l" if large else "tiny_s")
        for tag, args in (("sens", ["--sensitive"]), ("vfast", ["--very-fast"]), ("k5", ["-k", "5"]), ("local", ["--local"])):
            out = os.path.join(HERE, "align_golden_%s_%s.sam" % ("l" if large else "s", tag))
            subprocess.check_call([exe] + args + ["-x", base, "-U", fq, "-p", "1", "-S", out], stderr=subprocess.DEVNULL)
            lines = [l for l in open(out) if not l.startswith("@PG")]
            open(out, "w").writelines(lines)


def pair_golden(refs):
    """Paired-end golden: the reference binary's SAM for a fixed set of pairs (proper, over-long, wrong orientation, junk
    mate, different sequences) against the tiny indexes."""
    rnd = random.Random(77)
    comp = str.maketrans("ACGTN", "TGCAN")
    rc = lambda x: x.translate(comp)[::-1]
    r1, r2 = [], []
    for i in range(160):
        _, s = refs[rnd.randrange(len(refs))]
        L1, L2 = rnd.randrange(30, 101), rnd.randrange(30, 101)
        kind = rnd.random()
        frag = max(int(rnd.gauss(220, 30)), max(L1, L2) + 5)
        if kind < 0.06:
            frag = rnd.randrange(560, 900)
        frag = min(frag, len(s) - 2)
        p = rnd.randrange(0, len(s) - frag - 1)
        f = s[p:p + frag]
        m1, m2 = f[:L1], rc(f[-L2:])
        if kind > 0.94:
            m2 = rc(m2)
        if 0.88 < kind <= 0.94:
            m2 = "".join(rnd.choice("ACGT") for _ in range(L2))
        if 0.82 < kind <= 0.88:
            _, s2 = refs[rnd.randrange(len(refs))]
            q = rnd.randrange(0, len(s2) - L2 - 1)
            m2 = rc(s2[q:q + L2])
        if rnd.random() < 0.5:
            m1, m2 = m2, m1
        mut = lambda x: "".join(c if rnd.random() > 0.015 else rnd.choice("ACGT") for c in x)
        m1, m2 = mut(m1), mut(m2)
        r1.append(("p%d/1" % i, m1, "".join(rnd.choice("IIIIHH?5") for _ in m1)))
        r2.append(("p%d/2" % i, m2, "".join(rnd.choice("IIIIHH?5") for _ in m2)))
    f1, f2 = os.path.join(HERE, "pe_reads_1.fq"), os.path.join(HERE, "pe_reads_2.fq")
    write_fastq(f1, r1)
    write_fastq(f2, r2)
    for large in (False, True):
        exe = ref_bin("bowtie2-align-l" if large else "bowtie2-align-s")
        base = os.path.join(HERE, "tiny_l" if large else "tiny_s")
        for tag, args in (("sens", ["--sensitive"]), ("local", ["--local", "-k", "2"])):
            out = os.path.join(HERE, "pe_golden_%s_%s.sam" % ("l" if large else "s", tag))
            subprocess.check_call([exe] + args + ["-x", base, "-1", f1, "-2", f2, "-p", "1", "-S", out], stderr=subprocess.DEVNULL)
            lines = [l for l in open(out) if not l.startswith("@PG")]
            open(out, "w").writelines(lines)


if __name__ == "__main__":
    refs = build_tiny()
    if len(sys.argv) > 1 and sys.argv[1] == "pairs":      # only the paired-end fixtures
        pair_golden(refs)
        sys.exit(0)
    align_golden(refs)
    pair_golden(refs)
    for large in (False, True):
        with open(os.path.join(HERE, "fm_golden_%s.json" % ("l" if large else "s")), "w") as f:
            json.dump(fm_golden(large, refs), f, separators=(",", ":"))
    with open(os.path.join(HERE, "dp_golden.json"), "w") as f:
        json.dump(dp_golden(refs), f, separators=(",", ":"))
    with open(os.path.join(HERE, "rng_golden.json"), "w") as f:
        json.dump(rng_golden(), f, separators=(",", ":"))
    print("golden vectors written to", HERE)

"""Index builder: the files written must equal bowtie2-build's byte for byte.

CPU (`-m "not gpu"`): the builder's logic compiled against std:: primitives (tests/hostsim/build_hostsim.cpp, test-only)
vs the committed golden index of tests/golden/tiny.fa (built by the reference) and, where oracle/_ref is present, vs
the reference builder on generated FASTA files with the edge cases its parser has (empty and all-N sequences, leading /
trailing / interior N runs, IUPAC codes, lower case, blank lines, 1-base sequences, texts shorter than the ftab width,
exact repeats and homopolymers that need many doubling rounds, non-default -o/-t, the 64-bit-position code path).
GPU (`-m gpu`): the product executables bowtie2_amd/bin/bowtie2-build-{s,l} (libbt2g.so: rocPRIM sort + HIP kernels)
against the same references, plus the in-memory C-ABI entry point."""
import filecmp
import os
import random
import subprocess

import pytest

from bt2test import CACHE_DIR, have_ref, ref_bin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
HS = os.path.join(ROOT, "tests", "hostsim")
SUFFIXES = ["1", "2", "3", "4", "rev.1", "rev.2"]


def gen_fasta(path, seed, kind):
    r = random.Random(seed)
    with open(path, "w") as f:
        for s in range(r.randint(1, 5)):
            if kind == "weird" and r.random() < 0.2:
                f.write(">empty%d\n" % s)
                continue
            L = r.choice([1, 5, 9, 10, 11, 30, 100, 1000, 5000]) if kind != "big" else r.randint(20000, 80000)
            seq = [r.choice("ACGT") for _ in range(L)]
            if kind in ("rep", "weird", "big"):
                for _ in range(r.randint(1, 4)):            # exact repeats: long shared prefixes
                    ln = r.randint(1, max(1, L // 3)); a = r.randint(0, L - ln); b = r.randint(0, L - ln)
                    seq[b:b + ln] = seq[a:a + ln]
                if r.random() < 0.5:                          # homopolymer
                    ln = r.randint(1, max(1, L // 4)); a = r.randint(0, L - ln); seq[a:a + ln] = r.choice("ACGT") * ln
            if kind in ("weird", "big"):
                for _ in range(r.randint(0, 4)):
                    ln = r.randint(1, max(1, L // 10)); a = r.randint(0, L - ln); seq[a:a + ln] = [r.choice("NNNNRYK-n")] * ln
                if r.random() < 0.3:
                    seq[:r.randint(1, 5)] = "N" * r.randint(1, 5)
                if r.random() < 0.3:
                    seq[-3:] = "NNN"
            s_ = "".join(seq)
            if kind == "weird" and r.random() < 0.3:
                s_ = s_.lower()
            f.write(">seq%d some description %d\n" % (s, seed))
            w = r.choice([60, 80, 7, 1000000])
            for i in range(0, len(s_), w):
                f.write(s_[i:i + w] + "\n")
            if kind == "weird" and r.random() < 0.3:
                f.write("\n")


def same_files(a, b, ext):
    return [s for s in SUFFIXES if not filecmp.cmp("%s.%s.%s" % (a, s, ext), "%s.%s.%s" % (b, s, ext), shallow=False)]


def differential(builder_s, builder_l, tmp, seeds, extra_ok=True):
    """builder_* : argv prefix of the builder under test for .bt2 / .bt2l"""
    res = {"ok": 0, "bothfail": 0}
    for seed in seeds:
        for kind in ("plain", "rep", "weird"):
            for large in (False, True):
                fa = os.path.join(tmp, "g_%s_%d.fa" % (kind, seed))
                gen_fasta(fa, seed, kind)
                extra = []
                if seed % 4 == 1:
                    extra = ["-o", "3", "-t", "8"]
                if seed % 4 == 2:
                    extra = ["-t", "4"]
                mine_extra = list(extra)
                if extra_ok and seed % 3 == 0 and large:
                    mine_extra.append("--idx64")
                ext = "bt2l" if large else "bt2"
                a = subprocess.run([ref_bin("bowtie2-build-l" if large else "bowtie2-build-s"), "-q"] + extra + [fa, fa + ".ref"], capture_output=True, text=True)
                b = subprocess.run((builder_l if large else builder_s) + ["-q"] + mine_extra + [fa, fa + ".mine"], capture_output=True, text=True)
                if a.returncode != 0 and b.returncode != 0:
                    res["bothfail"] += 1
                    continue
                assert a.returncode == 0 and b.returncode == 0, (fa, a.returncode, b.returncode, a.stderr[-300:], b.stderr[-300:])
                assert same_files(fa + ".ref", fa + ".mine", ext) == [], (fa, large, extra)
                res["ok"] += 1
    return res


# ------------------------------------------------------------------ CPU ----
@pytest.fixture(scope="module")
def build_hostsim():
    exe = os.path.join(HS, "build_hostsim")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HS, "build_hostsim.cpp"), "-lz"])
    return exe


@pytest.mark.parametrize("large", [False, True])
def test_host_twin_reproduces_golden_index(build_hostsim, tmp_path, large):
    base = str(tmp_path / "tiny")
    # the golden index was built with --ftabchars 5 --offrate 3 (tests/golden/make_golden.py)
    subprocess.check_call([build_hostsim, "-q", "--ftabchars", "5", "--offrate", "3"] + (["--large-index"] if large else []) + [os.path.join(GOLD, "tiny.fa"), base])
    assert same_files(base, os.path.join(GOLD, "tiny_l" if large else "tiny_s"), "bt2l" if large else "bt2") == []


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
def test_host_twin_vs_reference_builder(build_hostsim, tmp_path):
    res = differential([build_hostsim], [build_hostsim, "--large-index"], str(tmp_path), range(10))
    assert res["ok"] >= 50


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
def test_host_twin_cmdline_sequences(build_hostsim, tmp_path):
    seqs = "ACGTTGCANNACGT,GGGGGGGGGGGGGGGGGGGGGGGG,A,NNNN,ACGTACGTACGTACGTACGTAC"
    for large in (False, True):
        a, b = str(tmp_path / "ref"), str(tmp_path / "mine")
        subprocess.check_call([ref_bin("bowtie2-build-l" if large else "bowtie2-build-s"), "-q", "-c", seqs, a], stdout=subprocess.DEVNULL)
        subprocess.check_call([build_hostsim, "-q", "-c"] + (["--large-index"] if large else []) + [seqs, b])
        assert same_files(a, b, "bt2l" if large else "bt2") == []


def test_rejects_bad_input(build_hostsim, tmp_path):
    p = tmp_path / "notfasta.txt"
    p.write_text("hello\nworld\n")
    assert subprocess.run([build_hostsim, "-q", str(p), str(tmp_path / "x")], capture_output=True).returncode != 0
    q = tmp_path / "onlyn.fa"
    q.write_text(">a\nNNNNNN\n")
    assert subprocess.run([build_hostsim, "-q", str(q), str(tmp_path / "y")], capture_output=True).returncode != 0


def test_failed_build_leaves_no_index_files(build_hostsim, tmp_path):
    """a gzip'ed FASTA cut short (or corrupted) must fail the build before anything is written -- it used to end the input early and
    produce an index of the truncated genome with exit status 0; and a build that fails for any reason leaves no files behind
    (every file is written under a temporary name and renamed when the whole build has succeeded)"""
    import glob
    import gzip
    import random
    rng = random.Random(3)
    fa = ">chr1\n" + "\n".join("".join(rng.choice("ACGT") for _ in range(60)) for _ in range(4000)) + "\n"
    whole = tmp_path / "g.fa.gz"
    with gzip.open(whole, "wb") as g:
        g.write(fa.encode())
    data = whole.read_bytes()
    cut = tmp_path / "cut.fa.gz"
    cut.write_bytes(data[:len(data) // 2])
    bad = tmp_path / "bad.fa.gz"
    bad.write_bytes(data[:3000] + bytes(64) + data[3064:])
    ok = subprocess.run([build_hostsim, "-q", str(whole), str(tmp_path / "ok")], capture_output=True, text=True)
    assert ok.returncode == 0 and len(glob.glob(str(tmp_path / "ok.*"))) == 6
    for k, f in enumerate((cut, bad)):
        r = subprocess.run([build_hostsim, "-q", str(f), str(tmp_path / ("x%d" % k))], capture_output=True, text=True)
        assert r.returncode != 0 and "corrupt or truncated" in r.stderr, r.stderr[-300:]
        assert glob.glob(str(tmp_path / ("x%d.*" % k))) == []
    # an output directory that does not exist: the failure comes when the first file is opened; nothing is left anywhere
    r = subprocess.run([build_hostsim, "-q", str(whole), str(tmp_path / "nodir" / "y")], capture_output=True, text=True)
    assert r.returncode != 0
    assert not os.path.exists(tmp_path / "nodir")


# ------------------------------------------------------------------ GPU ----
BIN_S = os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-build-s")
BIN_L = os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-build-l")


@pytest.mark.gpu
@pytest.mark.parametrize("large", [False, True])
def test_gpu_builder_reproduces_golden_index(tmp_path, large):
    base = str(tmp_path / "tiny")
    p = subprocess.run([BIN_L if large else BIN_S, "--ftabchars=5", "--offrate", "3", os.path.join(GOLD, "tiny.fa"), base], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "MI355X" in p.stderr
    assert same_files(base, os.path.join(GOLD, "tiny_l" if large else "tiny_s"), "bt2l" if large else "bt2") == []


@pytest.mark.gpu
@pytest.mark.skipif(not have_ref(), reason="oracle/_ref did not travel")
def test_gpu_builder_vs_reference_builder(tmp_path):
    res = differential([BIN_S], [BIN_L], str(tmp_path), range(6), extra_ok=False)
    assert res["ok"] >= 30
    # a genome-like case: 3 Mbp, planted diverged + exact repeats, N stretches; both widths
    fa = str(tmp_path / "big.fa")
    gen_fasta(fa, 4242, "big")
    r = random.Random(9)
    with open(fa, "a") as f:
        fam = "".join(r.choice("ACGT") for _ in range(300))
        s = []
        for _ in range(3000):
            s.append("".join(r.choice("ACGT") for _ in range(700)))
            s.append("".join(c if r.random() > 0.1 else r.choice("ACGT") for c in fam))
        s = "".join(s)
        s = s + s[100000:160000]                      # a 60 kbp exact duplication: ~11 doubling rounds
        f.write(">chrBig\n")
        for i in range(0, len(s), 80):
            f.write(s[i:i + 80] + "\n")
    for large in (False, True):
        ext = "bt2l" if large else "bt2"
        subprocess.check_call([ref_bin("bowtie2-build-l" if large else "bowtie2-build-s"), "-q", "--threads", "8", fa, fa + ".ref"], stdout=subprocess.DEVNULL)
        # .bt2l run: also through the 64-bit-position code path (two-pass sort per doubling round)
        env = dict(os.environ, BT2G_BUILD_FORCE_IDX64="1") if large else None
        p = subprocess.run([BIN_L if large else BIN_S, fa, fa + ".mine"], capture_output=True, text=True, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        assert same_files(fa + ".ref", fa + ".mine", ext) == []


@pytest.mark.gpu
def test_gpu_builder_in_memory_entry(tmp_path):
    """bt2g_index_build_mem (the bench uses it): same files as the executable run on the equivalent FASTA."""
    import ctypes as C
    import bowtie2_amd as b
    r = random.Random(3)
    seqs = ["".join(r.choice("ACGTN" if i == 1 else "ACGT") for _ in range(5000 + 777 * i)) for i in range(3)]
    names = ["chr%d" % i for i in range(3)]
    fa = tmp_path / "m.fa"
    fa.write_text("".join(">%s\n%s\n" % (n, s) for n, s in zip(names, seqs)))
    for large in (False, True):
        ext = "bt2l" if large else "bt2"
        st = b.build_index_mem(names, [s.encode() for s in seqs], str(tmp_path / "mem"), large=large)
        assert st.len == sum(len(s) - s.count("N") for s in seqs)
        p = subprocess.run([BIN_L if large else BIN_S, "-q", str(fa), str(tmp_path / "exe")], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-2000:]
        assert same_files(str(tmp_path / "mem"), str(tmp_path / "exe"), ext) == []

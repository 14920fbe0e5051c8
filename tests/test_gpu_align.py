"""End-to-end parity of the fused per-read worker on the GPU: the drop-in binary
(bowtie2_amd/bin/bowtie2-align-s -> libbt2g.so -> k_align_reads) must write SAM byte-identical to the
reference's, on the committed golden read set and -- where oracle/_ref travelled along -- on a larger
repeat-rich synthetic genome across presets and both index widths."""
import os
import random
import subprocess

import pytest

pytestmark = pytest.mark.gpu

from bt2test import (CACHE_DIR, build_index, have_ref, ref_bin, synth_genome, synth_reads, write_fasta, write_fastq)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-align-s")


def run_ours(args, timeout=300):
    p = subprocess.run([EXE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    return [l for l in p.stdout.splitlines() if not l.startswith("@PG")], p.stderr


@pytest.mark.parametrize("idx,tag,args", [("tiny_s", "s_sens", ["--sensitive"]), ("tiny_s", "s_vfast", ["--very-fast"]),
                                           ("tiny_l", "l_sens", ["--sensitive"]), ("tiny_l", "l_vfast", ["--very-fast"]),
                                           ("tiny_s", "s_k5", ["-k", "5"]), ("tiny_l", "l_k5", ["-k", "5"]),
                                           ("tiny_s", "s_local", ["--local"]), ("tiny_l", "l_local", ["--local"])])
def test_golden_sam(idx, tag, args):
    got, err = run_ours(args + ["-x", os.path.join(GOLD, idx), "-U", os.path.join(GOLD, "align_reads.fq")])
    want = open(os.path.join(GOLD, "align_golden_%s.sam" % tag)).read().splitlines()
    assert "Warning" not in err
    assert got == want


@pytest.mark.parametrize("idx,tag,args", [("tiny_s", "s_sens", ["--sensitive"]), ("tiny_l", "l_sens", ["--sensitive"]),
                                           ("tiny_s", "s_local", ["--local", "-k", "2"]), ("tiny_l", "l_local", ["--local", "-k", "2"])])
def test_golden_sam_paired(idx, tag, args):
    got, err = run_ours(args + ["-x", os.path.join(GOLD, idx), "-1", os.path.join(GOLD, "pe_reads_1.fq"), "-2", os.path.join(GOLD, "pe_reads_2.fq")])
    want = open(os.path.join(GOLD, "pe_golden_%s.sam" % tag)).read().splitlines()
    assert "Warning" not in err
    assert got == want


def repeat_genome():
    rnd = random.Random(5)
    refs = synth_genome(n_refs=3, total=200000, seed=21)
    elem = "".join(rnd.choice("ACGT") for _ in range(300))
    out = []
    for name, s in refs:
        s = list(s)
        for _ in range(25):
            p = rnd.randrange(0, len(s) - 400)
            s[p:p + 300] = [c if rnd.random() > 0.02 else rnd.choice("ACGT") for c in elem]
        p = rnd.randrange(0, len(s) - 400)
        s[p:p + 200] = list("A" * 200)
        p = rnd.randrange(0, len(s) - 400)
        s[p:p + 200] = list("AC" * 100)
        out.append((name, "".join(s)))
    reads = (synth_reads(out, 2500, 100, seed=1) + synth_reads(out, 1200, 150, seed=2, sub=0.03, ins=0.005, dele=0.005)
             + synth_reads(out, 600, 50, seed=3, n_rate=0.01) + synth_reads(out, 500, 250, seed=4, sub=0.02)
             + synth_reads(out, 200, 30, seed=5, len_jitter=12))
    reads += [("rand%d" % i, "".join(rnd.choice("ACGT") for _ in range(100)), "I" * 100) for i in range(100)]
    for i in range(150):
        p = rnd.randrange(0, 200)
        s = "".join(c if rnd.random() > 0.01 else rnd.choice("ACGT") for c in elem[p:p + 100])
        reads.append(("el%d" % i, s, "".join(rnd.choice("GGG?5-") for _ in s)))
    reads += [("polyA", "A" * 100, "I" * 100), ("acac", "AC" * 50, "I" * 100)]
    rnd.shuffle(reads)
    return out, reads


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
@pytest.mark.parametrize("large", [False, True], ids=["bt2", "bt2l"])
def test_differential_vs_reference_binary(large):
    d = os.path.join(CACHE_DIR, "rep_%s" % ("l" if large else "s"))
    os.makedirs(d, exist_ok=True)
    refs, reads = repeat_genome()
    fa, fq, base = os.path.join(d, "rep.fa"), os.path.join(d, "rep.fq"), os.path.join(d, "rep")
    write_fasta(fa, refs)
    write_fastq(fq, reads)
    build_index(fa, base, large)
    ref_exe = ref_bin("bowtie2-align-l" if large else "bowtie2-align-s")
    for args in (["--sensitive"], ["--very-sensitive"], ["--very-fast"], ["--sensitive", "--norc"], ["--nofw"], ["-k", "5"],
                 ["-k", "20", "--very-fast"], ["--local"], ["--very-fast-local"], ["--very-sensitive-local", "-k", "3"],
                 ["-N", "1"], ["-N", "1", "-L", "12", "-i", "C,6,0", "-k", "3"], ["-N", "1", "--very-sensitive-local"]):
        rs = os.path.join(d, "ref.sam")
        subprocess.check_call([ref_exe] + args + ["-x", base, "-U", fq, "-p", "8", "--reorder", "-S", rs], stderr=subprocess.DEVNULL)
        want = [l.rstrip("\n") for l in open(rs) if not l.startswith("@PG")]
        got, err = run_ours(args + ["-x", base, "-U", fq])
        assert "Warning" not in err
        assert len(got) == len(want)
        bad = [i for i in range(len(got)) if got[i] != want[i]]
        assert not bad, (args, len(bad), want[bad[0]], got[bad[0]])


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
@pytest.mark.parametrize("large", [False, True], ids=["bt2", "bt2l"])
def test_long_reads_16bit_dp(large):
    """Reads of 300-512 bp: above 423 bp the default --score-min drops below -254 and the reference switches to its
    16-bit DP kernels (aligner_sw.cpp:517), with a different RNG reseeding protocol in nextAlignment."""
    d = os.path.join(CACHE_DIR, "long_%s" % ("l" if large else "s"))
    os.makedirs(d, exist_ok=True)
    refs, _ = repeat_genome()
    reads = (synth_reads(refs, 300, 500, seed=11, sub=0.02, ins=0.002, dele=0.002) + synth_reads(refs, 300, 460, seed=12, sub=0.04, ins=0.004, dele=0.004)
             + synth_reads(refs, 200, 430, seed=13, sub=0.01, len_jitter=80) + synth_reads(refs, 100, 500, seed=14, sub=0.08)
             + synth_reads(refs, 100, 424, seed=15, n_rate=0.01))
    fa, fq, base = os.path.join(d, "rep.fa"), os.path.join(d, "long.fq"), os.path.join(d, "rep")
    write_fasta(fa, refs)
    write_fastq(fq, reads)
    build_index(fa, base, large)
    ref_exe = ref_bin("bowtie2-align-l" if large else "bowtie2-align-s")
    for args in (["--sensitive"], ["--very-sensitive"], ["-k", "3"], ["--local"]):
        rs = os.path.join(d, "ref.sam")
        subprocess.check_call([ref_exe] + args + ["-x", base, "-U", fq, "-p", "8", "--reorder", "-S", rs], stderr=subprocess.DEVNULL)
        want = [l.rstrip("\n") for l in open(rs) if not l.startswith("@PG")]
        got, err = run_ours(args + ["-x", base, "-U", fq])
        assert "Warning" not in err
        bad = [i for i in range(len(got)) if got[i] != want[i]]
        assert len(got) == len(want) and not bad, (args, len(bad), want[bad[0]][:300], got[bad[0]][:300])


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
@pytest.mark.parametrize("large", [False, True], ids=["bt2", "bt2l"])
def test_run_to_run_determinism(large):
    """The per-read work counters (BW ops, extension lengths, DP/backtrace counts) must be identical
    across repeated runs -- a wave-level race would show up here long before it changes a SAM line."""
    d = os.path.join(CACHE_DIR, "rep_%s" % ("l" if large else "s"))
    os.makedirs(d, exist_ok=True)
    refs, reads = repeat_genome()
    fa, fq, base = os.path.join(d, "rep.fa"), os.path.join(d, "rep.fq"), os.path.join(d, "rep")
    if not os.path.exists(fq):
        write_fasta(fa, refs)
        write_fastq(fq, reads)
        build_index(fa, base, large)
    outs = set()
    for _ in range(8):
        p = subprocess.run([EXE, "--met", "-x", base, "-U", fq, "-S", "/dev/null"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert p.returncode == 0
        outs.add("\n".join(l for l in p.stderr.splitlines() if l.startswith("MET")))
    assert len(outs) == 1


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
@pytest.mark.parametrize("large", [False, True], ids=["bt2", "bt2l"])
@pytest.mark.parametrize("args", [["--sensitive"], ["--very-sensitive"], ["--local"], ["-k", "4", "--very-fast"]])
def test_bulk_fm_kernels_equal_inline_search(large, args):
    """Stage check of the batch pre-computation kernels (k_exact_sweep, k_one_mm, k_seed_search_exact incl. the re-seeding rounds,
    k_extend_hits incl. the text-comparison form and the cached offsets): with BT2G_NO_PRECOMP=1 the worker computes every
    one of those phases itself, inline (the code path the CPU twin pins to the reference).  SAM and the per-read work
    counters that do not depend on who did the search (DPs, backtraces, iterations, alignments found, extension lengths)
    must be identical on a repeat-rich genome -- a wrong range, extension or offset from a bulk kernel shows up here."""
    d = os.path.join(CACHE_DIR, "rep_%s" % ("l" if large else "s"))
    os.makedirs(d, exist_ok=True)
    refs, reads = repeat_genome()
    fa, fq, base = os.path.join(d, "rep.fa"), os.path.join(d, "rep.fq"), os.path.join(d, "rep")
    if not os.path.exists(fq):
        write_fasta(fa, refs)
        write_fastq(fq, reads)
        build_index(fa, base, large)
    outs = []
    for env in (None, dict(os.environ, BT2G_NO_PRECOMP="1")):
        p = subprocess.run([EXE, "--met"] + args + ["-x", base, "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-1000:]
        met = []
        for l in p.stderr.splitlines():
            if l.startswith("MET"):
                f = dict(kv.split("=") for kv in l.split("\t")[2].split())
                met.append((l.split("\t")[1], f["iters"], f["dps"], f["ugs"], f["bt"], f["nalns"], f["extl"], f["extr"], f["red"]))
        outs.append(([l for l in p.stdout.splitlines() if not l.startswith("@PG")], met))
    assert outs[0][0] == outs[1][0]
    assert outs[0][1] == outs[1][1]


def test_two_device_workers_keep_input_order():
    """--gpu a,b runs one worker (context + index replica) per listed device and deals batches to whichever is free;
    the SAM must come out in input order and unchanged.  A single-GPU box lists its device twice."""
    base = os.path.join(GOLD, "tiny_s")
    fq = os.path.join(GOLD, "align_reads.fq")
    want = open(os.path.join(GOLD, "align_golden_s_sens.sam")).read().splitlines()
    for extra in (["--gpu", "0,0", "--batch", "7", "-p", "3"], ["--gpu", "0,0,0", "--batch", "64"], ["--batch", "1"]):
        got, err = run_ours(["--sensitive"] + extra + ["-x", base, "-U", fq])
        assert got == want, extra


def test_sharded_driver_on_gpu(tmp_path):
    """bowtie2_amd.mgpu with the product executable as the engine: two ranks (both on this box's one GPU, so the process
    group is gloo; on an N-GPU node it is RCCL) each align their --shard of the input; the merged SAM must equal the golden
    SAM recorded from the reference, and the merged summary the single-process one."""
    import socket
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    common = ["--sensitive", "--batch", "64", "--gpu", "0", "-x", os.path.join(GOLD, "tiny_s"), "-U", os.path.join(GOLD, "align_reads.fq")]
    one, one_err = run_ours(common)
    out = tmp_path / "merged.sam"
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   GLOO_SOCKET_IFNAME="lo", PYTHONPATH=ROOT)
        procs.append(subprocess.Popen([sys.executable, "-m", "bowtie2_amd.mgpu", "--backend", "gloo", "--"] + common + ["-S", str(out)],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    errs = []
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-2000:]
        errs.append(se)
    got = [l for l in open(out).read().splitlines() if not l.startswith("@PG")]
    assert got == open(os.path.join(GOLD, "align_golden_s_sens.sam")).read().splitlines()
    assert got == one
    summary = lambda t: [l for l in t.splitlines() if "aligned" in l or "reads; of these" in l or "were unpaired" in l or "overall alignment rate" in l]
    assert len(summary(one_err)) == 6 and summary(errs[0]) == summary(one_err)

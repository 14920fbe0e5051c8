"""A suffix-array sample too sparse for the resident layout must be REFUSED, never read wrong.

The reference samples the suffix array by row (bt2_idx.h: (row & offMask) == row), so the LF walk from a row to the next sampled row has a
geometric length with mean 2^offRate -- no upper bound.  The device's full suffix array keeps that step count in 16 bits next to the offset
(joff_pack, bt2g_device.hpp); a row further than 65534 steps from a sample cannot be represented and its seed hits would vanish without a
word.  bt2g_index_load (and the CPU twin's loader) count such rows while they build the array and fail with a message.  An index whose
sample is sparse but whose walks all fit (--offrate 10 here) loads and aligns exactly as the reference does."""
import os
import subprocess

import pytest

from bt2test import CACHE_DIR, ROOT, build_hostsim, ref_bin, synth_genome, synth_reads, write_fasta, write_fastq

HS = os.path.join(ROOT, "tests", "hostsim")


def sparse_index(off_rate):
    d = os.path.join(CACHE_DIR, "sparse_sa_o%d" % off_rate)
    base = os.path.join(d, "idx")
    refs = synth_genome(1, 1500000, 11, n_frac=0.0)
    if not os.path.exists(base + ".rev.2.bt2"):
        os.makedirs(d, exist_ok=True)
        write_fasta(os.path.join(d, "genome.fa"), refs)
        subprocess.check_call([ref_bin("bowtie2-build-s"), "-q", "-o", str(off_rate), os.path.join(d, "genome.fa"), base], stdout=subprocess.DEVNULL)
    fq = os.path.join(d, "reads.fq")
    if not os.path.exists(fq):
        write_fastq(fq, synth_reads(refs, 300, 80, seed=3))
    return base, fq


@pytest.fixture(scope="module")
def hostsim():
    return build_hostsim(os.path.join(HS, "hostsim"))


def test_cpu_twin_refuses_walks_longer_than_the_step_field(hostsim):
    base, fq = sparse_index(15)      # 1.5 Mbp, one row in 32 768 sampled: every eighth row is further than 65 534 steps from a sample
    p = subprocess.run([hostsim, "-x", base, "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode != 0 and "LF steps from a sampled row" in p.stderr, p.stderr[-500:]


def test_cpu_twin_sparse_but_representable_sample_matches_reference(hostsim):
    base, fq = sparse_index(10)
    want = subprocess.run([ref_bin("bowtie2-align-s"), "-x", base, "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True).stdout
    got = subprocess.run([hostsim, "-x", base, "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True).stdout
    strip = lambda t: [l for l in t.splitlines() if not l.startswith("@PG")]
    assert strip(got) == strip(want)


@pytest.mark.gpu
def test_device_loader_refuses_walks_longer_than_the_step_field():
    import bowtie2_amd as b
    base, fq = sparse_index(15)
    ctx = b.Context(0)
    try:
        with pytest.raises(b.Bt2gError) as ei:
            ctx.load_index(base)
        assert "LF steps from a sampled row" in str(ei.value)
    finally:
        ctx.close()
    exe = os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-align-s")
    p = subprocess.run([exe, "-x", base, "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode != 0 and "LF steps from a sampled row" in p.stderr


@pytest.mark.gpu
def test_device_sparse_but_representable_sample_matches_reference():
    base, fq = sparse_index(10)
    exe = os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-align-s")
    want = subprocess.run([ref_bin("bowtie2-align-s"), "-x", base, "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True).stdout
    got = subprocess.run([exe, "-x", base, "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True).stdout
    strip = lambda t: [l for l in t.splitlines() if not l.startswith("@PG")]
    assert strip(got) == strip(want)

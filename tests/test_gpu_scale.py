"""GPU parity at scale: the product executables against the unmodified reference (oracle/_ref, which
travels to the GPU box) on inputs the size of a real run, both index widths:
  * BASELINE.json config 1 and the paired example: tests/golden/example (phage lambda, the reference's own 10 000 read pairs,
    30-250 bp, Ns included), index built by the product's GPU builder;
  * 200 000 x 150 bp reads against a 32 Mbp hg38-like repeat-rich genome (the bench generator), .bt2 and .bt2l.
SAM must be byte-identical (minus @PG)."""
import gzip
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

from bt2test import CACHE_DIR, have_ref, ref_bin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "tests", "golden", "example")
BIN = os.path.join(ROOT, "bowtie2_amd", "bin")


def sam(cmd, timeout=900):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    return [l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG")], p.stderr.decode(errors="replace")


@pytest.fixture(scope="module")
def lambda_idx(tmp_path_factory):
    d = tmp_path_factory.mktemp("lambda")
    out = {}
    for w in ("s", "l"):
        base = str(d / ("lambda_" + w))
        subprocess.check_call([os.path.join(BIN, "bowtie2-build-" + w), "-q", os.path.join(EX, "lambda_virus.fa"), base])
        out[w] = base
    for k in ("1", "2"):
        with gzip.open(os.path.join(EX, "reads_%s.fq.gz" % k), "rb") as f, open(str(d / ("reads_%s.fq" % k)), "wb") as g:
            shutil.copyfileobj(f, g)
    out["r1"], out["r2"] = str(d / "reads_1.fq"), str(d / "reads_2.fq")
    return out


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref did not travel")
@pytest.mark.parametrize("w", ["s", "l"])
@pytest.mark.parametrize("args", [["--sensitive"], ["--local"], ["--very-sensitive", "-k", "3"]])
def test_lambda_example_unpaired(lambda_idx, w, args):
    """config 1: example/reads/reads_1.fq vs lambda_virus"""
    a = args + ["-x", lambda_idx[w], "-U", lambda_idx["r1"]]
    want, _ = sam([ref_bin("bowtie2-align-" + w)] + a + ["-p", "8", "--reorder"])
    got, err = sam([os.path.join(BIN, "bowtie2-align-" + w)] + a)
    assert "Warning" not in err
    assert got == want


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref did not travel")
@pytest.mark.parametrize("w,extra", [("s", []), ("l", []), ("s", ["--local"]), ("l", ["-X", "800"]), ("s", ["--local", "--dovetail", "-X", "900"])])
def test_lambda_example_paired(lambda_idx, w, extra):
    """The reference's 10 000 example pairs.  With --local (the gap allowance of 250-bp mates is large), -X 800 or --dovetail the window in
    which a mate is looked for next to its partner is wider than the 1 100 columns a launch holds by default: the driver asks bt2g_align_batch
    for what the batch needs (bt2g_align_params::max_dp_cols, up to 2 176) -- 89 / 244 pairs of these runs used to be flagged."""
    a = extra + ["-x", lambda_idx[w], "-1", lambda_idx["r1"], "-2", lambda_idx["r2"]]
    want, _ = sam([ref_bin("bowtie2-align-" + w)] + a + ["-p", "8", "--reorder"])
    got, err = sam([os.path.join(BIN, "bowtie2-align-" + w)] + a)
    assert "Warning" not in err
    assert got == want


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref did not travel")
@pytest.mark.parametrize("w", ["s", "l"])
def test_repeat_rich_genome_200k_reads(tmp_path, w):
    import torch
    sys.path.insert(0, ROOT)
    import bench
    import bowtie2_amd as b
    dev = torch.device("cuda", 0)
    G, lens = bench.synth_genome_gpu(32, 5, dev)
    base = str(tmp_path / "g32")
    bench.build_index_gpu(base, G, lens, w == "l", 0)
    n = 200000
    seq, qual = bench.synth_reads_gpu(G, n, 150, 77, dev)
    fq = str(tmp_path / "reads.fq")
    bench.write_fastq_fixed(fq, seq, qual, bench.read_names(0, n))
    del G
    torch.cuda.empty_cache()
    a = ["--sensitive", "-x", base, "-U", fq]
    ref_exe = ref_bin("bowtie2-align-%s-v256" % w)
    if not os.path.exists(ref_exe):
        ref_exe = ref_bin("bowtie2-align-" + w)
    want, _ = sam([ref_exe] + a + ["-p", str(bench.nproc()), "--reorder"])
    got, err = sam([os.path.join(BIN, "bowtie2-align-" + w)] + a)
    assert "Warning" not in err
    assert len(got) == len(want) and got == want

"""Reads of 513 ... 1 999 bp (first half).  The reference stays on its plain DP path up to 1 999 bp
(aligner_sw.cpp:514: checkpointing from 2 000) -- this build's worker stopped at 512 until round 5.  A batch whose longest read is above
512 bp now runs in the worker's long-read class (2 048 DP rows, 128 seed positions per strand, the 16-bit end-to-end fill with its per-row
state in scratch, the packed local fill with 8 / 16 rows per block).  On 240 of the reference's own example long reads against phage lambda
the SAM must equal the reference binary's, read by read, end to end and with --local, and nothing may be flagged (these simulated reads carry
~10 % errors: the class's result records hold 640 edits per alignment instead of 200, its candidate lists a million cells of a local window
instead of 65 536, its walks 1 344 edits).
Reads of 2 000 bp and more are refused."""
import gzip
import os
import subprocess

import pytest

from bt2test import CACHE_DIR, ROOT, build_hostsim, ref_bin

EX = os.path.join(ROOT, "tests", "golden", "example")
HS = os.path.join(ROOT, "tests", "hostsim")
BIN = os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-align-s")


def workload():
    d = os.path.join(CACHE_DIR, "long_reads_lambda")
    base, fq = os.path.join(d, "lambda"), os.path.join(d, "long.fq")
    if not os.path.exists(base + ".rev.2.bt2"):
        os.makedirs(d, exist_ok=True)
        subprocess.check_call([ref_bin("bowtie2-build-s"), "-q", os.path.join(EX, "lambda_virus.fa"), base], stdout=subprocess.DEVNULL)
    if not os.path.exists(fq):
        with gzip.open(os.path.join(EX, "longreads_513_1999.fq.gz"), "rb") as f, open(fq, "wb") as g:
            g.write(f.read())
    return base, fq


def by_read(text):
    d = {}
    for l in text.splitlines():
        if not l.startswith("@"):
            d.setdefault(l.split("\t", 1)[0], []).append(l)
    return d


def check(exe, args, product):
    base, fq = workload()
    want = by_read(subprocess.run([ref_bin("bowtie2-align-s")] + args + ["-x", base, "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True).stdout)
    p = subprocess.run([exe] + args + ["-x", base, "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=3000)
    warns = [l for l in p.stderr.splitlines() if l.startswith("Warning: read")]
    flagged = {l.split()[2].rstrip(":") for l in warns}
    assert p.returncode == (1 if flagged and product else 0), p.stderr[-800:]
    # nothing is flagged: the long-read class holds 640 edits per alignment (BT2G_MAX_EDITS_LONG), a million candidate cells per local DP window
    # (the example reads need up to ~400 000) and 1 344 edits in a local walk that has not ended yet
    assert not warns, warns[:3]
    got = by_read(p.stdout)
    assert set(got) == set(want) and len(want) == 240
    bad = [n for n in want if n not in flagged and got[n] != want[n]]
    assert not bad, (len(bad), bad[:3])
    aligned = sum(1 for n in want if n not in flagged and not int(want[n][0].split("\t")[1]) & 4)
    assert aligned > 120, aligned


@pytest.fixture(scope="module")
def hostsim():
    return build_hostsim(os.path.join(HS, "hostsim"))


@pytest.mark.parametrize("args", [["--sensitive"], ["--local"], ["--very-sensitive", "-k", "3"]])
def test_long_reads_hostsim(hostsim, args):
    check(hostsim, args, False)


def test_reads_of_2000_bp_are_refused(hostsim, tmp_path):
    base, _ = workload()
    fq = tmp_path / "r.fq"
    fq.write_text("@too_long\n%s\n+\n%s\n" % ("ACGT" * 500, "I" * 2000))
    p = subprocess.run([hostsim, "-x", base, "-U", str(fq)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode != 0 and "longer than 1999 bp" in p.stderr, p.stderr[-300:]


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["--sensitive"], ["--local"], ["--very-sensitive", "-k", "3"]])
def test_long_reads_gpu(args):
    check(BIN, args, True)


@pytest.mark.gpu
def test_long_and_short_batches_in_one_run_gpu(tmp_path):
    """short reads first, long ones behind them, batches of 64: the driver hands every batch to the class that holds it"""
    base, fq = workload()
    short = os.path.join(ROOT, "tests", "golden", "align_reads.fq")
    both = tmp_path / "both.fq"
    both.write_bytes(open(short, "rb").read() + open(fq, "rb").read())
    want = by_read(subprocess.run([ref_bin("bowtie2-align-s"), "-x", base, "-U", str(both)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True).stdout)
    p = subprocess.run([BIN, "--batch", "64", "-x", base, "-U", str(both)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=3000)
    flagged = {l.split()[2].rstrip(":") for l in p.stderr.splitlines() if l.startswith("Warning: read")}
    got = by_read(p.stdout)
    assert set(got) == set(want)
    assert not [n for n in want if n not in flagged and got[n] != want[n]]


@pytest.mark.gpu
def test_pairs_with_mates_in_the_16_bit_range_gpu():
    """Mates of 430-510 bp: the anchor's and the opposite mate's windows go through the 16-bit end-to-end kernel's arithmetic (minimum score
    below -254) -- on the band fill since round 6.  tools/long_pairs_check.py: four option sets against the reference binary."""
    p = subprocess.run(["python3", os.path.join(ROOT, "tools", "long_pairs_check.py"), "300", "13"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    assert p.returncode == 0, p.stdout[-1500:]
    assert p.stdout.count("differing 0, reads flagged 0") == 4, p.stdout[-1500:]


@pytest.mark.gpu
def test_reads_of_424_to_512_bp_short_read_classes_gpu(tmp_path):
    """Unpaired reads just above the 8-bit kernel's range (424-512 bp at the default threshold) run in the general class, not the long-read one:
    cut from the example long reads, against the reference binary."""
    base, fq = workload()
    lines = open(fq).read().split("\n")
    out = []
    for k in range(0, len(lines) - 3, 4):
        L = 424 + (k // 4 * 7) % 89
        if len(lines[k + 1]) >= L:
            out += [lines[k], lines[k + 1][:L], "+", lines[k + 3][:L]]
    cut = tmp_path / "cut.fq"
    cut.write_text("\n".join(out) + "\n")
    want = by_read(subprocess.run([ref_bin("bowtie2-align-s"), "-x", base, "-U", str(cut)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True).stdout)
    p = subprocess.run([BIN, "-x", base, "-U", str(cut)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=3000)
    assert p.returncode == 0 and "Warning: read" not in p.stderr, p.stderr[-500:]
    got = by_read(p.stdout)
    assert set(got) == set(want) and len(want) > 200
    assert not [n for n in want if got[n] != want[n]]

"""-a on reads with more alignments than a result record holds (64): flagged, never silently cut (tests/golden/all_hits/README.md)."""
import os
import subprocess

import pytest

from bt2test import build_index, build_hostsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "all_hits")
HS = os.path.join(ROOT, "tests", "hostsim")


def by_read(lines):
    d = {}
    for l in lines:
        if not l.startswith("@"):
            d.setdefault(l.split("\t", 1)[0], []).append(l)
    return d


def check(exe, tmp_path, want_rc):
    base = str(tmp_path / "g")
    build_index(os.path.join(GOLD, "genome.fa"), base, False)
    p = subprocess.run([exe, "-a", "-x", base, "-U", os.path.join(GOLD, "reads.fq")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == want_rc, p.stderr[-500:]
    flagged = sorted(l.split()[2].rstrip(":") for l in p.stderr.splitlines() if l.startswith("Warning: read"))
    assert flagged == ["r59", "r6"], p.stderr[-800:]
    want = by_read(open(os.path.join(GOLD, "reference_a.sam")).read().splitlines())
    got = by_read(p.stdout.splitlines())
    assert len(want["r6"]) == 74 and len(want["r59"]) == 70
    for name in want:
        if name not in ("r6", "r59"):
            assert got.get(name) == want[name], name


@pytest.fixture(scope="module")
def hostsim():
    exe = os.path.join(HS, "hostsim_allhits")
    build_hostsim(exe)
    return exe


def test_all_hits_beyond_record_capacity_is_flagged_hostsim(hostsim, tmp_path):
    check(hostsim, tmp_path, 0)      # the test-only host twin warns; only the product binary turns flagged reads into exit status 1


@pytest.mark.gpu
def test_all_hits_beyond_record_capacity_is_flagged_gpu(tmp_path):
    check(os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-align-s"), tmp_path, 1)

"""Host logic: the per-read worker's control code (compiled for the CPU by tests/hostsim, test-only)
must reproduce the reference's SAM byte for byte on the committed golden read set."""
import os
import subprocess

import pytest

from bt2test import build_hostsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
HS = os.path.join(ROOT, "tests", "hostsim")


@pytest.fixture(scope="module")
def hostsim():
    exe = os.path.join(HS, "hostsim")
    build_hostsim(exe)
    return exe


@pytest.mark.parametrize("idx,tag,args", [("tiny_s", "s_sens", ["--sensitive"]), ("tiny_s", "s_vfast", ["--very-fast"]),
                                           ("tiny_l", "l_sens", ["--sensitive"]), ("tiny_l", "l_vfast", ["--very-fast"]),
                                           ("tiny_s", "s_k5", ["-k", "5"]), ("tiny_l", "l_k5", ["-k", "5"]),
                                           ("tiny_s", "s_local", ["--local"]), ("tiny_l", "l_local", ["--local"])])
def test_sam_identical_to_reference(hostsim, idx, tag, args):
    p = subprocess.run([hostsim] + args + ["-x", os.path.join(GOLD, idx), "-U", os.path.join(GOLD, "align_reads.fq")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True)
    got = [l for l in p.stdout.splitlines() if not l.startswith("@PG")]
    want = open(os.path.join(GOLD, "align_golden_%s.sam" % tag)).read().splitlines()
    assert got == want
    assert "overall alignment rate" in p.stderr


@pytest.mark.parametrize("idx,tag,args", [("tiny_s", "s_sens", ["--sensitive"]), ("tiny_l", "l_sens", ["--sensitive"]),
                                           ("tiny_s", "s_local", ["--local", "-k", "2"]), ("tiny_l", "l_local", ["--local", "-k", "2"])])
def test_paired_sam_identical_to_reference(hostsim, idx, tag, args):
    p = subprocess.run([hostsim] + args + ["-x", os.path.join(GOLD, idx), "-1", os.path.join(GOLD, "pe_reads_1.fq"), "-2", os.path.join(GOLD, "pe_reads_2.fq")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True)
    got = [l for l in p.stdout.splitlines() if not l.startswith("@PG")]
    want = open(os.path.join(GOLD, "pe_golden_%s.sam" % tag)).read().splitlines()
    assert got == want
    assert "were paired; of these:" in p.stderr


@pytest.mark.parametrize("idx,tag", [("tiny_s", "s_local"), ("tiny_l", "l_local")])
def test_packed_local_fill_replayed_on_every_window(hostsim, idx, tag):
    """BT2G_CHECK_LOCAL_PK=1: every local DP window the twin fills is filled a second time the way the DEVICE does it -- 64 lanes side by side,
    two blocks of rows per lane, the packed 16-bit cell arithmetic of bowtie2_amd/csrc/bt2g_local_pk.hpp (the source the device kernel compiles),
    the lane-to-lane hand-over and the per-column bookkeeping of fill_local_pk -- and every predecessor byte somebody can look at, every H, the
    best score, lastsolcol and the saturation flag are compared with the scalar fill (which the golden SAM pins to the reference).  A difference
    aborts the run.  Unpaired reads and pairs (opposite-mate windows are wider and shorter reads start their blocks in other lanes)."""
    env = dict(os.environ, BT2G_CHECK_LOCAL_PK="1")
    runs = ((["--local", "-U", os.path.join(GOLD, "align_reads.fq")], "align_golden_%s.sam" % tag),
            (["--local", "-k", "2", "-1", os.path.join(GOLD, "pe_reads_1.fq"), "-2", os.path.join(GOLD, "pe_reads_2.fq")], "pe_golden_%s.sam" % tag))
    for args, gold in runs:
        p = subprocess.run([hostsim] + args + ["-x", os.path.join(GOLD, idx)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        assert p.returncode == 0, p.stderr[-1500:]
        assert "local pk check active" in p.stderr
        assert [l for l in p.stdout.splitlines() if not l.startswith("@PG")] == open(os.path.join(GOLD, gold)).read().splitlines()

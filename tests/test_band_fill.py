"""The band argument of the 8-bit end-to-end fill (bt2g_align.hpp, EeBand): cells outside the band of diagonals a valid alignment
can touch are never computed.  Property test on the host twin: with BT2G_HOST_FULL_RECT=1 the same fill covers every diagonal of
every DP rectangle; SAM (scores, CIGARs, XS:i, the RNG-dependent tie breaks) must not change, for gapped and repeat-rich reads, both
index widths, tight and loose score thresholds."""
import os
import subprocess

import pytest

from bt2test import CACHE_DIR, build_index, have_ref, write_fasta, write_fastq, build_hostsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HS = os.path.join(ROOT, "tests", "hostsim")


@pytest.fixture(scope="module")
def hostsim():
    exe = os.path.join(HS, "hostsim_band")
    build_hostsim(exe)
    return exe


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present (index builder)")
@pytest.mark.parametrize("large", [False, True], ids=["bt2", "bt2l"])
@pytest.mark.parametrize("args", [["--sensitive"], ["--very-sensitive", "--score-min", "L,-0.6,-1.2"], ["-k", "5", "--rdg", "2,1", "--rfg", "2,1"],
                                  ["--score-min", "C,-30", "--mp", "2,2"]], ids=["sens", "loose", "cheap_gaps", "tight"])
def test_band_fill_equals_full_rectangle_fill(hostsim, large, args):
    from test_gpu_align import repeat_genome      # the repeat-rich genome + 5 252 reads (30-250 bp, indels, Ns, poly-A) of the GPU parity tests
    d = os.path.join(CACHE_DIR, "rep_%s" % ("l" if large else "s"))
    os.makedirs(d, exist_ok=True)
    fa, fq, base = os.path.join(d, "rep.fa"), os.path.join(d, "rep.fq"), os.path.join(d, "rep")
    if not os.path.exists(fq):
        refs, reads = repeat_genome()
        write_fasta(fa, refs)
        write_fastq(fq, reads)
        build_index(fa, base, large)
    outs = []
    for full in (False, True):
        env = dict(os.environ)
        if full:
            env["BT2G_HOST_FULL_RECT"] = "1"
        else:
            env.pop("BT2G_HOST_FULL_RECT", None)
        p = subprocess.run([hostsim] + args + ["-x", base, "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900)
        assert p.returncode == 0, p.stderr[-500:]
        outs.append([l for l in p.stdout.splitlines() if not l.startswith("@PG")])
    assert len(outs[0]) > 1000
    assert outs[0] == outs[1]

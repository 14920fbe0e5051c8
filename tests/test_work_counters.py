"""Work parity, not only result parity: the reference's own --met-stderr totals (PerfMetrics, bt2_search.cpp:1968-2200) against the
worker's per-read counters (--met) summed over the same reads.  Compared are the counters whose definition is the same on both sides:
  ResReport      seed-hit elements taken by extendSeeds            == sum iters
  DP{8,16}ExDps  extension DP problems filled                      == sum dps
  RedundantSHit  elements skipped because their diagonal was seen  == sum red
  DP*ExBt        backtrace attempts (candidate cells tried)        == sum bt
  DP*ExGathSol   candidate cells gathered (host twin only)         == sum cands
  DP*ExBtCell    cells the backtraces walked (host twin only)      == sum btsteps
(AlBWOp / ResBWOp are not comparable: the worker counts the exact sweep and the 1-mismatch search under the same counter as the seed
search, and it re-walks a row's offset when a later seeding round samples the row again where the reference's seed cache remembers it.)"""
import collections
import os
import re
import subprocess

import pytest

from bt2test import CACHE_DIR, build_index, have_ref, ref_bin, write_fasta, write_fastq, build_hostsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HS = os.path.join(ROOT, "tests", "hostsim")


def workload(large):
    from test_gpu_align import repeat_genome
    d = os.path.join(CACHE_DIR, "rep_%s" % ("l" if large else "s"))
    os.makedirs(d, exist_ok=True)
    fa, fq, base = os.path.join(d, "rep.fa"), os.path.join(d, "rep.fq"), os.path.join(d, "rep")
    if not os.path.exists(fq):
        refs, reads = repeat_genome()
        write_fasta(fa, refs)
        write_fastq(fq, reads)
        build_index(fa, base, large)
    return base, fq


def reference_totals(base, fq, large, args):
    p = subprocess.run([ref_bin("bowtie2-align-l" if large else "bowtie2-align-s")] + args + ["-p", "1", "--met-stderr", "-x", base, "-U", fq, "-S", "/dev/null"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-500:]
    rows = [l.split("\t") for l in p.stderr.splitlines() if "\t" in l]
    head = rows[0]
    last = [r for r in rows[1:] if len(r) >= len(head) - 1][-1]      # the final line carries the totals of the run
    return dict(zip(head, last))


def our_totals(exe, base, fq, args):
    p = subprocess.run([exe] + args + ["--met", "-x", base, "-U", fq, "-S", "/dev/null"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-500:]
    tot = collections.Counter()
    n = 0
    for l in p.stderr.splitlines():
        if l.startswith("MET\t"):
            n += 1
            for k, v in re.findall(r"(\w+)=(\d+)", l):
                tot[k] += int(v)
    return n, tot


def check(exe, large, args, host_twin):
    base, fq = workload(large)
    ref = reference_totals(base, fq, large, args)
    n, ours = our_totals(exe, base, fq, args)
    both = lambda k: int(ref["DP8" + k]) + int(ref["DP16" + k])
    assert n == int(ref["Read"]) > 5000
    assert ours["iters"] == int(ref["ResReport"])
    assert ours["dps"] == both("ExDps")
    assert ours["red"] == int(ref["RedundantSHit"])
    assert ours["bt"] == both("ExBt")
    if host_twin:
        assert ours["cands"] == both("ExGathSol")
        assert ours["btsteps"] == both("ExBtCell")
    assert ours["dps"] > 5000 and ours["bt"] > 2000      # the workload exercised them


@pytest.fixture(scope="module")
def hostsim():
    exe = os.path.join(HS, "hostsim_met")
    build_hostsim(exe)
    return exe


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
@pytest.mark.parametrize("large,args", [(False, ["--sensitive"]), (True, ["--very-sensitive"])], ids=["bt2_sens", "bt2l_vsens"])
def test_work_counters_equal_reference_hostsim(hostsim, large, args):
    check(hostsim, large, args, True)


@pytest.mark.gpu
@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
@pytest.mark.parametrize("large,args", [(False, ["--sensitive"]), (True, ["--very-sensitive"])], ids=["bt2_sens", "bt2l_vsens"])
def test_work_counters_equal_reference_gpu(large, args):
    check(os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-align-l" if large else "bowtie2-align-s"), large, args, False)

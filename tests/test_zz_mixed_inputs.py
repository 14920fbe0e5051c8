"""Paired and unpaired inputs in one run (-1/-2 or --interleaved together with -U; PatternComposer, pat.cpp:225-420): the pair sources are
read to their end, then the unpaired files; SAM = the pairs' records followed by the unpaired reads', one summary in which a pair counts
as one read (AlnSink::printAlSumm, aln_sink.cpp:349-560).  Differential against the reference binary -- run with -p 1: with more threads
bowtie2 2.5.5 itself does not finish on such input.  FASTQ only (with FASTA the reference drops the first unpaired record after the
pairs, with BAM its -U source takes paired records; both combinations are refused here).  CPU: host-compiled worker; GPU: product binary,
where the switch from the pair kernel to the unpaired kernel happens inside one run of one device context."""
import os
import subprocess

import pytest

from bt2test import have_ref, ref_bin, build_hostsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
HS = os.path.join(ROOT, "tests", "hostsim")
EXE = os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-align-s")
M1, M2, FQ = (os.path.join(GOLD, n) for n in ("pe_reads_1.fq", "pe_reads_2.fq", "align_reads.fq"))

# -s/-u count within each source (every PatternSource numbers its own reads); a -u that ends the pairs ends the run (160 pairs in the fixture)
OPTION_SETS = [[], ["--local", "-k", "3"], ["--no-mixed", "--no-discordant"], ["--very-fast", "--no-unal"], ["-N", "1", "-L", "18", "--ff"],
               ["-u", "200"], ["-u", "100"], ["-u", "160"], ["-u", "159"], ["-s", "100", "-u", "61"], ["-s", "170", "-u", "20"], ["-s", "1000"]]


@pytest.fixture(scope="module")
def hostsim():
    exe = os.path.join(HS, "hostsim")
    build_hostsim(exe)
    return exe


def run(exe, args, extra=()):
    p = subprocess.run([exe] + args + list(extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-1500:]
    sam = [l for l in p.stdout.splitlines() if not l.startswith("@PG")]
    summ = [l for l in p.stderr.splitlines() if not l.startswith("Warning") and "amdgpu.ids" not in l]
    return sam, summ


def check(exe, idx, tmp, extra):
    ref = ref_bin("bowtie2-align-l" if idx.endswith("_l") else "bowtie2-align-s")
    base = os.path.join(GOLD, idx)
    inter = os.path.join(tmp, "inter.fq")
    l1, l2 = open(M1).read().splitlines(), open(M2).read().splitlines()
    open(inter, "w").write("".join("\n".join(l1[i:i + 4] + l2[i:i + 4]) + "\n" for i in range(0, len(l1), 4)))
    for opts in OPTION_SETS:
        for src in (["-1", M1, "-2", M2], ["--interleaved", inter]):
            a = opts + ["-x", base] + src + ["-U", FQ + "," + FQ]
            want = run(ref, a, ["-p", "1"])
            if not any(o in opts for o in ("-s", "-u")):
                assert any("were paired" in l for l in want[1]) and any("were unpaired" in l for l in want[1])
            assert run(exe, a, extra) == want, (opts, src[0])


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
@pytest.mark.parametrize("idx", ["tiny_s", "tiny_l"])
def test_mixed_inputs_match_reference_hostsim(hostsim, idx, tmp_path):
    check(hostsim, idx, str(tmp_path), [])


def test_mixed_inputs_refused_where_the_reference_misbehaves(hostsim, tmp_path):
    fa = os.path.join(str(tmp_path), "r.fa")
    open(fa, "w").write(">a\nACGTACGTACGTACGTACGTACGTAACC\n>b\nACGTACGTACGTACGTACGTACGTAACC\n")
    for a in (["-f", "-1", fa, "-2", fa, "-U", fa], ["-b", "--align-paired-reads", "-1", fa, "-2", fa, "-U", fa], ["--tab5", fa, "-1", M1, "-2", M2]):
        p = subprocess.run([hostsim] + a + ["-x", os.path.join(GOLD, "tiny_s")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert p.returncode != 0 and p.stdout == "", a


@pytest.mark.gpu
@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
def test_mixed_inputs_match_reference_gpu(tmp_path):
    check(EXE, "tiny_s", str(tmp_path), ["-p", "2"])
    # batches smaller than either input: several pair batches, the switch of kernels, several unpaired batches, two worker threads per stage
    check(EXE, "tiny_l", str(tmp_path), ["-p", "3", "--batch", "64"])

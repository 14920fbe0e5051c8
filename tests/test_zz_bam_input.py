"""Ingest and SAM options either side of the hot path, differential against the reference binary:
unaligned BAM input (-b: BAMPatternSource, pat.cpp:1249-1515) with --preserve-tags and --align-paired-reads,
--sam-append-comment (sam.h:415-463) and --soft-clipped-unmapped-tlen (aligner_result.h:894-909).
CPU: the host-compiled worker (same reader, parser and SAM writer as the product).  GPU: the product binary."""
import os
import struct
import subprocess

import pytest

from bt2test import bam_record, have_ref, ref_bin, write_bam, build_hostsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
HS = os.path.join(ROOT, "tests", "hostsim")
EXE = os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-align-s")


@pytest.fixture(scope="module")
def hostsim():
    exe = os.path.join(HS, "hostsim")
    build_hostsim(exe)
    return exe


def fastq(path):
    l = open(path).read().splitlines()
    return [(l[i][1:], l[i + 1], l[i + 3]) for i in range(0, len(l), 4)]


def run(exe, args):
    p = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-1500:]
    sam = [l for l in p.stdout.splitlines() if not l.startswith("@PG")]
    summ = [l for l in p.stderr.splitlines() if not l.startswith("Warning") and "amdgpu.ids" not in l]
    return sam, summ


def make_inputs(tmp):
    """an unpaired BAM (with mapped and paired records the reader must skip, tags of every BAM value type, records that straddle
    BGZF blocks), a paired BAM (mates interleaved with records to skip), and FASTQ/FASTA files whose names carry comments"""
    tags_all = (b"RGZgrp1\0" + b"NHC\x05" + b"XSs" + struct.pack("<h", -7) + b"XFf" + struct.pack("<f", 1.5) + b"XCAq" + b"XII" + struct.pack("<I", 4000000000)
                + b"Xcc\xfe" + b"XTS" + struct.pack("<H", 65000) + b"Xii" + struct.pack("<i", -123456) + b"XBBS" + struct.pack("<IHHH", 3, 1, 2, 3))
    recs, k = [], 0
    for n, s, q in fastq(os.path.join(GOLD, "align_reads.fq")):
        if not s or any(c not in "ACGTN" for c in s):
            continue
        k += 1
        flag = 4 if k % 5 else (0 if k % 10 else 4 | 1 | 0x40)
        recs.append(bam_record(n.split()[0][:200], flag, s, q, tags_all if k % 3 == 0 else b"BCZACGT\0" if k % 3 == 1 else b""))
    se = os.path.join(tmp, "se.bam")
    write_bam(se, recs, block=600)
    recs = []
    for k, ((n1, s1, q1), (n2, s2, q2)) in enumerate(zip(fastq(os.path.join(GOLD, "pe_reads_1.fq")), fastq(os.path.join(GOLD, "pe_reads_2.fq")))):
        if not s1 or not s2 or any(c not in "ACGTN" for c in s1 + s2):
            continue
        recs.append(bam_record(n1, 77, s1, q1, b"BCZAC\0"))
        if k % 7 == 0:
            recs.append(bam_record("single%d" % k, 4, s1[:30], q1[:30]))
        recs.append(bam_record(n2, 141, s2, q2, b"NHC\x02"))
        if k % 11 == 0:
            recs.append(bam_record("mapped%d" % k, 99, s1, q1))
    pe = os.path.join(tmp, "pe.bam")
    write_bam(pe, recs, block=5000)
    comments = ["1:N:0:ACGT", "2:Y:18:GGG extra words", "plain comment", "1:N:1:ACGT", "3:N:0:A", "nocolon", "1:X:0:A", "", "1:N:0", "a b c"]
    cfq, cfa = os.path.join(tmp, "cm.fq"), os.path.join(tmp, "cm.fa")
    with open(cfq, "w") as f, open(cfa, "w") as g:
        for i, (n, s, q) in enumerate(fastq(os.path.join(GOLD, "align_reads.fq"))):
            c = comments[i % len(comments)]
            name = n.split()[0] + (" " + c if c or i % 3 == 0 else "")
            f.write("@%s\n%s\n+\n%s\n" % (name, s, q))
            if s:
                g.write(">%s\n%s\n" % (name, s))
    pc = []
    for m in (1, 2):
        p = os.path.join(tmp, "pc_%d.fq" % m)
        with open(p, "w") as f:
            for i, (n, s, q) in enumerate(fastq(os.path.join(GOLD, "pe_reads_%d.fq" % m))):
                f.write("@%s %d:N:0:ACGT%s\n%s\n+\n%s\n" % (n, m, " x" if i % 2 else "", s, q))
        pc.append(p)
    return se, pe, cfq, cfa, pc


def cases(tmp):
    se, pe, cfq, cfa, pc = make_inputs(tmp)
    g1, g2 = os.path.join(GOLD, "pe_reads_1.fq"), os.path.join(GOLD, "pe_reads_2.fq")
    return [
        ["-b", "-U", se], ["-b", "--preserve-tags", "-U", se], ["-b", "-5", "3", "-3", "4", "--preserve-tags", "-s", "5", "-u", "40", "-U", se],
        ["-b", "--local", "--preserve-tags", "--reorder", "-U", se], ["-b", "--phred64", "-k", "3", "-U", se],
        ["-b", "-U", se + "," + se],                 # the reference reads the first file of a BAM list only
        ["-b", "-U", pe],                            # only the unpaired records of a file of pairs
        ["-b", "--align-paired-reads", "-1", pe, "-2", pe], ["-b", "--align-paired-reads", "--preserve-tags", "--local", "-1", pe, "-2", pe],
        ["-b", "--align-paired-reads", "--local", "--soft-clipped-unmapped-tlen", "-1", pe, "-2", pe],
        ["--local", "--soft-clipped-unmapped-tlen", "-1", g1, "-2", g2], ["--very-sensitive-local", "--soft-clipped-unmapped-tlen", "-k", "3", "--ff", "-1", g1, "-2", g2],
        ["--sam-append-comment", "-U", cfq], ["--sam-append-comment", "--sam-no-qname-trunc", "-k", "3", "-U", cfq],
        ["--sam-append-comment", "--no-unal", "--local", "-U", cfq], ["--sam-append-comment", "-f", "-U", cfa],
        ["--sam-append-comment", "-1", pc[0], "-2", pc[1]], ["--sam-append-comment", "--local", "--no-mixed", "-1", pc[0], "-2", pc[1]],
    ]


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
@pytest.mark.parametrize("idx", ["tiny_s", "tiny_l"])
def test_bam_and_sam_options_match_reference_hostsim(hostsim, idx, tmp_path):
    ref = ref_bin("bowtie2-align-l" if idx.endswith("_l") else "bowtie2-align-s")
    for c in cases(str(tmp_path)):
        args = c + ["-x", os.path.join(GOLD, idx)]
        assert run(hostsim, args) == run(ref, args), c


def test_option_preconditions(hostsim):
    """bt2_search.cpp:1699-1718, 1804-1807: the options that only make sense with BAM / local / FASTA-FASTQ input are refused elsewhere"""
    fq = os.path.join(GOLD, "align_reads.fq")
    for opts in (["--preserve-tags"], ["--align-paired-reads"], ["--soft-clipped-unmapped-tlen"], ["--sam-append-comment", "-r"], ["--int-quals"], ["-b", "--interleaved", fq]):
        p = subprocess.run([hostsim] + opts + ["-x", os.path.join(GOLD, "tiny_s"), "-U", fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert p.returncode != 0 and p.stdout == "", opts


def test_bad_bam_is_an_error(hostsim, tmp_path):
    """a file that is not BAM, and a BAM file cut in the middle of a record, end the run with an error (never a short SAM with exit 0)"""
    notbam = os.path.join(str(tmp_path), "x.bam")
    open(notbam, "w").write(open(os.path.join(GOLD, "align_reads.fq")).read())
    recs = [bam_record("r%d" % i, 4, "ACGTACGTACGTACGTACGTACGTACGTAC", "I" * 30) for i in range(50)]
    cut = os.path.join(str(tmp_path), "cut.bam")
    write_bam(cut, recs, block=1 << 16)
    import gzip
    raw = gzip.open(cut, "rb").read()
    with gzip.open(cut, "wb") as f:
        f.write(raw[:len(raw) - 17])
    for path in (notbam, cut):
        p = subprocess.run([hostsim, "-b", "-x", os.path.join(GOLD, "tiny_s"), "-U", path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert p.returncode != 0, path


@pytest.mark.gpu
@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
def test_bam_and_sam_options_match_reference_gpu(tmp_path):
    ref = ref_bin("bowtie2-align-s")
    for c in cases(str(tmp_path)):
        args = c + ["-x", os.path.join(GOLD, "tiny_s")]
        assert run(EXE, args + ["-p", "2"]) == run(ref, args), c
    # the quality-scale variants of tests/test_cli_options.py that were added with this file
    import test_cli_options as t
    for opts, path in t.input_variants(str(tmp_path)):
        if opts and opts[0].startswith("--solexa"):
            args = opts + ["-x", os.path.join(GOLD, "tiny_s"), "-U", path]
            assert run(EXE, args + ["-p", "2"]) == run(ref, args), opts

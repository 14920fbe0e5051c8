"""Option coverage of the drop-in command line (bt2_search.cpp:1040-1620): for each option set the SAM and the
stderr summary must equal the reference binary's.  CPU: the host-compiled worker (tests/hostsim, same parser, same
reader, same SAM writer as the product).  GPU: the product binary itself (marked gpu)."""
import gzip
import os
import subprocess

import pytest

from bt2test import have_ref, ref_bin, build_hostsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
HS = os.path.join(ROOT, "tests", "hostsim")
EXE = os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-align-s")
FQ = os.path.join(GOLD, "align_reads.fq")

OPTION_SETS = [
    ["--mp", "4,2"], ["--mp", "5"], ["--np", "3"], ["--rdg", "4,2", "--rfg", "6,4"], ["--ignore-quals"], ["--ignore-quals", "--mp", "4,1"],
    ["--gbar", "10"], ["--dpad", "5"], ["--no-1mm-upfront"], ["--no-unal"], ["--xeq"], ["-k", "4", "--omit-sec-seq"],
    ["--rg-id", "grp1", "--rg", "SM:x", "--rg", "PL:illumina"], ["-5", "7", "-3", "11"], ["-3", "200"], ["--no-hd"], ["--no-sq"],
    ["-s", "100", "-u", "200"], ["-M", "3"], ["-D", "5", "-R", "1", "-L", "18", "-i", "C,10,0"],
    ["--score-min", "L,-1,-0.3", "--n-ceil", "L,0,0.5"], ["--seed", "77"], ["--very-sensitive", "--nofw"], ["--fast", "--norc"],
    ["--local"], ["--very-fast-local"], ["--very-sensitive-local", "-k", "3"], ["--local", "--ma", "3", "--mp", "4,2"], ["--local", "--score-min", "G,1,10"],
    ["--sensitive-local", "--no-unal", "--xeq"], ["-a"], ["-a", "--local"], ["--all", "--very-fast"],
    ["--bwa-sw-like"], ["--bwa-sw-like", "-k", "3"], ["--policy", "MMP=C4;NP=C2;RDG=4,2;MIN=L,-2,-0.4;SEEDLEN=18;IVAL=C,8,0;DPS=8;ROUNDS=1"],
    ["--policy", "MMP=Q,5,1;NCEIL=L,0,0.4", "-X", "300", "-I", "10", "--fr", "--no-mixed"],
    ["-N", "1"], ["-N", "1", "--local", "-k", "3"], ["-N", "1", "-L", "10", "-i", "C,3,0"], ["--multiseed", "1,18,S,1,0.5"],
    ["-N", "1", "-L", "32", "--very-fast", "--nofw"], ["-N", "1", "--no-1mm-upfront", "-L", "12"], ["-N", "1", "-a", "-L", "25"],
    ["--policy", "SEED=1;SEEDLEN=16", "--very-sensitive"],
    ["-d", "-a", "--no-exact-upfront", "--no-1mm-upfront"], ["-d", "-a", "--no-exact-upfront", "--no-1mm-upfront", "--local", "-N", "1", "-L", "20"],
    ["--no-exact-upfront"], ["--no-exact-upfront", "--no-1mm-upfront", "-k", "2"],
    ["--passthrough"], ["--passthrough", "-k", "3", "--local"],
    ["--policy", "MMP=R"], ["--policy", "MMP=R;NP=Q", "--local", "-k", "2"],
]

# added when round 2's GPU minutes were spent: CPU only here (an empty run: "0 reads" without a section, aln_sink.cpp:364-370)
LATE_SETS = [["-s", "100000"]]


@pytest.fixture(scope="module")
def hostsim():
    exe = os.path.join(HS, "hostsim")
    build_hostsim(exe)
    return exe


def run(exe, args, tmp):
    out = os.path.join(tmp, "o.sam")
    p = subprocess.run([exe] + args + ["-S", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    all_mode = "-a" in args or "--all" in args
    # -a on the 3-bp read "short" yields hundreds of alignments: over the 64-alignment record of this build, which flags the
    # read and exits 1 by design; every other read must still be identical
    assert p.returncode == 0 or (all_mode and "exceeded a limit of this build" in p.stderr), p.stderr[-1500:]
    sam = [l for l in open(out).read().splitlines() if not l.startswith("@PG") and not (all_mode and l.startswith("short\t"))]
    summ = [] if all_mode else [l for l in p.stderr.splitlines() if not l.startswith("Warning") and "amdgpu.ids" not in l]
    return sam, summ


def input_variants(tmp):
    """the golden reads as FASTA, raw, phred64 FASTQ and gzipped FASTQ"""
    lines = open(FQ).read().splitlines()
    recs = [(lines[i][1:], lines[i + 1], lines[i + 3]) for i in range(0, len(lines), 4)]
    fa, raw, p64, gz = (os.path.join(tmp, n) for n in ("r.fa", "r.raw", "r64.fq", "r.fq.gz"))
    open(fa, "w").write("".join(">%s\n%s\n" % (n, s) for n, s, _ in recs))
    open(raw, "w").write("".join("%s\n" % s for _, s, _ in recs if s))
    open(p64, "w").write("".join("@%s\n%s\n+\n%s\n" % (n, s, "".join(chr(ord(c) + 31) for c in q)) for n, s, q in recs))
    # 64-based Solexa (log-odds) qualities over the whole range the old pipelines wrote, -5 .. 40 (--solexa-quals, qual.h:105-123)
    sol = os.path.join(tmp, "rsol.fq")
    open(sol, "w").write("".join("@%s\n%s\n+\n%s\n" % (n, s, "".join(chr(59 + (7 * k + 3 * len(n) + ord(c)) % 46) for k, c in enumerate(q))) for n, s, q in recs))
    with gzip.open(gz, "wt") as f:
        f.write(open(FQ).read())
    tab = os.path.join(tmp, "r.tab5")
    open(tab, "w").write("".join("%s\t%s\t%s\n" % (n, s, q) for n, s, q in recs if s))
    cmdline = ",".join("%s:%s" % (s, q) if i % 2 else s for i, (_, s, q) in enumerate(recs[:12]) if len(s) > 20 and ":" not in q and "," not in q)
    # a comma-separated list of inputs, plain and gzipped mixed
    half = os.path.join(tmp, "half.fq")
    open(half, "w").write("".join("@%s\n%s\n+\n%s\n" % r for r in recs[:len(recs) // 2]))
    return [(["-f"], fa), (["-r"], raw), (["--phred64"], p64), (["--solexa1.3-quals"], p64), (["--solexa-quals"], sol), (["--solexa-quals", "--phred33"], FQ), ([], gz), (["--tab5"], tab), (["--tab6"], tab), (["-c"], cmdline), ([], half + "," + gz),
            (["-f", "--passthrough"], fa), (["--tab5", "--passthrough"], tab), (["-c", "--passthrough"], cmdline)]


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
@pytest.mark.parametrize("idx", ["tiny_s", "tiny_l"])
def test_options_match_reference_hostsim(hostsim, idx, tmp_path):
    ref = ref_bin("bowtie2-align-l" if idx.endswith("_l") else "bowtie2-align-s")
    base = os.path.join(GOLD, idx)
    for opts in OPTION_SETS + LATE_SETS:
        args = opts + ["-x", base, "-U", FQ]
        assert run(hostsim, args, str(tmp_path)) == run(ref, args, str(tmp_path)), opts
    for opts, path in input_variants(str(tmp_path)):
        args = (opts[:1] + [path] + opts[1:] + ["-x", base]) if opts and opts[0].startswith("--tab") else (opts + ["-x", base, "-U", path])
        assert run(hostsim, args, str(tmp_path)) == run(ref, args, str(tmp_path)), opts


def test_unsupported_options_are_refused(hostsim):
    for opts in (["-1", "a.fq"], ["-f", "-1", "a.fa", "-2", "b.fa"], ["-N", "2"], ["-k", "1001"], ["--frobnicate"], ["-d"], ["-d", "-k", "2", "--no-exact-upfront", "--no-1mm-upfront"]):
        p = subprocess.run([hostsim] + opts + ["-x", os.path.join(GOLD, "tiny_s"), "-U", FQ], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert p.returncode != 0 and p.stdout == "", opts


@pytest.mark.gpu
@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
def test_options_match_reference_gpu(tmp_path):
    ref = ref_bin("bowtie2-align-s")
    base = os.path.join(GOLD, "tiny_s")
    for opts in OPTION_SETS:
        args = opts + ["-x", base, "-U", FQ]
        assert run(EXE, args + ["-p", "4"], str(tmp_path)) == run(ref, args, str(tmp_path)), opts
    for opts, path in input_variants(str(tmp_path)):
        if opts and opts[0].startswith("--solexa"):
            continue        # added when round 2's GPU minutes were spent: their GPU twin lives in tests/test_zz_bam_input.py, which sorts last (-x)
        args = (opts[:1] + [path] + opts[1:] + ["-x", base]) if opts and opts[0].startswith("--tab") else (opts + ["-x", base, "-U", path])
        assert run(EXE, args, str(tmp_path)) == run(ref, args, str(tmp_path)), opts

"""The product's aligner driver on the CPU: bowtie2_amd/csrc/bt2g_search.cpp (the code behind extern "C" bowtie() and the drop-in
executables) compiled unchanged into tests/hostsim/driver_twin -- against a dozen HIP runtime calls on host memory and the C ABI of
include/bt2g.h backed by the host-compiled worker.  Checked here, without a GPU, is what the driver adds around the kernels: reader /
device-stage / writer threads with several batches in flight and ordered output, small --batch values, packed result records, the switch
from pair batches to unpaired batches of a mixed run, --shard blocks + the shard index the N-GPU driver (bowtie2_amd.mgpu) merges."""
import os
import socket
import subprocess
import sys

import pytest

from bt2test import have_ref, ref_bin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
HS = os.path.join(ROOT, "tests", "hostsim")
M1, M2, FQ = (os.path.join(GOLD, n) for n in ("pe_reads_1.fq", "pe_reads_2.fq", "align_reads.fq"))


@pytest.fixture(scope="module")
def twin():
    exe = os.path.join(HS, "hostsim_driver_twin")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-w", "-I" + os.path.join(HS, "fakehip"), "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(HS, "driver_twin.cpp"), os.path.join(ROOT, "bowtie2_amd", "csrc", "bt2g_index.cpp"), "-lz", "-lpthread"])
    return exe


def run(exe, args):
    p = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-1500:]
    return [l for l in p.stdout.splitlines() if not l.startswith("@PG")], [l for l in p.stderr.splitlines() if not l.startswith("Warning")]


def golden(name):
    return open(os.path.join(GOLD, name)).read().splitlines()


@pytest.mark.parametrize("extra", [[], ["-p", "4", "--batch", "64"], ["-p", "2", "--batch", "2"], ["-p", "3", "--batch", "1000"]], ids=["default", "b64", "b2", "b1000"])
def test_golden_sam_through_the_driver(twin, extra):
    """several batches in flight on three device-stage threads, host threads for parsing / formatting: the SAM is the reference's golden SAM"""
    for w in ("s", "l"):
        base = os.path.join(GOLD, "tiny_" + w)
        assert run(twin, ["--sensitive", "-x", base, "-U", FQ] + extra)[0] == golden("align_golden_%s_sens.sam" % w)
        assert run(twin, ["--local", "-x", base, "-U", FQ] + extra)[0] == golden("align_golden_%s_local.sam" % w)
        assert run(twin, ["-k", "5", "-x", base, "-U", FQ] + extra)[0] == golden("align_golden_%s_k5.sam" % w)
        assert run(twin, ["--sensitive", "-x", base, "-1", M1, "-2", M2] + extra)[0] == golden("pe_golden_%s_sens.sam" % w)
        assert run(twin, ["--local", "-k", "2", "-x", base, "-1", M1, "-2", M2] + extra)[0] == golden("pe_golden_%s_local.sam" % w)


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
def test_mixed_inputs_through_the_driver(twin):
    """pairs, then unpaired reads, in one run of the driver: the kernel is chosen per batch, the summary is the mixed one
    (the reference runs with -p 1: 2.5.5 does not finish on mixed input with more threads)"""
    for opts in ([], ["--local", "-k", "3"], ["--no-mixed", "--no-unal"], ["-u", "100"], ["-u", "200"], ["-s", "100", "-u", "61"]):
        a = opts + ["-x", os.path.join(GOLD, "tiny_s"), "-1", M1, "-2", M2, "-U", FQ + "," + FQ]
        p = subprocess.run([ref_bin("bowtie2-align-s")] + a + ["-p", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        want = ([l for l in p.stdout.splitlines() if not l.startswith("@PG")], [l for l in p.stderr.splitlines() if not l.startswith("Warning")])
        for extra in ([], ["-p", "3", "--batch", "64"], ["--batch", "2"]):
            assert run(twin, a + extra) == want, (opts, extra)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("inp", ["unpaired", "paired"])
def test_sharded_ranks_through_the_driver(twin, inp, tmp_path):
    """bowtie2_amd.mgpu, world size 2 on gloo, every rank running the product's driver (--shard r/N, --shard-index): the SAM rank 0
    reassembles and the summary it sums equal the one-process run and the golden SAM"""
    src = ["-U", FQ] if inp == "unpaired" else ["-1", M1, "-2", M2]
    common = ["--sensitive", "--batch", "64", "-x", os.path.join(GOLD, "tiny_s")] + src
    one = run(twin, common)
    port = free_port()
    out = tmp_path / "merged.sam"
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo", PYTHONPATH=ROOT)
        procs.append(subprocess.Popen([sys.executable, "-m", "bowtie2_amd.mgpu", "--engine", twin, "--backend", "gloo", "--"] + common + ["-S", str(out)],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    errs = []
    for p in procs:
        _, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-2000:]
        errs.append(se)
    got = [l for l in open(out).read().splitlines() if not l.startswith("@PG")]
    assert got == one[0] == golden("align_golden_s_sens.sam" if inp == "unpaired" else "pe_golden_s_sens.sam")
    nsum = len(one[1])
    assert [l for l in errs[0].strip().splitlines() if not l.startswith("Warning")][-nsum:] == one[1]


def test_two_device_contexts_keep_input_order(twin):
    """--gpu 0,1: one context and three device-stage threads per listed device, batches dealt to whichever is free; the writer puts
    them back in input order (with 16-read batches every device sees dozens of them)"""
    base = os.path.join(GOLD, "tiny_s")
    for extra in (["--gpu", "0,1", "--batch", "16", "-p", "4"], ["--gpu", "0,1,2", "--batch", "2"]):
        assert run(twin, ["--sensitive", "-x", base, "-U", FQ] + extra)[0] == golden("align_golden_s_sens.sam")
        assert run(twin, ["--sensitive", "-x", base, "-1", M1, "-2", M2] + extra)[0] == golden("pe_golden_s_sens.sam")


def test_input_errors_end_the_run_with_an_error(twin, tmp_path):
    """malformed input met by the reader thread while other stages are running: exit status 1 and an error message, never a short SAM with status 0"""
    lines = open(FQ).read().splitlines()
    cut = tmp_path / "cut.fq"
    cut.write_text("\n".join(lines[:4 * 300 + 2]) + "\n")             # ends inside record 301
    shortq = tmp_path / "shortq.fq"
    bad = list(lines)
    bad[4 * 200 + 3] = bad[4 * 200 + 3][:-3]                           # record 201: fewer qualities than bases
    shortq.write_text("\n".join(bad) + "\n")
    m2short = tmp_path / "m2.fq"
    m2short.write_text("\n".join(open(M2).read().splitlines()[:4 * 100]) + "\n")
    base = os.path.join(GOLD, "tiny_s")
    for a in (["-U", str(cut)], ["-U", str(shortq)], ["-1", M1, "-2", str(m2short)], ["-U", str(tmp_path / "missing.fq")]):
        p = subprocess.run([twin, "-x", base, "--batch", "64", "-p", "3"] + a, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert p.returncode == 1 and "Error" in p.stderr, (a, p.returncode, p.stderr[-300:])


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
def test_option_surface_through_the_driver(twin, tmp_path):
    """every option set, input format and paired option set of tests/test_cli_options.py / tests/test_paired.py, this time through the
    product's driver (48-read batches, three host threads) against the reference binary: SAM and summary"""
    import test_cli_options as t
    import test_paired as tp
    ref = ref_bin("bowtie2-align-s")
    base = os.path.join(GOLD, "tiny_s")
    extra = ["--batch", "48", "-p", "3"]
    tmp = str(tmp_path)

    def both(args, all_mode=False):
        a = t.run(ref, args, tmp)
        out = os.path.join(tmp, "o.sam")
        p = subprocess.run([twin] + args + extra + ["-S", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        # -a on the 3-bp read "short" is over the record capacity: the driver flags it and exits 1 by design, every other read must agree
        assert p.returncode == 0 or (all_mode and "exceeded a limit of this build" in p.stderr), p.stderr[-1500:]
        sam = [l for l in open(out).read().splitlines() if not l.startswith("@PG") and not (all_mode and l.startswith("short\t"))]
        summ = [] if all_mode else [l for l in p.stderr.splitlines() if not l.startswith("Warning")]
        assert (sam, summ) == a, args
    for opts in t.OPTION_SETS + t.LATE_SETS:
        both(opts + ["-x", base, "-U", t.FQ], "-a" in opts or "--all" in opts)
    for opts, path in t.input_variants(tmp):
        both((opts[:1] + [path] + opts[1:] + ["-x", base]) if opts and opts[0].startswith("--tab") else (opts + ["-x", base, "-U", path]))
    for opts in tp.OPTION_SETS:
        both(list(opts) + ["-x", base, "-1", M1, "-2", M2])


def test_degenerate_fastq_files(twin, tmp_path):
    """an empty file is an empty run ("0 reads"); a file of nothing but blank lines is not FASTQ (the reference aborts on it, pat.cpp:1070);
    blank lines before the first or after the last record are skipped; a last line without its newline and CRLF line ends are read"""
    base = os.path.join(GOLD, "tiny_s")
    rec = "@r1\nACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII"
    files = {"empty": "", "blank": "\n\n\n", "lead": "\n\n" + rec + "\n", "trail": rec + "\n\n\n\n", "nonl": rec, "crlf": rec.replace("\n", "\r\n") + "\r\n"}
    for name, text in files.items():
        p = tmp_path / (name + ".fq")
        p.write_bytes(text.encode())
        r = subprocess.run([twin, "-x", base, "-U", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
        body = [l for l in r.stdout.splitlines() if not l.startswith("@")]
        if name == "blank":
            assert r.returncode == 1 and "does not look like a FASTQ file" in r.stderr
        elif name == "empty":
            assert r.returncode == 0 and body == [] and r.stderr.splitlines()[-2:] == ["0 reads", "0.00% overall alignment rate"]
        else:
            assert r.returncode == 0 and len(body) == 1 and body[0].startswith("r1\t"), name


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
def test_degenerate_fasta_files(twin, tmp_path):
    """FASTA records without a single sequence character are not reads for the reference (pat.cpp:849-851: they take their number and are
    dropped), a last line without its newline loses its last character, a file of blank lines only is an error: SAM and summary equal the reference's"""
    base = os.path.join(GOLD, "tiny_s")
    files = {"holes": ">a\n\n>b\nACGTACGTACGTAGCTAGCTAGCTAGCTAGCATCGAT\n>c\n>d\n\n\n>e\nTTTTGACGATCGATCGATGCTAGCTAGCTAG\n>f\n",
             "tail1": ">a\nACGTACGTACGTAGCTAGCTAGCTAGC\n>z\nA", "nonl": ">a\nACGTACGTACGTACGTACGTACGTACGTAAAC", "multi": ">a\nACGTACGTACGTACGTACGT\nACGTACGTACGTAAAC\n>b desc\nTTGACGATCGATCGATCGATTTAGCAGCG\n",
             "empty": ""}
    for name, text in files.items():
        p = tmp_path / (name + ".fa")
        p.write_text(text)
        a = ["-f", "-x", base, "-U", str(p)]
        r = subprocess.run([ref_bin("bowtie2-align-s")] + a + ["-p", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
        want = ([l for l in r.stdout.splitlines() if not l.startswith("@PG")], [l for l in r.stderr.splitlines() if "reads" in l or "aligned" in l or "overall" in l])
        got = run(twin, a + ["--batch", "2"])
        assert (got[0], [l for l in got[1] if "reads" in l or "aligned" in l or "overall" in l]) == want, name
    p = tmp_path / "blank.fa"
    p.write_text("\n\n")
    r = subprocess.run([twin, "-f", "-x", base, "-U", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 1 and "does not look like a FASTA file" in r.stderr

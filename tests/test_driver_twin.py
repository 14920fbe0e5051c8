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

from bt2test import HOSTSIM_CLASS_FLAGS, have_ref, ref_bin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
HS = os.path.join(ROOT, "tests", "hostsim")
M1, M2, FQ = (os.path.join(GOLD, n) for n in ("pe_reads_1.fq", "pe_reads_2.fq", "align_reads.fq"))


@pytest.fixture(scope="module")
def twin():
    exe = os.path.join(HS, "hostsim_driver_twin")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-w"] + HOSTSIM_CLASS_FLAGS + ["-I" + os.path.join(HS, "fakehip"), "-I" + os.path.join(ROOT, "include"), "-o", exe,
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


def mgpu_run(twin, common, out, world, sharding=None, env_extra=None):
    port = free_port()
    procs = []
    pre = ["--sharding", sharding] if sharding else []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo", PYTHONPATH=ROOT)
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, "-m", "bowtie2_amd.mgpu", "--engine", twin, "--backend", "gloo"] + pre + ["--"] + common + ["-S", str(out)],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    errs = []
    for p in procs:
        _, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-2000:]
        errs.append(se)
    return errs


@pytest.mark.parametrize("inp", ["unpaired", "paired"])
@pytest.mark.parametrize("sharding,world", [("blocks", 2), ("bytes", 2), ("bytes", 3), ("bytes", 8), ("blocks", 8)])      # 8: the node the scaling bench runs on
def test_sharded_ranks_through_the_driver(twin, inp, sharding, world, tmp_path):
    """bowtie2_amd.mgpu on gloo, every rank running the product's driver.  blocks: --shard r/N, SAM pieces gathered over the process group.
    bytes: --shard-bytes, every rank parses its own byte range of the reads file(s) only and the pieces concatenate.  Either way the merged
    SAM and the summed summary equal the one-process run and the reference's golden SAM -- including the @PG line, which shows the user's
    command line, not a rank's."""
    src = ["-U", FQ] if inp == "unpaired" else ["-1", M1, "-2", M2]
    common = ["--sensitive", "--batch", "64" if world < 8 else "16", "-x", os.path.join(GOLD, "tiny_s")] + src      # (8 ranks: enough blocks for every rank to own several)
    one = run(twin, common)
    out = tmp_path / "merged.sam"
    errs = mgpu_run(twin, common, out, world, sharding, {"BT2G_MGPU_REPORT_BYTES": "1"})
    text = open(out).read().splitlines()
    got = [l for l in text if not l.startswith("@PG")]
    assert got == one[0] == golden("align_golden_s_sens.sam" if inp == "unpaired" else "pe_golden_s_sens.sam")
    pg = [l for l in text if l.startswith("@PG")]
    assert len(pg) == 1 and "--shard" not in pg[0] and "piece" not in pg[0] and " ".join(common) in pg[0]
    lines0 = [l for l in errs[0].strip().splitlines() if not l.startswith("Warning") and not l.startswith("[mgpu]")]
    assert lines0 == one[1]          # one summary on rank 0: the merged one, equal to the one-process summary (no rank prints its own share)
    rep = [l for l in errs[0].splitlines() if l.startswith("[mgpu]")][0].split()
    vals = dict(kv.split("=") for kv in rep[1:])
    total = sum(os.path.getsize(p) for p in src[1::2])
    if sharding == "bytes":
        # no rank parses another rank's bytes: together they read the input once, each about 1/N of it
        assert vals["sharding"] == "bytes" and int(vals["parsed_bytes_all"]) == total
        assert abs(int(vals["parsed_bytes_this_rank"]) - total / world) < 0.02 * total + 2000
    else:
        assert vals["sharding"] == "blocks" and int(vals["parsed_bytes_all"]) == world * total       # the fallback reads everything on every rank


@pytest.mark.parametrize("extra", [["-X", "1500", "--dovetail"], ["--local", "-X", "1200"], ["-X", "5000"]])
def test_wide_mate_windows_through_the_driver(twin, extra):
    """Pairs whose opposite-mate windows are wider than the 1 100 columns a launch holds by default: the driver derives the widest window of
    each batch from its pairs' lengths, minimum scores and -I/-X (mate_window_bound) and asks bt2g_align_batch for it
    (bt2g_align_params::max_dp_cols, at most 2 176).  Same SAM as the worker under a plain main, which always holds the maximum; -X 5000 is
    beyond that: both flag the same pairs (a warning; the driver exits 1) instead of printing something else."""
    common = ["--sensitive", "--batch", "32"] + extra + ["-x", os.path.join(GOLD, "tiny_s"), "-1", M1, "-2", M2]
    p = subprocess.run([twin] + common, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    hs = os.path.join(HS, "hostsim")
    q = subprocess.run([hs] + common, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    body = lambda t: [l for l in t.splitlines() if not l.startswith("@PG")]
    assert body(p.stdout) == body(q.stdout)
    if "5000" in extra:
        assert p.returncode == 1 and "Warning" in p.stderr and "Warning" in q.stderr      # (the plain main only warns)
    else:
        assert p.returncode == 0 and "Warning" not in p.stderr


@pytest.mark.parametrize("inp", ["unpaired"])      # ("paired" works the same way and passes -- 50 s on the CPU twin, not part of the default run)
def test_varied_batch_sizes_do_not_change_the_sam(twin, inp, tmp_path):
    """--batch-max N: the reader ramps the batch size up from 64 K reads and, for one plain file, tapers it off towards the end of the input
    (bt2g_search.cpp); batches are recycled with their memory either way (HostBatch::recycle).  However the input is cut, the SAM and the summary
    are those of the fixed-size run.  The input here is the golden reads repeated until the ramp has several steps to take."""
    src_files = [FQ] if inp == "unpaired" else [M1, M2]
    big = []
    for k, f in enumerate(src_files):
        text = open(f).read()
        recs = text.splitlines()
        out = []
        for rep in range(140 if inp == "unpaired" else 215):      # 73 K unpaired reads / 2 x 34 K mates: the first 64 K-read batch, then the taper
            for i in range(0, len(recs), 4):
                nm = recs[i].split()[0]
                nm = (nm[:-2] + "_%d" % rep + nm[-2:]) if nm.endswith(("/1", "/2")) else nm + "_%d" % rep
                out += [nm, recs[i + 1], recs[i + 2], recs[i + 3]]
        p = tmp_path / ("big_%d.fq" % k)
        p.write_text("\n".join(out) + "\n")
        big.append(str(p))
    src = ["-U", big[0]] if inp == "unpaired" else ["-1", big[0], "-2", big[1]]
    common = ["--very-fast", "-x", os.path.join(GOLD, "tiny_s")] + src
    fixed = run(twin, ["--batch", "50000"] + common)
    varied = run(twin, ["--batch-max", "131072"] + common)
    assert varied[0] == fixed[0] and len(fixed[0]) > 60000
    assert varied[1] == fixed[1]


def test_byte_sharding_falls_back_for_gzip_and_odd_files(twin, tmp_path):
    """gzip'ed input, or a file that is not strict 4-line FASTQ (a blank line between records), cannot be cut by byte ranges: the ranks
    agree on block mode and the output is still the one-process output"""
    import gzip
    gz = tmp_path / "reads.fq.gz"
    with open(FQ, "rb") as f, gzip.open(gz, "wb") as g:
        g.write(f.read())
    odd = tmp_path / "blank.fq"
    lines = open(FQ).read().splitlines()
    odd.write_text("\n".join(lines[:400]) + "\n\n" + "\n".join(lines[400:]) + "\n")
    want = golden("align_golden_s_sens.sam")
    for k, path in enumerate((gz, odd)):
        common = ["--sensitive", "--batch", "64", "-x", os.path.join(GOLD, "tiny_s"), "-U", str(path)]
        out = tmp_path / ("m%d.sam" % k)
        errs = mgpu_run(twin, common, out, 2, None, {"BT2G_MGPU_REPORT_BYTES": "1"})
        assert "sharding=blocks" in errs[0]
        assert [l for l in open(out).read().splitlines() if not l.startswith("@PG")] == want


def test_byte_ranges_partition_any_file(tmp_path):
    """plan_byte_ranges on one process per 'rank' emulated in-process: for many rank counts the ranges tile the file, start on record
    boundaries, agree between the two mate files on the record number, and tiny files with more ranks than records still work"""
    sys.path.insert(0, ROOT)
    from bowtie2_amd import mgpu
    import random
    rng = random.Random(7)
    recs1, recs2 = [], []
    for i in range(257):
        L = rng.randint(1, 180)
        nm = "r%d" % rng.randint(0, 10 ** rng.randint(1, 9))
        recs1.append("@%s/1\n%s\n+\n%s\n" % (nm, "A" * L, "@" * L))          # quality lines that start with '@'
        L2 = rng.randint(1, 180)
        recs2.append("@%s/2 extra words\n%s\n+%s\n%s\n" % (nm, "C" * L2, nm, "+" * L2))
    f1, f2 = tmp_path / "a.fq", tmp_path / "b.fq"
    f1.write_text("".join(recs1)); f2.write_text("".join(recs2)[:-1])             # the second file ends without a newline
    off1 = [0]; off2 = [0]
    for a, b in zip(recs1, recs2):
        off1.append(off1[-1] + len(a)); off2.append(off2[-1] + len(b))
    off2[-1] -= 1

    class FakeDist:       # all_gather over ranks emulated by computing every rank's contribution in this process
        def __init__(self, files, world):
            self.files, self.world = files, world
        def all_gather(self, outs, t):
            import torch
            for q in range(self.world):
                sizes = [os.path.getsize(p) for p in self.files]
                vals = [mgpu.count_newlines(p, sz * q // self.world, sz * (q + 1) // self.world) for p, sz in zip(self.files, sizes)]
                tails = []
                for p, sz in zip(self.files, sizes):
                    tl = 0
                    if q == self.world - 1 and sz > 0:
                        with open(p, "rb") as f:
                            f.seek(sz - 1); tl = 0 if f.read(1) == b"\n" else 1
                    tails.append(tl)
                outs[q].copy_(torch.tensor(vals + tails, dtype=torch.int64))
    import torch
    for files, offs in (([str(f1)], [off1]), ([str(f1), str(f2)], [off1, off2])):
        for world in (1, 2, 3, 7, 64, 300):
            prev_end = [0] * len(files)
            nxt = 0
            for rank in range(world):
                plan = mgpu.plan_byte_ranges(FakeDist(files, world), files, rank, world, torch.device("cpu"))
                assert plan is not None
                ranges, first, count = plan
                assert first == nxt
                for k, (a, b) in enumerate(ranges):
                    assert a == prev_end[k] and a == offs[k][first] and b == offs[k][first + count]
                    prev_end[k] = b
                nxt = first + count
            assert nxt == 257 and prev_end == [o[-1] for o in offs]


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


def test_output_errors_end_the_run_with_an_error(twin, tmp_path):
    """a SAM stream that cannot be written (full device) is an error with exit status 1, not a truncated file and status 0; a reads file
    that cannot be opened is reported before the index is loaded"""
    base = os.path.join(GOLD, "tiny_s")
    if os.path.exists("/dev/full"):
        p = subprocess.run([twin, "-x", base, "-U", FQ, "-S", "/dev/full"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert p.returncode == 1 and "could not write SAM output" in p.stderr, (p.returncode, p.stderr[-300:])
        with open("/dev/full", "w") as full:
            p = subprocess.run([twin, "-x", base, "-U", FQ], stdout=full, stderr=subprocess.PIPE, text=True, timeout=300)
        assert p.returncode == 1 and "could not write SAM output" in p.stderr, (p.returncode, p.stderr[-300:])
    p = subprocess.run([twin, "-x", str(tmp_path / "no_such_index"), "-U", str(tmp_path / "missing.fq")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 1 and "missing.fq" in p.stderr and "index" not in p.stderr.lower().replace("no_such_index", ""), p.stderr[-300:]


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

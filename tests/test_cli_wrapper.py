"""The process-level boundary: the reference's Perl wrapper `bowtie2` asks the aligner binary
next to it for its option table (`--wrapper basic-0 --arg-desc`, bowtie2:104) and then execs it
(`bowtie2:482`).  Runs without a GPU: the option table must equal the reference binary's, and the wrapper must
get as far as exec'ing our binary, which then refuses to run without an MI355X (no CPU path)."""
import os
import shutil
import subprocess

import pytest

from bt2test import have_ref, ref_bin, build_hostsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bowtie2_amd", "bin")
GOLD = os.path.join(ROOT, "tests", "golden")
WRAPPER = "/root/reference/bowtie2"


def need_bin():
    exe = os.path.join(BIN, "bowtie2-align-s")
    if not os.path.exists(exe):
        pytest.skip("drop-in binary not built")
    return exe


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
def test_arg_desc_equals_reference():
    exe = need_bin()
    ours = subprocess.run([exe, "--wrapper", "basic-0", "--arg-desc"], stdout=subprocess.PIPE, text=True, check=True).stdout
    ref = subprocess.run([ref_bin("bowtie2-align-s"), "--wrapper", "basic-0", "--arg-desc"], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert ours == ref and len(ours.splitlines()) > 200


def gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(not os.path.exists(WRAPPER) or shutil.which("perl") is None, reason="reference wrapper or perl not available")
def test_perl_wrapper_execs_our_binary(tmp_path):
    need_bin()
    # the wrapper looks for bowtie2-align-{s,l} in its own (symlink-resolved) directory: give it a scratch copy there
    shutil.copy(WRAPPER, tmp_path / "bowtie2")
    for n in ("bowtie2-align-s", "bowtie2-align-l"):
        shutil.copy(os.path.join(BIN, n), tmp_path / n)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "bowtie2_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = tmp_path / "o.sam"
    p = subprocess.run(["perl", str(tmp_path / "bowtie2"), "--sensitive", "-x", os.path.join(GOLD, "tiny_s"), "-U", os.path.join(GOLD, "align_reads.fq"),
                        "-S", str(out)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    if gpu_present():
        assert p.returncode == 0, p.stderr[-1000:]
        got = [l for l in open(out).read().splitlines() if not l.startswith("@PG")]
        assert got == open(os.path.join(GOLD, "align_golden_s_sens.sam")).read().splitlines()
    else:
        assert p.returncode != 0
        assert "no usable MI355X" in p.stderr          # our binary was reached, and it has no CPU path


@pytest.mark.skipif(not os.path.exists(WRAPPER) or shutil.which("perl") is None or not have_ref(), reason="reference wrapper, perl or oracle/_ref not available")
def test_perl_wrapper_un_al_conc_with_host_build(tmp_path):
    """The wrapper's --un-conc / --al-conc / --un / --al work through `--passthrough` (bowtie2:567-620).  The reference
    wrapper is run twice on the same pairs -- once over the reference binaries, once over our host-compiled worker standing in
    as bowtie2-align-s/-l (test infrastructure; the product binary needs a GPU) -- and every output file must be identical."""
    hs = os.path.join(ROOT, "tests", "hostsim", "hostsim")
    build_hostsim(hs)
    outs = {}
    for tag in ("ref", "ours"):
        d = tmp_path / tag
        d.mkdir()
        shutil.copy(WRAPPER, d / "bowtie2")
        for n in ("bowtie2-align-s", "bowtie2-align-l"):
            shutil.copy(ref_bin(n) if tag == "ref" else hs, d / n)
        for mode, extra in (("pe", ["-1", os.path.join(GOLD, "pe_reads_1.fq"), "-2", os.path.join(GOLD, "pe_reads_2.fq"),
                                     "--un-conc", str(d / "unc_%.fq"), "--al-conc", str(d / "alc_%.fq")]),
                            ("se", ["-U", os.path.join(GOLD, "align_reads.fq"), "--un", str(d / "un.fq"), "--al", str(d / "al.fq"), "--no-unal"])):
            p = subprocess.run(["perl", str(d / "bowtie2"), "--sensitive", "-x", os.path.join(GOLD, "tiny_s")] + extra + ["-S", str(d / (mode + ".sam"))],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            assert p.returncode == 0, p.stderr[-1000:]
        outs[tag] = {f: [l for l in open(d / f).read().splitlines() if not l.startswith("@PG")] for f in sorted(os.listdir(d)) if f.endswith((".fq", ".sam"))}
    assert set(outs["ref"]) == set(outs["ours"]) and len(outs["ref"]) == 8, sorted(outs["ours"])
    for f in outs["ref"]:
        assert outs["ref"][f] == outs["ours"][f], f
        assert outs["ref"][f], f        # none of the files is empty in this case

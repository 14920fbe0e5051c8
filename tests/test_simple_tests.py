"""The reference's own regression table, re-hosted: tests/golden/simple_tests.json holds the unpaired and paired cases of
scripts/test/simple_tests.pl (reference sequences, reads, arguments) together with the SAM the *reference* binaries
print for them (tools/make_simple_tests_fixture.py).  Our binaries must print the same SAM, byte for byte, on both
index widths -- or refuse the option set outright.  CPU: the host-compiled worker; GPU: the product binary."""
import json
import os
import subprocess

import pytest

from bt2test import have_ref, ref_bin, build_hostsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HS = os.path.join(ROOT, "tests", "hostsim")
FIXTURE = os.path.join(ROOT, "tests", "golden", "simple_tests.json")
# No option set of the table is refused
MAX_REFUSED = 0


@pytest.fixture(scope="module")
def hostsim():
    exe = os.path.join(HS, "hostsim")
    build_hostsim(exe)
    return exe


def run_cases(exe_s, exe_l, tmp, jobs=1):
    """jobs > 1: the runs are independent processes with their own input files -- the CPU twin's go through a thread pool (the device's stay
    one at a time: concurrent processes would share the GPU)."""
    from concurrent.futures import ThreadPoolExecutor
    cases = json.load(open(FIXTURE))
    assert len(cases) > 150
    refused, compared, bad = [], 0, []
    # the 142 indexes of the table (reference builder, a CPU process each) are built side by side
    built = {}
    for rec in cases:
        for width in ("s", "l"):
            built.setdefault((tuple(rec["ref"]), width), os.path.join(tmp, "idx%d" % len(built), "idx"))

    def build(item):
        (ref, width), base = item
        os.makedirs(os.path.dirname(base))
        fa = os.path.join(os.path.dirname(base), "ref.fa")
        open(fa, "w").write("".join(">%d\n%s\n" % (i, s) for i, s in enumerate(ref)))
        subprocess.check_call([ref_bin("bowtie2-build-l" if width == "l" else "bowtie2-build-s"), "--quiet", fa, base], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(build, built.items()))
    todo = []
    for ri, rec in enumerate(cases):
        for width, exe in (("s", exe_s), ("l", exe_l)):
            key = (tuple(rec["ref"]), width)
            cmd = [exe] + rec["args"] + ["-x", built[key]]
            tag = "%d%s" % (ri, width)
            if "m1" in rec:             # paired case
                if rec["flag"] == "-c":
                    cmd += ["-c", "-1", rec["m1"].strip(), "-2", rec["m2"].strip()]
                else:
                    f1, f2 = os.path.join(tmp, "m1_%s.txt" % tag), os.path.join(tmp, "m2_%s.txt" % tag)
                    open(f1, "w").write(rec["m1"])
                    open(f2, "w").write(rec["m2"])
                    cmd += [rec["flag"], "-1", f1, "-2", f2]
            elif rec["flag"] == "-c":
                cmd += ["-c", "-U", rec["input"].strip()]
            else:
                rf = os.path.join(tmp, "reads_%s.txt" % tag)
                open(rf, "w").write(rec["input"])
                cmd += ([rec["flag"], rf] if rec["flag"] == "--tab5" else ([rec["flag"], "-U", rf] if rec["flag"] else ["-U", rf]))
            todo.append((rec, width, cmd))

    def run(item):
        return subprocess.run(item[2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)

    if jobs > 1:
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            done = list(ex.map(run, todo))
    else:
        done = [run(t) for t in todo]
    for (rec, width, cmd), p in zip(todo, done):
        got = [l for l in p.stdout.splitlines() if not l.startswith("@PG")]
        if rec.get("abort"):        # malformed input or invalid arguments: the reference aborts, we must exit non-zero without alignments
            compared += 1
            if p.returncode == 0 or any(l and not l.startswith("@") for l in got):
                bad.append((rec["name"], rec["fw"], width, "should abort"))
            continue
        if p.returncode != 0 and not got:
            refused.append((rec["name"], " ".join(rec["args"]), p.stderr.strip().splitlines()[-1:] if p.stderr.strip() else ""))
            continue
        compared += 1
        if got != rec["sam"][width]:
            bad.append((rec["name"], rec["fw"], width, " ".join(rec["args"])))
    return compared, refused, bad


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref (bowtie2-build) not present")
def test_reference_regression_table_hostsim(hostsim, tmp_path):
    compared, refused, bad = run_cases(hostsim, hostsim, str(tmp_path), jobs=min(8, os.cpu_count() or 1))
    assert not bad, (len(bad), bad[:5])
    assert compared >= 860 and len(refused) <= MAX_REFUSED, (compared, len(refused), refused[:5])


@pytest.mark.gpu
@pytest.mark.skipif(not have_ref(), reason="oracle/_ref (bowtie2-build) not present")
def test_reference_regression_table_gpu(tmp_path):
    """The whole table on the device: every case on both index widths (862 runs of the product binary)."""
    b = os.path.join(ROOT, "bowtie2_amd", "bin")
    compared, refused, bad = run_cases(os.path.join(b, "bowtie2-align-s"), os.path.join(b, "bowtie2-align-l"), str(tmp_path))
    assert not bad, (len(bad), bad[:5])
    assert compared >= 860 and len(refused) <= MAX_REFUSED, (compared, len(refused), refused[:5])

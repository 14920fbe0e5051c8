"""The effort knobs behind the presets and the long spellings of the short options (bt2_search.cpp:505-705, 1274-1310, 1461-1477):
--extends / --dp-fails / --ug-fails (hard limits of the extension loop), --seed-boost (re-seeding threshold), --tighten (-M score
tightening), --no-extend (no exact extension of seed hits), --no-ungapped (every hit goes to the gapped DP), --khits / --seedlen /
--seedmms / --seedival / --index / --unpaired / --seed-rounds / --fail-streak / --minins / --maxins / --contain / --overlap, and the options
the reference accepts without effect on its output (--reads-per-batch, --thread-ceiling, --thread-piddir, --1mm-minlen).
Differential against the reference binary on the repeat-rich workloads of the other tests, where the limits actually bind (each
option set gives a SAM different from the default's).  CPU: the host-compiled worker; GPU: the product binary."""
import os
import subprocess

import pytest

from bt2test import CACHE_DIR, have_ref, ref_bin, build_hostsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HS = os.path.join(ROOT, "tests", "hostsim")
EXE = os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-align-s")

SE_SETS = [["--extends", "20"], ["--dp-fails", "5", "--ug-fails", "8"], ["--seed-boost", "20"], ["--tighten", "1"], ["--tighten", "2", "-M", "3"], ["--tighten", "0"],
           ["--no-extend"], ["--no-ungapped"], ["--no-ungapped", "--local", "-k", "3"], ["--khits", "3", "--seedlen", "18", "--seedmms", "1", "--seedival", "C,9,0"],
           ["--seed-rounds", "3", "--fail-streak", "4", "--reads-per-batch", "7", "--thread-ceiling", "3", "--1mm-minlen", "10"], ["--extends", "35", "-k", "2", "--local"]]
PE_SETS = [["--extends", "30", "--dp-fails", "10"], ["--no-ungapped", "--tighten", "1"], ["--no-extend", "--local"], ["--seed-boost", "10", "--contain", "--overlap", "--no-contain"],
           ["--minins", "100", "--maxins", "400", "--ug-fails", "3"]]


@pytest.fixture(scope="module")
def hostsim():
    exe = os.path.join(HS, "hostsim")
    build_hostsim(exe)
    return exe


def run(exe, args, extra=()):
    p = subprocess.run([exe] + args + list(extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-1500:]
    sam = [l for l in p.stdout.splitlines() if not l.startswith("@PG")]
    summ = [l for l in p.stderr.splitlines() if not l.startswith("Warning") and "amdgpu.ids" not in l]
    return sam, summ


def workloads():
    from bt2test import build_index, write_fasta, write_fastq
    from test_paired import make_pairs
    from test_work_counters import workload
    base, fq = workload(False)
    d = os.path.join(CACHE_DIR, "knobs_pe")
    os.makedirs(d, exist_ok=True)
    fa, m1, m2, pbase = os.path.join(d, "g.fa"), os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq"), os.path.join(d, "g")
    if not os.path.exists(pbase + ".rev.2.bt2"):
        refs, r1, r2 = make_pairs(500, 9)
        write_fasta(fa, refs)
        write_fastq(m1, r1)
        write_fastq(m2, r2)
        build_index(fa, pbase, False)
    return base, fq, pbase, m1, m2


def check(exe, extra, jobs=1):
    """jobs > 1 (the CPU twin): the option sets are independent processes and go through a thread pool."""
    from concurrent.futures import ThreadPoolExecutor
    base, fq, pbase, m1, m2 = workloads()
    ref = ref_bin("bowtie2-align-s")
    rp = ["-p", "8", "--reorder"]
    default = run(ref, ["-x", base, "-U", fq], rp)

    def se(opts):
        a = opts + ["--index" if "--khits" in opts else "-x", base, "--unpaired" if "--khits" in opts else "-U", fq]
        return run(ref, a, rp), run(exe, a, extra)

    def pe(opts):
        a = opts + ["-x", pbase, "-1", m1, "-2", m2]
        return run(ref, a, rp), run(exe, a, extra)

    if jobs > 1:
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            d_se, d_pe = list(ex.map(se, SE_SETS)), list(ex.map(pe, PE_SETS))
    else:
        d_se, d_pe = [se(o) for o in SE_SETS], [pe(o) for o in PE_SETS]
    for opts, (want, got) in zip(SE_SETS, d_se):
        assert want[0] != default[0], opts                      # the knob binds on this workload
        assert got == want, opts
    for opts, (want, got) in zip(PE_SETS, d_pe):
        assert got == want, opts


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
def test_effort_knobs_match_reference_hostsim(hostsim):
    check(hostsim, [], jobs=min(8, os.cpu_count() or 1))


@pytest.mark.gpu
@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
def test_effort_knobs_match_reference_gpu():
    check(EXE, ["-p", "4"])

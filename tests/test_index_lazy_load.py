"""bt2g_index_load streams the three large sections of an index (BWT sides, SA sample, 2-bit reference) from their files to the device
(FileStreamer, bt2g_capi.hip) instead of reading them into host memory first: load_index(..., lazy = true) only says where they are.  The
spans must describe exactly the bytes the eager load reads, on either index width (the GPU suite then checks what arrives: every test there
loads its index this way)."""
import os
import subprocess

import pytest

from bt2test import ROOT, ref_bin

HS = os.path.join(ROOT, "tests", "hostsim")
EX = os.path.join(ROOT, "tests", "golden", "example")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("lazy") / "lazy_index_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", exe, os.path.join(HS, "lazy_index_check.cpp"),
                           os.path.join(ROOT, "bowtie2_amd", "csrc", "bt2g_index.cpp")])
    return exe


@pytest.mark.parametrize("large", [False, True])
def test_lazy_spans_equal_eager_sections(checker, tmp_path, large):
    base = str(tmp_path / "lambda")
    subprocess.check_call([ref_bin("bowtie2-build-l" if large else "bowtie2-build-s"), "-q", os.path.join(EX, "lambda_virus.fa"), base], stdout=subprocess.DEVNULL)
    p = subprocess.run([checker, base], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0 and p.stdout.startswith("ok 4 "), p.stdout
    assert int(p.stdout.split()[2]) > 40000


def test_truncated_section_is_refused_by_lazy_load(checker, tmp_path):
    base = str(tmp_path / "lambda")
    subprocess.check_call([ref_bin("bowtie2-build-s"), "-q", os.path.join(EX, "lambda_virus.fa"), base], stdout=subprocess.DEVNULL)
    f = base + ".2.bt2"
    os.truncate(f, os.path.getsize(f) - 100)
    p = subprocess.run([checker, base], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 1 and p.stdout.startswith("lazy:") and "truncated SA sample" in p.stdout, p.stdout

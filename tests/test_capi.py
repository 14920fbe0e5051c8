"""The C-ABI shared library: builds, loads, exports every symbol include/bt2g.h declares, and
refuses to compute without a gfx950 device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "bt2g.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bt2g_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import bowtie2_amd
    if not os.path.exists(bowtie2_amd.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = C.CDLL(bowtie2_amd.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(L, s), "libbt2g.so does not export %s" % s
    # and the Python mirror binds exactly the declared set
    assert sorted(n for n, _, _ in bowtie2_amd.ABI) == syms


def test_bowtie_entry_point_returns_status():
    """extern "C" int bowtie(argc, argv) (the reference's bt2_search.cpp:5223): exported, and an argument error comes back as
    a return value instead of ending the calling process."""
    import bowtie2_amd
    L = C.CDLL(bowtie2_amd.LIB_PATH)
    assert hasattr(L, "bowtie")
    argv = (C.c_char_p * 4)(b"bowtie2-align-s", b"-x", b"/nonexistent/index", b"--no-such-option")
    assert L.bowtie(4, argv) == 1
    argv = (C.c_char_p * 2)(b"bowtie2-align-s", b"--version")
    assert L.bowtie(2, argv) == 0
    assert hasattr(L, "bowtie_build")
    argv = (C.c_char_p * 2)(b"bowtie2-build-s", b"--no-such-option")
    assert L.bowtie_build(2, argv) != 0


def test_struct_layouts_match_header():
    import bowtie2_amd as b
    assert C.sizeof(b.SweepOut) == 48
    assert C.sizeof(b.SeedHit) == 32
    assert C.sizeof(b.Resolved) == 40
    assert C.sizeof(b.DpProblem) == 40 and C.sizeof(b.DpOut) == 32
    assert C.sizeof(b.Scoring) == 40
    assert C.sizeof(b.Counters) == 40


def test_no_cpu_fallback():
    import torch
    import bowtie2_amd as b
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert b.lib().bt2g_ctx_create(0, C.byref(h)) == -1   # BT2G_ERR_NO_DEVICE
    with pytest.raises(b.Bt2gError):
        b.Context(0)


def test_product_does_not_reference_oracle():
    """The shipped sources must not include, link or import anything under oracle/."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "bowtie2_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", "Makefile")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"oracle/|bt2_oracle|liboracle|bt2ref", txt):
                    bad.append(os.path.join(d, f))
    assert not bad, bad

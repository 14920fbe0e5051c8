"""The HBM layout of the FM index (bowtie2_amd/csrc/bt2g_device.hpp: 64-byte rank blocks, the full suffix array) against the oracle's
plain restatement of the on-disk layout (oracle/bt2_oracle.c: countBt2Side over the sides, getOffset by LF walk to the SA sample),
EVERY row of the tiny golden indexes, both widths and both index directions.  The layout is built by bt2g_rankidx.hpp -- on the device
by the kernels of bt2g_rankidx.hip, here by the same per-block / per-segment functions in host loops (tests/hostsim, test-only)."""
import ctypes as C
import os
import subprocess

import pytest

from bt2test import ROOT, Index, build_hostsim, cached_synth_index, oracle, u64


def check_rows(L, idx, rows, large):
    """rows: iterable of the 16 columns bt2g_index_rows / BT2G_INDEX_DUMP give per row (the character column signed)."""
    n = idx.fwd.len
    a = (u64 * 4)()
    ns = u64()
    side_len = 384 if large else 192
    for v in rows:
        row = v[0]
        assert L.bt2o_get_offset(C.byref(idx.fwd), row, C.byref(ns)) == v[1], row
        assert ns.value == v[2], ("steps of the reference's walk", row)
        L.bt2o_rank4(C.byref(idx.fwd), row, a)
        assert list(a) == v[3:7], ("rank4 fw", row)
        r = u64(row)
        ch = L.bt2o_map_lf1(C.byref(idx.fwd), C.byref(r))
        assert ch == v[7] and (ch < 0 or r.value == v[8]), ("mapLF1", row)
        L.bt2o_rank4(C.byref(idx.bwd), row, a)
        assert list(a) == v[9:13], ("rank4 bw", row)
        bot = min(row + 37, n)
        c = row & 3
        assert [L.bt2o_rank(C.byref(idx.fwd), row, c), L.bt2o_rank(C.byref(idx.fwd), bot, c)] == v[13:15], ("rank pair", row)
        assert v[15] == (1 if row // side_len == bot // side_len else 2), ("sides the reference reads for the pair", row)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["tiny_s", "tiny_l", "synth_s", "synth_l"])
def test_every_row_of_the_device_layout(golden_dir, which):
    """The same every-row comparison on what the DEVICE built: bt2g_index_load transcodes the .bt2 files with k_make_rank_blocks /
    k_sa_segments (bt2g_rankidx.hip) and bt2g_index_rows reads every row back through the resulting layout with the device's own rank /
    LF / offset functions.  Tiny golden indexes and the 180 kbp synthetic one (default ftab and offrate), both widths."""
    import bowtie2_amd as b
    large = which.endswith("_l")
    base = os.path.join(golden_dir, which) if which.startswith("tiny") else cached_synth_index(large=large)[0]
    L = oracle()
    idx = Index()
    assert L.bt2o_index_load(C.byref(idx), base.encode()) == 0
    ctx = b.Context(0)
    try:
        info = ctx.load_index(base)
        n = idx.fwd.len
        assert info.len == n
        t = ctx.index_rows(0, n + 1).cpu().numpy()
        assert t.shape == (n + 1, 16)
        assert [int(x) for x in t[:, 0]] == list(range(n + 1))
        check_rows(L, idx, ([int(x) & (2**64 - 1) if k != 7 else int(x) for k, x in enumerate(row)] for row in t), large)
    finally:
        ctx.close()


@pytest.mark.parametrize("large", [False, True])
def test_every_row_matches_the_oracle(golden_dir, tmp_path, large):
    exe = build_hostsim(os.path.join(ROOT, "tests", "hostsim", "hostsim"))
    base = os.path.join(golden_dir, "tiny_l" if large else "tiny_s")
    env = dict(os.environ, BT2G_INDEX_DUMP="1")
    out = subprocess.run([exe, "-x", base, "-U", "/dev/null"], stdout=subprocess.PIPE, env=env, check=True, text=True).stdout.splitlines()
    L = oracle()
    idx = Index()
    assert L.bt2o_index_load(C.byref(idx), base.encode()) == 0
    n = idx.fwd.len
    assert len(out) == n + 1
    check_rows(L, idx, ([int(x) for x in line.split()] for line in out), large)

"""Differential check of the plain-C oracle against the reference's own classes (oracle/ref_shim.cpp)
on a larger synthetic genome with default ftabChars=10.  Skipped where oracle/_ref is not built."""
import ctypes as C
import random

import pytest

from bt2test import (Index, Mm1Hit, Scoring, SeedHit, SweepOut, cached_synth_index, encode, have_ref, oracle, refshim,
                     revcomp, synth_reads, u64)

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (no /root/reference)")


@pytest.fixture(scope="module", params=[False, True], ids=["bt2", "bt2l"])
def ctx(request):
    large = request.param
    base, refs = cached_synth_index(large=large)
    L = oracle()
    R = refshim(large)
    idx = Index()
    assert L.bt2o_index_load(C.byref(idx), base.encode()) == 0
    h = R.ref_open(base.encode())
    assert h
    yield L, R, idx, h, refs
    R.ref_close(h)


def test_rank_lf_offsets(ctx):
    L, R, idx, h, refs = ctx
    n = idx.fwd.len
    rnd = random.Random(1)
    rows = [0, 1, n - 1, n, idx.fwd.zoff, idx.fwd.zoff + 1, idx.bwd.zoff, idx.bwd.zoff + 1] + [rnd.randrange(0, n + 1) for _ in range(4000)]
    a = (u64 * 4)()
    b = (u64 * 4)()
    ns = u64()
    for d, e in ((0, idx.fwd), (1, idx.bwd)):
        for row in rows:
            L.bt2o_rank4(C.byref(e), row, a)
            R.ref_rank4(h, d, row, b)
            assert list(a) == list(b)
            for c in range(4):
                x = L.bt2o_map_lf1c(C.byref(e), row, c)
                y = R.ref_map_lf1c(h, d, row, c)
                assert x == (y if y != 2**64 - 1 else e.off_mask)
    for row in rows:
        assert L.bt2o_get_offset(C.byref(idx.fwd), row, C.byref(ns)) == R.ref_get_offset(h, row)


def test_sweep_and_seed_rounds(ctx):
    L, R, idx, h, refs = ctx
    reads = (synth_reads(refs, 150, 100, seed=3) + synth_reads(refs, 40, 150, seed=4, sub=0.0, ins=0, dele=0)
             + synth_reads(refs, 40, 60, seed=5, n_rate=0.01) + synth_reads(refs, 10, 12, seed=6))
    o = (u64 * 10)()
    so = SweepOut()
    out = (u64 * (5 * 2 * 64))()
    bw = u64()
    sh = SeedHit()
    for nm, s, q in reads:
        R.ref_exact_sweep(h, s.encode(), q.encode(), o)
        L.bt2o_exact_sweep(C.byref(idx.fwd), encode(s), encode(revcomp(s)), len(s), 0, 0, 2, C.byref(so))
        assert list(o) == [so.mine[0], so.mine[1], so.hit[0], so.hit[1], so.top[0], so.bot[0], so.top[1], so.bot[1], so.nelt, so.bwops]
        for (sl, iv, off) in ((22, 12, 0), (20, 7, 3)):
            if off > 0 and sl + off > len(s):
                continue
            ns_ = R.ref_seed_round(h, s.encode(), q.encode(), sl, iv, off, out, 5 * 2 * 64, C.byref(bw))
            eff = min(sl, len(s))
            mybw = 0
            for fwi in range(2):
                for i in range(ns_):
                    depth = i * iv + off
                    sub = s[depth:depth + eff]
                    if fwi:
                        sub = revcomp(sub)
                    want = list(out[(fwi * ns_ + i) * 5:(fwi * ns_ + i) * 5 + 5])
                    if "N" in sub:
                        mine = [0] * 5
                    else:
                        L.bt2o_seed_search_exact(C.byref(idx.fwd), C.byref(idx.bwd), encode(sub), len(sub), C.byref(sh))
                        mybw += sh.bwops
                        mine = [1, sh.topf, sh.botf, sh.topb, sh.botb] if sh.botf > sh.topf else [0] * 5
                    assert mine == want
            assert mybw == bw.value


def one_mm_reads(refs, seed):
    """Reads that oneMmSearch has something to say about: copies of the genome with exactly one substitution (anywhere, incl. the ends and
    the middle where the near half ends), with one N, exact copies, two substitutions (nothing to find), very short reads, homopolymers."""
    rnd = random.Random(seed)
    g = [s for _, s in refs]
    out = []
    for k in range(260):
        s = g[rnd.randrange(len(g))]
        L = rnd.choice([2, 3, 5, 9, 10, 11, 12, 20, 21, 33, 50, 100, 101, 150, 250])
        p = rnd.randrange(0, len(s) - L)
        r = list(s[p:p + L])
        if "N" in r:
            continue
        kind = k % 6
        if kind in (0, 1):
            i = rnd.choice([0, L - 1, L // 2, L // 2 - 1, (L + 1) // 2, rnd.randrange(L)]) % L
            r[i] = rnd.choice([c for c in "ACGT" if c != r[i]])
        elif kind == 2:
            r[rnd.randrange(L)] = "N"
        elif kind == 3:
            for _ in range(2):
                i = rnd.randrange(L); r[i] = rnd.choice([c for c in "ACGT" if c != r[i]])
        elif kind == 4 and L > 4:
            r = list(rnd.choice("ACGT") * L)       # low complexity: wide ranges, many branches
        r = "".join(r)
        if rnd.random() < 0.5:
            r = revcomp(r)
        out.append(("m%d" % k, r, "".join(rnd.choice("I5+#?") for _ in r)))
    return out


def oracle_one_mm(L, idx, sc, s, q, nceil, minsc, nofw, norc, local, repex, rep1mm, cap=512):
    hits = (Mm1Hit * cap)()
    n = L.bt2o_one_mm_search(C.byref(idx.fwd), C.byref(idx.bwd), encode(s), q.encode(), len(s), C.byref(sc), nceil, minsc, nofw, norc, local, repex, rep1mm, hits, cap)
    assert n <= cap
    return [hits[i] for i in range(n)]


def test_one_mm_search(ctx):
    """SeedAligner::oneMmSearch (aligner_seed.cpp:975-1325) against bt2o_one_mm_search: same hits, same order, end to end and with the local
    validity rule, with and without exact hits reported, one strand switched off."""
    L, R, idx, h, refs = ctx
    sc = Scoring()
    L.bt2o_scoring_default(C.byref(sc))
    out = (u64 * (7 * 512))()
    ex = (u64 * 6)()
    nhits = 0
    for local in (0, 1):
        sc.match_bonus = 2 if local else 0
        R.ref_set_match_bonus(h, sc.match_bonus)
        for k, (nm, s, q) in enumerate(one_mm_reads(refs, 21 + local)):
            n = len(s)
            nceil = min(int(0 + 0.15 * n), n)
            minsc = (20 + int(8.0 * __import__("math").log(n))) if local else int(-0.6 - 0.6 * n)
            nofw, norc = (k % 11 == 3), (k % 13 == 5)
            for repex, rep1mm in ((0, 1), (1, 1), (1, 0)):
                nref = R.ref_one_mm(h, s.encode(), q.encode(), minsc, nofw, norc, local, repex, rep1mm, out, 512, ex)
                mine = oracle_one_mm(L, idx, sc, s, q, nceil, minsc, nofw, norc, local, repex, rep1mm)
                m1 = [x for x in mine if x.kind == 1]
                assert len(m1) == nref, (nm, s, local, repex, rep1mm)
                for i, x in enumerate(m1):
                    assert [x.top, x.bot, x.score & (2**64 - 1), x.off5p, x.chr, x.qchr, x.fw] == list(out[i * 7:i * 7 + 7]), (nm, s, i)
                e0 = [x for x in mine if x.kind == 0]
                want = [[1, ex[1], ex[2]]] * int(ex[0]) + [[0, ex[4], ex[5]]] * int(ex[3])
                assert [[x.fw, x.top, x.bot] for x in e0] == want, (nm, s, repex)
                nhits += nref
    R.ref_set_match_bonus(h, 0)
    assert nhits > 300


def test_dp_fill_random(ctx):
    L, R, idx, h, refs = ctx
    sc = Scoring()
    L.bt2o_scoring_default(C.byref(sc))
    rnd = random.Random(8)
    g = refs[0][1].replace("N", "A")
    for t in range(25):
        rows = rnd.choice([20, 50, 100, 150, 33])
        cols = rows + rnd.choice([0, 12, 60])
        pos = rnd.randrange(0, len(g) - cols - 2)
        window = g[pos:pos + cols + 1]
        rd = list(window[(cols - rows) // 2:(cols - rows) // 2 + rows])
        for i in range(rows):
            if rnd.random() < 0.04:
                rd[i] = rnd.choice("ACGTN")
        rd = "".join(rd)
        qu = "".join(rnd.choice("GGG?5-I#") for _ in range(rows))
        rf = bytes(1 << "ACGTN".index(c) for c in window)
        bufs = [C.create_string_buffer(rows * cols) for _ in range(6)]
        flag = C.c_int()
        best_ref = R.ref_sw_fill_ee_u8(h, rd.encode(), qu.encode(), rf, cols, -250, bufs[0], bufs[1], bufs[2], C.byref(flag))
        best = L.bt2o_sw_fill_ee_u8(C.byref(sc), encode(rd), bytes(ord(c) - 33 for c in qu), rows, rf, cols, bufs[3], bufs[4], bufs[5])
        assert [b.raw for b in bufs[:3]] == [b.raw for b in bufs[3:]]
        if flag.value == 0:
            assert best == best_ref


@pytest.mark.parametrize("kind", [1, 2, 3], ids=["ee_i16", "local_u8", "local_i16"])
def test_dp_fill_other_kinds_random(ctx, kind):
    """The 16-bit end-to-end fill and the two local fills of the restatement against the reference's own SSE kernels
    (aligner_swsse_ee_i16.cpp:780, aligner_swsse_loc_u8.cpp:927, aligner_swsse_loc_i16.cpp:938): every H, E and F word of every column
    the kernel filled, the returned score, the flag (below the minimum / saturated) -- on windows drawn from the genome with reads of
    20-400 rows, sparse and dense mismatches, Ns, low and high minimum scores (early column stop), and 8-bit saturation."""
    L, R, idx, h, refs = ctx
    i32p = C.POINTER(C.c_int32)
    L.bt2o_sw_fill_kind.argtypes = [C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int64, i32p, i32p, i32p,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.bt2o_sw_fill_kind.restype = C.c_int64
    sc = Scoring()
    L.bt2o_scoring_default(C.byref(sc))
    local = kind != 1
    if local:
        sc.match_bonus = 2
        R.ref_set_match_bonus(h, 2)
    try:
        rnd = random.Random(80 + kind)
        g = refs[0][1].replace("N", "A")
        seen_flags = set()
        stopped_early = 0
        for t in range(60):
            rows = rnd.choice([20, 33, 50, 100, 150, 250, 400] if kind != 2 else [20, 33, 50, 100, 120, 150, 250])
            cols = rows + rnd.choice([0, 12, 60])
            pos = rnd.randrange(0, len(g) - cols - 2)
            window = g[pos:pos + cols + 1]
            rd = list(window[(cols - rows) // 2:(cols - rows) // 2 + rows])
            mm = rnd.choice([0.0, 0.04, 0.04, 0.3, 0.75])
            for i in range(rows):
                if rnd.random() < mm:
                    rd[i] = rnd.choice("ACGTN")
            if rnd.random() < 0.3:      # an indel, so that E and F carry the best path somewhere
                p = rnd.randrange(5, rows - 5)
                rd = rd[:p] + rd[p + rnd.randrange(1, 4):] + list("ACG")
                rd = rd[:rows]
            rd = "".join(rd)
            rows = len(rd)
            qu = "".join(rnd.choice("GGG?5-I#") for _ in range(rows))
            rf = bytes(1 << "ACGTN".index(c) for c in window)
            if local:
                minsc = rnd.choice([0, 20 + 8 * 5, int(20 + 8.0 * __import__("math").log(rows)), 2 * rows - 10, 2 * rows + 50])
            else:
                minsc = rnd.choice([-30000, int(-0.6 - 0.6 * rows), -20, -600])
            n = rows * cols
            bufs = [(C.c_int32 * n)() for _ in range(6)]
            flag, flag2, colstop = C.c_int(), C.c_int(), C.c_int()
            want = R.ref_sw_fill_kind(h, kind, rd.encode(), qu.encode(), rf, cols, minsc, bufs[0], bufs[1], bufs[2], C.byref(flag))
            got = L.bt2o_sw_fill_kind(kind, C.byref(sc), encode(rd), bytes(ord(c) - 33 for c in qu), rows, rf, cols, minsc, bufs[3], bufs[4], bufs[5],
                                      C.byref(flag2), C.byref(colstop))
            assert (got, flag2.value) == (want, flag.value), (t, rows, cols, minsc)
            ncol = colstop.value
            stopped_early += ncol < cols
            seen_flags.add(flag.value)
            for a, b in zip(bufs[:3], bufs[3:]):
                for i in range(rows):
                    assert a[i * cols:i * cols + ncol] == b[i * cols:i * cols + ncol], (t, rows, cols, i)
        assert 0 in seen_flags and -1 in seen_flags
        if kind == 2:
            assert -2 in seen_flags          # 8-bit saturation seen
        if local:
            assert stopped_early > 0
    finally:
        if local:
            R.ref_set_match_bonus(h, 0)

"""The plain-C oracle (oracle/bt2_oracle.c) against golden vectors recorded from the reference
itself (tests/golden/make_golden.py).  Needs neither a GPU nor /root/reference."""
import ctypes as C
import json
import os
import struct

import pytest

from bt2test import (Index, Rng, Scoring, SeedHit, SweepOut, encode, oracle, revcomp, sha, u64)


def load(golden_dir, large):
    L = oracle()
    idx = Index()
    base = os.path.join(golden_dir, "tiny_l" if large else "tiny_s")
    assert L.bt2o_index_load(C.byref(idx), base.encode()) == 0
    with open(os.path.join(golden_dir, "fm_golden_%s.json" % ("l" if large else "s"))) as f:
        g = json.load(f)
    return L, idx, g


@pytest.mark.parametrize("large", [False, True])
def test_header_and_rank(golden_dir, large):
    L, idx, g = load(golden_dir, large)
    assert idx.fwd.off_size == (8 if large else 4)
    assert idx.fwd.len == g["len"] and idx.bwd.len == g["len"]
    assert [idx.fwd.zoff, idx.bwd.zoff] == g["zoff"]
    a = (u64 * 4)()
    for d, e in ((0, idx.fwd), (1, idx.bwd)):
        for k, row in enumerate(g["rows"]):
            L.bt2o_rank4(C.byref(e), row, a)
            assert list(a) == g["rank4"][d][k], (d, row)
            for c in range(4):
                v = L.bt2o_map_lf1c(C.byref(e), row, c)
                assert (-1 if v == e.off_mask else v) == g["lf1c"][d][k][c], (d, row, c)
                assert L.bt2o_rank(C.byref(e), row, c) == g["rank4"][d][k][c]


@pytest.mark.parametrize("large", [False, True])
def test_offsets(golden_dir, large):
    L, idx, g = load(golden_dir, large)
    ns = u64()
    for row, want in zip(g["rows"], g["get_offset"]):
        assert L.bt2o_get_offset(C.byref(idx.fwd), row, C.byref(ns)) == want
    for q, off, rej, ti, toff, tlen, strad in g["joined"]:
        t = [u64(), u64(), u64()]
        s = C.c_int()
        L.bt2o_joined_to_text_off(C.byref(idx.fwd), q, off, C.byref(t[0]), C.byref(t[1]), C.byref(t[2]), rej, C.byref(s))
        got_ti = -1 if t[0].value == idx.fwd.off_mask else t[0].value
        assert got_ti == ti and s.value == strad
        if ti >= 0:
            assert (t[1].value, t[2].value) == (toff, tlen)


@pytest.mark.parametrize("large", [False, True])
def test_sweep_and_seeds(golden_dir, large):
    L, idx, g = load(golden_dir, large)
    so = SweepOut()
    sh = SeedHit()
    for (s, q), want, per in zip(g["reads"], g["sweeps"], g["seeds"]):
        L.bt2o_exact_sweep(C.byref(idx.fwd), encode(s), encode(revcomp(s)), len(s), 0, 0, 2, C.byref(so))
        got = [so.mine[0], so.mine[1], so.hit[0], so.hit[1], so.top[0], so.bot[0], so.top[1], so.bot[1], so.nelt, so.bwops]
        assert got == want, s
        for sl, iv, off, ns, flat, bw in per:
            mybw = 0
            eff = min(sl, len(s))
            for fwi in range(2):
                for i in range(ns):
                    depth = i * iv + off
                    sub = s[depth:depth + eff]
                    if fwi:
                        sub = revcomp(sub)
                    want5 = flat[(fwi * ns + i) * 5:(fwi * ns + i) * 5 + 5]
                    if "N" in sub:
                        mine = [0, 0, 0, 0, 0]
                    else:
                        L.bt2o_seed_search_exact(C.byref(idx.fwd), C.byref(idx.bwd), encode(sub), len(sub), C.byref(sh))
                        mybw += sh.bwops
                        mine = [1, sh.topf, sh.botf, sh.topb, sh.botb] if sh.botf > sh.topf else [0, 0, 0, 0, 0]
                    assert mine == want5, (s, sl, iv, off, fwi, i)
            assert mybw == bw


def test_dp_fill(golden_dir):
    L = oracle()
    sc = Scoring()
    L.bt2o_scoring_default(C.byref(sc))
    with open(os.path.join(golden_dir, "dp_golden.json")) as f:
        probs = json.load(f)
    for p in probs:
        rows, cols = p["rows"], p["cols"]
        rf = bytes(1 << "ACGTN".index(c) for c in p["rf"])
        qb = bytes(ord(c) - 33 for c in p["qu"])
        H = C.create_string_buffer(rows * cols)
        E = C.create_string_buffer(rows * cols)
        F = C.create_string_buffer(rows * cols)
        best = L.bt2o_sw_fill_ee_u8(C.byref(sc), encode(p["rd"]), qb, rows, rf, cols, H, E, F)
        assert best == p["best"]
        assert sha(H.raw + E.raw + F.raw) == p["sha"]


def test_rng(golden_dir):
    L = oracle()
    with open(os.path.join(golden_dir, "rng_golden.json")) as f:
        streams = json.load(f)
    for st in streams:
        r = Rng()
        L.bt2o_rng_init(C.byref(r), st["seed"])
        for op, want in zip(st["ops"], st["out"]):
            if op == 0:
                v = L.bt2o_rng_next_u32(C.byref(r))
            elif op == 1:
                v = L.bt2o_rng_next_bool(C.byref(r))
            elif op == 2:
                v = L.bt2o_rng_next_u2(C.byref(r))
            elif op == 3:
                v = struct.unpack("I", struct.pack("f", L.bt2o_rng_next_float(C.byref(r))))[0]
            else:
                x = L.bt2o_rng_next_u64(C.byref(r))
                v = (x ^ (x >> 32)) & 0xffffffff
            assert v == want


def test_ref_fetch_matches_fasta(golden_dir):
    L, idx, g = load(golden_dir, False)
    seqs = []
    cur = None
    for line in open(os.path.join(golden_dir, "tiny.fa")):
        if line.startswith(">"):
            cur = []
            seqs.append(cur)
        else:
            cur.append(line.strip())
    seqs = ["".join(s) for s in seqs]
    for ti, s in enumerate(seqs):
        buf = C.create_string_buffer(len(s) + 20)
        L.bt2o_ref_get_stretch(C.byref(idx.ref), buf, ti, -10, len(s) + 20)
        got = "".join("ACGTN"[b] for b in buf.raw)
        assert got == "N" * 10 + s + "N" * 10


def test_dp_fill_other_kinds(golden_dir):
    """16-bit end-to-end, local 8-bit and local 16-bit fills of the restatement against vectors recorded from the reference's SSE kernels
    (tests/golden/make_dp_kinds_golden.py): score, flag (0 / -1 below the minimum / -2 saturated), columns filled, sha256 of H|E|F."""
    import hashlib
    import struct
    L = oracle()
    i32p = C.POINTER(C.c_int32)
    L.bt2o_sw_fill_kind.argtypes = [C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int64, i32p, i32p, i32p,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.bt2o_sw_fill_kind.restype = C.c_int64
    with open(os.path.join(golden_dir, "dp_kinds_golden.json")) as f:
        probs = json.load(f)
    assert {p["kind"] for p in probs} == {1, 2, 3}
    for p in probs:
        sc = Scoring()
        L.bt2o_scoring_default(C.byref(sc))
        sc.match_bonus = p["match_bonus"]
        rows, cols = p["rows"], p["cols"]
        rf = bytes(1 << "ACGTN".index(c) for c in p["rf"])
        bufs = [(C.c_int32 * (rows * cols))() for _ in range(3)]
        flag, colstop = C.c_int(), C.c_int()
        got = L.bt2o_sw_fill_kind(p["kind"], C.byref(sc), encode(p["rd"]), bytes(ord(c) - 33 for c in p["qu"]), rows, rf, cols, p["minsc"],
                                  bufs[0], bufs[1], bufs[2], C.byref(flag), C.byref(colstop))
        assert (None if got == -2**63 else got, flag.value, colstop.value) == (p["score"], p["flag"], p["ncol"]), (p["kind"], rows, cols)
        h = hashlib.sha256()
        for b in bufs:
            for i in range(rows):
                h.update(struct.pack("<%di" % p["ncol"], *b[i * cols:i * cols + p["ncol"]]))
        assert h.hexdigest() == p["sha"], (p["kind"], rows, cols)


@pytest.mark.parametrize("large", [False, True])
def test_one_mm_search(golden_dir, large):
    """bt2o_one_mm_search against what SeedAligner::oneMmSearch itself produced (tests/golden/make_one_mm_golden.py): the 1-mismatch hits in
    the order the reference adds them and the exact hits, end to end and local, all repex / rep1mm combinations, strands switched off."""
    from bt2test import Mm1Hit
    L, idx, _ = load(golden_dir, large)
    with open(os.path.join(golden_dir, "one_mm_golden.json")) as f:
        cases = json.load(f)["l" if large else "s"]
    sc = Scoring()
    L.bt2o_scoring_default(C.byref(sc))
    hits = (Mm1Hit * 512)()
    total = 0
    for c in cases:
        sc.match_bonus = 2 if c["local"] else 0
        s = c["seq"]
        n = L.bt2o_one_mm_search(C.byref(idx.fwd), C.byref(idx.bwd), encode(s), c["qual"].encode(), len(s), C.byref(sc), c["nceil"], c["minsc"],
                                 c["nofw"], c["norc"], c["local"], c["repex"], c["rep1mm"], hits, 512)
        mine = [hits[i] for i in range(n)]
        m1 = [[x.top, x.bot, x.score, x.off5p, x.chr, x.qchr, x.fw] for x in mine if x.kind == 1]
        assert m1 == c["hits"], (s, c["local"], c["repex"], c["rep1mm"])
        ex = c["exact"]
        want = [[1, ex[1], ex[2]]] * ex[0] + [[0, ex[4], ex[5]]] * ex[3]
        assert [[x.fw, x.top, x.bot] for x in mine if x.kind == 0] == want, (s, c["repex"])
        total += len(m1)
    assert total > 300

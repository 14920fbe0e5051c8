"""Parity of the HIP stage kernels (through the C ABI) against the plain-C oracle on seeded inputs,
against the committed golden vectors, and -- at larger sizes -- through round-trip properties."""
import ctypes as C
import json
import os
import random

import pytest

pytestmark = pytest.mark.gpu

import bowtie2_amd as b
from bt2test import (Index, Mm1Hit, Scoring, SeedHit, SweepOut, cached_synth_index, encode, oracle, revcomp, sha,
                     synth_reads, u64)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def tiny_refs():
    refs, cur = [], None
    for line in open(os.path.join(GOLD, "tiny.fa")):
        if line.startswith(">"):
            cur = [line[1:].strip(), ""]
            refs.append(cur)
        else:
            cur[1] += line.strip()
    return [tuple(r) for r in refs]


def make_case(name):
    if name == "tiny_s":
        return os.path.join(GOLD, "tiny_s"), tiny_refs()
    if name == "tiny_l":
        return os.path.join(GOLD, "tiny_l"), tiny_refs()
    if name == "synth_s":
        return cached_synth_index(large=False)
    return cached_synth_index(large=True)


@pytest.fixture(scope="module", params=["tiny_s", "tiny_l", "synth_s", "synth_l"])
def case(request):
    import torch
    assert torch.cuda.is_available()
    base, refs = make_case(request.param)
    ctx = b.Context(0)
    info = ctx.load_index(base)
    L = oracle()
    idx = Index()
    assert L.bt2o_index_load(C.byref(idx), base.encode()) == 0
    assert info.len == idx.fwd.len and info.off_size == idx.fwd.off_size
    yield ctx, L, idx, refs
    ctx.close()


def edge_reads(refs):
    rnd = random.Random(3)
    reads = synth_reads(refs, 300, 100, seed=31) + synth_reads(refs, 100, 150, seed=32, sub=0, ins=0, dele=0)
    reads += synth_reads(refs, 100, 50, seed=33, n_rate=0.02) + synth_reads(refs, 60, 30, seed=34, len_jitter=25)
    reads += [("e%d" % i, "".join(rnd.choice("ACGT") for _ in range(k)), "I" * k) for i, k in enumerate([1, 2, 3, 4, 5, 9, 10, 11, 12])]
    reads += [("empty", "", ""), ("allN", "N" * 40, "I" * 40), ("polyA", "A" * 80, "I" * 80)]
    rnd.shuffle(reads)
    return reads


def test_exact_sweep(case):
    ctx, L, idx, refs = case
    reads = edge_reads(refs)
    batch = ctx.upload_reads([encode(s) for _, s, _ in reads], [q.encode() for _, _, q in reads])
    for nofw, norc in ((False, False), (True, False), (False, True)):
        ctx.counters(reset=True)
        out = b.structs_from_tensor(ctx.exact_sweep(batch, nofw, norc, 2), b.SweepOut)
        so = SweepOut()
        tot_bwops = tot_rank = 0
        for k, (_, s, _) in enumerate(reads):
            if len(s) == 0:
                assert list(out[k].hit) == [0, 0] and list(out[k].mine) == [0, 0]
                continue
            L.bt2o_exact_sweep(C.byref(idx.fwd), encode(s), encode(revcomp(s)), len(s), int(nofw), int(norc), 2, C.byref(so))
            tot_bwops += so.bwops
            tot_rank += so.nrank
            for f in range(2):
                assert out[k].mine[f] == so.mine[f], (k, s)
                assert out[k].hit[f] == so.hit[f], (k, s)
                if so.hit[f]:
                    assert (out[k].top[f], out[k].bot[f]) == (so.top[f], so.bot[f]), (k, s)
        cnt = ctx.counters()
        assert cnt.bwops == tot_bwops          # same number of BW ops as the reference counts (AlBWOp parity)
        assert cnt.rank_queries == tot_rank    # and the roofline byte accounting agrees with the oracle's


def one_mm_reads(refs, seed):
    """Reads oneMmSearch has something to say about (the mix of tests/test_oracle_vs_ref.py::one_mm_reads): one substitution anywhere incl. the
    ends and around the middle, one N, exact copies, two substitutions, low complexity, very short."""
    rnd = random.Random(seed)
    g = [s for _, s in refs]
    out = []
    for k in range(600):
        s = g[rnd.randrange(len(g))]
        L = rnd.choice([2, 3, 5, 9, 10, 11, 12, 20, 21, 33, 50, 100, 101, 150, 250])
        if L >= len(s):
            continue
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
            r = list(rnd.choice("ACGT") * L)
        r = "".join(r)
        if rnd.random() < 0.5:
            r = revcomp(r)
        out.append(("m%d" % k, r, "".join(rnd.choice("I5+#?") for _ in r)))
    return out


@pytest.mark.parametrize("mode", [["--sensitive"], ["--local"], ["--sensitive", "--nofw"], ["--sensitive", "--norc", "--ignore-quals"]], ids=["e2e", "local", "nofw", "norc"])
def test_one_mm_search(case, mode):
    """k_one_mm_tasks / k_one_mm_scan / k_one_mm_cont / k_one_mm_fin through bt2g_one_mm_search against bt2o_one_mm_search (which
    tests/test_oracle_vs_ref.py pins to SeedAligner::oneMmSearch itself): per (read, strand, index direction) the same hits in the same order --
    range, score, mismatch position from the 5' end, reference and read character.  The kernels get the parameters the product derives from
    its command line (bt2g_cli_params) and the device's own exact sweep; a strand whose sweep proved two edits is not searched, as in the
    reference's call site (bt2_search.cpp:3704-3706)."""
    import numpy as np
    import torch
    ctx, L, idx, refs = case
    reads = [r for r in one_mm_reads(refs, 41) if len(r[1]) >= 1]
    batch = ctx.upload_reads([encode(s) for _, s, _ in reads], [q.encode() for _, _, q in reads])
    large = idx.fwd.off_size == 8
    P = None
    rps = (b.ReadParams * len(reads))()
    for k, (_, s, _) in enumerate(reads):
        P, rp = b.cli_params(mode, len(s), large_index=large)
        rps[k] = rp
    rparams = torch.from_numpy(np.frombuffer(bytes(rps), dtype=np.uint8).copy()).to(batch.seq.device)
    sweep_t = ctx.exact_sweep(batch, bool(P.nofw), bool(P.norc), 2)
    sweep = b.structs_from_tensor(sweep_t, b.SweepOut)
    cap = 16
    hits_t, cnt_t = ctx.one_mm_search(batch, rparams, P, sweep_t, cap)
    hits = b.structs_from_tensor(hits_t, b.Mm1Hit)
    cnt = cnt_t.cpu().numpy()
    sc = Scoring()
    L.bt2o_scoring_default(C.byref(sc))
    sc.match_bonus = P.match_bonus
    sc.mm_pen_type, sc.mm_max, sc.mm_min, sc.n_pen = P.mm_type, P.mm_max, P.mm_min, P.n_pen      # (--ignore-quals: constant penalty MX)
    local = P.match_bonus > 0
    ohits = (Mm1Hit * 512)()
    total = 0
    for k, (_, s, q) in enumerate(reads):
        rp = rps[k]
        for strand in range(2):
            searched = (rp.filt & 15) == 15 and len(s) >= 2 and sweep[k].mine[strand] <= 1 and not (P.nofw if strand == 0 else P.norc)
            want = [[], []]
            if searched:
                n = L.bt2o_one_mm_search(C.byref(idx.fwd), C.byref(idx.bwd), encode(s), q.encode(), len(s), C.byref(sc), rp.nceil, rp.minsc,
                                         int(strand != 0), int(strand != 1), int(local), 0, 1, ohits, 512)
                assert n <= 512
                for i in range(n):
                    h = ohits[i]
                    want[0 if h.ebwtfw else 1].append((h.top, h.bot, h.score, h.off5p, h.chr, h.qchr))
            for d in range(2):
                l = k * 4 + strand * 2 + d
                if len(want[d]) > cap:
                    assert cnt[l] == 255, (k, s, strand, d)
                    continue
                got = [(hits[l * cap + i].top, hits[l * cap + i].bot, hits[l * cap + i].score, hits[l * cap + i].epos, hits[l * cap + i].echr, hits[l * cap + i].eqchr) for i in range(int(cnt[l]))]
                assert got == want[d], (k, s, strand, d, got, want[d])
                total += len(got)
    assert total > 50      # (the tiny genome in local mode yields about 90)


def test_seed_search(case):
    import torch
    ctx, L, idx, refs = case
    reads = edge_reads(refs)
    batch = ctx.upload_reads([encode(s) for _, s, _ in reads], [q.encode() for _, _, q in reads])
    dev = batch.seq.device
    n = len(reads)
    sh = SeedHit()
    for (sl, iv, off) in ((22, 12, 0), (20, 7, 3), (10, 4, 1), (32, 1, 0)):
        lens = [len(s) for _, s, _ in reads]
        max_seeds = max(1, max((1 + max(0, (ln - off - min(sl, ln))) // iv) for ln in lens))
        t = lambda v: torch.full((n,), v, dtype=torch.int32, device=dev)
        ctx.counters(reset=True)
        out = b.structs_from_tensor(ctx.seed_search_exact(batch, t(sl), t(iv), t(off), max_seeds), b.SeedHit)
        tot_bw = 0
        for r, (_, s, _) in enumerate(reads):
            ln = len(s)
            eff = min(sl, ln)
            ns = 0
            if ln > 0 and not (off > 0 and eff + off > ln):
                ns = 1 + ((ln - off - eff) // iv if ln - off > eff else 0)
            for fwi in range(2):
                for i in range(max_seeds):
                    h = out[(r * 2 + fwi) * max_seeds + i]
                    got = (h.topf, h.botf, h.topb, h.botb)
                    want = (0, 0, 0, 0)
                    if i < ns:
                        sub = s[i * iv + off:i * iv + off + eff]
                        if fwi:
                            sub = revcomp(sub)
                        if "N" not in sub:
                            L.bt2o_seed_search_exact(C.byref(idx.fwd), C.byref(idx.bwd), encode(sub), len(sub), C.byref(sh))
                            tot_bw += sh.bwops
                            if sh.botf > sh.topf:
                                want = (sh.topf, sh.botf, sh.topb, sh.botb)
                    assert got == want, (r, s, sl, iv, off, fwi, i)
        assert ctx.counters().bwops == tot_bw


def test_resolve_offsets(case):
    import numpy as np
    import torch
    ctx, L, idx, refs = case
    n = idx.fwd.len
    rnd = random.Random(4)
    rows = [0, 1, n - 1, n, idx.fwd.zoff] + [rnd.randrange(0, n + 1) for _ in range(3000)]
    qlen = [rnd.choice([1, 10, 22, 100]) for _ in rows]
    dev = torch.device("cuda", 0)
    for rej in (False, True):
        out = b.structs_from_tensor(ctx.resolve_offsets(torch.tensor(rows, dtype=torch.int64, device=dev),
                                                        torch.tensor(qlen, dtype=torch.int32, device=dev), rej), b.Resolved)
        ns = u64()
        for k, row in enumerate(rows):
            jo = L.bt2o_get_offset(C.byref(idx.fwd), row, C.byref(ns))
            assert out[k].joined_off == jo and out[k].steps == ns.value
            if jo + qlen[k] > n:
                continue    # the reference never asks for hits running past the joined text
            t = [u64(), u64(), u64()]
            s = C.c_int()
            L.bt2o_joined_to_text_off(C.byref(idx.fwd), qlen[k], jo, C.byref(t[0]), C.byref(t[1]), C.byref(t[2]), int(rej), C.byref(s))
            want_t = 2**64 - 1 if t[0].value == idx.fwd.off_mask else t[0].value
            assert out[k].tidx == want_t and out[k].straddled == s.value
            if want_t != 2**64 - 1:
                assert (out[k].toff, out[k].tlen) == (t[1].value, t[2].value)


def run_dp_fill(ctx, probs, scoring=None):
    """probs: list of (kind, rd codes bytes, quality ASCII bytes, rf masks bytes [cols + 1], minsc) through bt2g_dp_fill -- the worker's own
    fill functions as a stage.  Returns per problem a dict: header fields + lastrow / pred (kind 0) or H, E, F (kinds 1, 2)."""
    import numpy as np
    import torch
    dev = torch.device("cuda", 0)
    L = b.lib()
    rd = b"".join(p[1] for p in probs)
    qu = b"".join(p[2] for p in probs)
    rf = b"".join(p[3] for p in probs)
    arr = (b.DpProblem * len(probs))()
    ro = fo = oo = 0
    for k, p in enumerate(probs):
        rows, cols = len(p[1]), len(p[3]) - 1
        arr[k] = b.DpProblem(ro, fo, rows, cols, p[4], p[0], oo)
        ro += rows
        fo += cols + 1
        oo += L.bt2g_dp_out_bytes(p[0], rows, cols)
    to = lambda x: torch.from_numpy(np.frombuffer(bytes(x) + b"\0", dtype=np.uint8).copy()).to(dev)
    out = torch.zeros(oo + 8, dtype=torch.uint8, device=dev)
    ctx.dp_fill(to(bytes(arr))[:-1], to(rd), to(qu), to(rf), out, scoring)
    torch.cuda.synchronize()
    raw = out.cpu().numpy().tobytes()
    res = []
    for k, p in enumerate(probs):
        rows, cols = len(p[1]), len(p[3]) - 1
        o = arr[k].out_off
        hdr = b.DpOut.from_buffer_copy(raw[o:o + C.sizeof(b.DpOut)])
        body = o + C.sizeof(b.DpOut)
        r = {"best": hdr.best, "lastsolcol": hdr.lastsolcol, "sat8": hdr.sat8, "band_lo": hdr.band_lo, "band_w": hdr.band_w, "has_matrix": hdr.has_matrix}
        if p[0] in (b.DP_EE_U8, b.DP_EE_I16_BAND):
            c4 = (cols + 3) & ~3
            r["lastrow"] = np.frombuffer(raw[body:body + 2 * c4], dtype="<i2")[:cols]
            if hdr.has_matrix:
                r["pred"] = np.frombuffer(raw[body + 2 * c4:body + 2 * c4 + rows * hdr.band_w], dtype=np.uint8).reshape(rows, hdr.band_w)
        elif p[0] == b.DP_LOCAL:
            r["pred"] = np.frombuffer(raw[body:body + rows * cols], dtype=np.uint8).reshape(rows, cols)
        else:
            m = np.frombuffer(raw[body:body + 12 * rows * cols], dtype="<i4").reshape(3, rows, cols)
            r["H"], r["E"], r["F"] = m[0], m[1], m[2]
        res.append(r)
    return res


def oracle_fill_kind(L, kind, sc, rd, phred, rf, cols, minsc):
    import numpy as np
    i32p = C.POINTER(C.c_int32)
    L.bt2o_sw_fill_kind.argtypes = [C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int64, i32p, i32p, i32p,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.bt2o_sw_fill_kind.restype = C.c_int64
    rows = len(rd)
    bufs = [(C.c_int32 * (rows * cols))() for _ in range(3)]
    flag, colstop = C.c_int(), C.c_int()
    got = L.bt2o_sw_fill_kind(kind, C.byref(sc), rd, phred, rows, rf, cols, minsc, bufs[0], bufs[1], bufs[2], C.byref(flag), C.byref(colstop))
    return got, flag.value, colstop.value, [np.frombuffer(x, dtype="<i4").reshape(rows, cols) for x in bufs]


def pred_bits_from_hef(sc, rd, phred, rf, H, E, F, L, bias=0xff):
    """The predecessor byte of every cell from the oracle's H/E/F (the reference's u8 encoding): the questions the reference's backtrace
    asks of its matrices (aligner_swsse_ee_u8.cpp:1330-1520), as the band fill answers them while the neighbours are in registers
    (PB_* in bt2g_align.hpp): 1 H came diagonally, 2 H == E, 4 H == F (both only where gaps are allowed), 8 E opens from H-left,
    16 E extends E-left, 32 F opens from H-up, 64 F extends F-up."""
    import numpy as np
    rows, cols = H.shape
    rdo, rde = sc.rd_gap_const + sc.rd_gap_linear, sc.rd_gap_linear
    rfo, rfe = sc.rf_gap_const + sc.rf_gap_linear, sc.rf_gap_linear
    out = np.zeros((rows, cols), dtype=np.uint8)
    for i in range(rows):
        ga = not (i < sc.gapbar or rows - i - 1 < sc.gapbar)
        for j in range(cols):
            m = rf[j]
            pen = -L.bt2o_score(C.byref(sc), rd[i], m, phred[i])
            h, e, f = int(H[i, j]), int(E[i, j]), int(F[i, j])
            hdiag = bias if i == 0 else (0 if j == 0 else int(H[i - 1, j - 1]))
            hl, el = (int(H[i, j - 1]), int(E[i, j - 1])) if j > 0 else (0, 0)
            hu, fu = (int(H[i - 1, j]), int(F[i - 1, j])) if i > 0 else (0, 0)
            c = 1 if hdiag - pen == h else 0
            c |= 2 if ga and h == e else 0
            c |= 4 if ga and h == f else 0
            c |= 8 if hl - rdo == e else 0
            c |= 16 if el - rde == e else 0
            c |= 32 if hu - rfo == f else 0
            c |= 64 if fu - rfe == f else 0
            out[i, j] = c
    return out


def local_pred_bits_from_hef(sc, rd, phred, rf, H, E, F, L):
    """Predecessor byte of every cell of a LOCAL fill from the oracle's plain scores: the questions of the local backtrace
    (aligner_swsse_loc_u8.cpp:1530-1660) -- as pred_bits_from_hef, but a neighbour only counts while its score is above the floor (0)."""
    import numpy as np
    rows, cols = H.shape
    rdo, rde = sc.rd_gap_const + sc.rd_gap_linear, sc.rd_gap_linear
    rfo, rfe = sc.rf_gap_const + sc.rf_gap_linear, sc.rf_gap_linear
    out = np.zeros((rows, cols), dtype=np.uint8)
    for i in range(rows):
        ga = not (i < sc.gapbar or rows - i - 1 < sc.gapbar)
        for j in range(cols):
            s = L.bt2o_score(C.byref(sc), rd[i], rf[j], phred[i])
            h, e, f = int(H[i, j]), int(E[i, j]), int(F[i, j])
            hdiag = int(H[i - 1, j - 1]) if (i > 0 and j > 0) else 0
            hl, el = (int(H[i, j - 1]), int(E[i, j - 1])) if j > 0 else (0, 0)
            hu, fu = (int(H[i - 1, j]), int(F[i - 1, j])) if i > 0 else (0, 0)
            c = 1 if (hdiag > 0 and hdiag + s == h) else 0
            c |= 2 if ga and h == e else 0
            c |= 4 if ga and h == f else 0
            c |= 8 if (hl > 0 and hl - rdo == e) else 0
            c |= 16 if (el > 0 and el - rde == e) else 0
            c |= 32 if (hu > 0 and hu - rfo == f) else 0
            c |= 64 if (fu > 0 and fu - rfe == f) else 0
            out[i, j] = c
    return out


def test_dp_fills_that_ship(case):
    """The fills k_align_reads runs -- the banded 8-bit end-to-end fill that stores predecessor bits (every pairs-per-lane class the window
    sizes allow), the 16-bit end-to-end fill, the local fill -- through bt2g_dp_fill, against the oracle's restatement of the reference's
    kernels (pinned to reference-recorded matrices by tests/test_oracle_golden.py)."""
    import numpy as np
    ctx, L, idx, refs = case
    sc = Scoring()
    L.bt2o_scoring_default(C.byref(sc))
    rnd = random.Random(11)

    def problem(rows, cols, match=0.9, n_at=None):
        rdc = bytes(rnd.choice([0, 1, 2, 3, 0, 1, 2, 3, 4]) for _ in range(rows))
        phred = bytes(rnd.choice([2, 12, 20, 30, 38, 40]) for _ in range(rows))
        rfm = bytearray(1 << rnd.randrange(4) for _ in range(cols + 1))
        off = max(0, (cols - rows) // 2)
        for i in range(min(rows, cols - off)):
            if rnd.random() < match and rdc[i] < 4:
                rfm[i + off] = 1 << rdc[i]
        if n_at is not None:
            rfm[n_at] = 16
        return rdc, phred, bytes(rfm)

    # ---- 8-bit end to end, band = the whole rectangle (minsc -254 allows 83 reference gaps: every diagonal of a <= 84-row problem is in
    #      the band), so EVERY stored predecessor byte can be checked; columns chosen to hit each pairs-per-lane class (RP 1,2,3,4,6,8,12)
    probs, shapes = [], []
    for rows, cols in [(1, 1), (2, 5), (30, 61), (64, 64), (84, 100), (70, 200), (80, 300), (84, 400), (60, 640), (84, 900), (50, 1090)]:
        rdc, phred, rfm = problem(rows, cols, n_at=(cols // 3 if cols > 40 else None))
        probs.append((b.DP_EE_U8, rdc, bytes(q + 33 for q in phred), rfm, -254))
        shapes.append((rdc, phred, rfm))
    res = run_dp_fill(ctx, probs)
    rp_seen = set()
    for (kind, rdc, qa, rfm, minsc), (rdc_, phred, _), r in zip(probs, shapes, res):
        rows, cols = len(rdc), len(rfm) - 1
        Hb, Eb, Fb = (C.create_string_buffer(rows * cols) for _ in range(3))
        want = L.bt2o_sw_fill_ee_u8(C.byref(sc), rdc, phred, rows, rfm, cols, Hb, Eb, Fb)
        H, E, F = (np.frombuffer(x.raw, dtype=np.uint8).reshape(rows, cols) for x in (Hb, Eb, Fb))
        assert r["best"] == want and r["has_matrix"] == (1 if want >= minsc else 0), (rows, cols, r["best"], want)
        if not r["has_matrix"]:
            continue
        assert r["band_lo"] == rows - 1 and r["band_w"] >= rows + cols - 1, (rows, cols, r["band_lo"], r["band_w"])
        rp_seen.add(r["band_w"] // 128)
        assert list(r["lastrow"]) == [int(v) - 0xff for v in H[rows - 1]], (rows, cols)
        want_pred = pred_bits_from_hef(sc, rdc, phred, rfm, H, E, F, L)
        got = np.zeros((rows, cols), dtype=np.uint8)
        for i in range(rows):
            got[i] = r["pred"][i, rows - 1 - i:rows - 1 - i + cols]       # cell (i, j) at byte j - i + band_lo
        assert (got == want_pred).all(), (rows, cols, np.argwhere(got != want_pred)[:5])
    assert rp_seen >= {1, 2, 3, 4, 6, 8}, rp_seen
    # ---- 8-bit end to end, a real band (the worker's case: 150-bp read, minsc -90 and tighter): best and every last-row score that
    #      reaches minsc equal the full-rectangle fill's
    probs, shapes = [], []
    for rows, cols, minsc, match in [(150, 211, -90, 0.93), (150, 211, -90, 0.6), (150, 211, -30, 0.97), (100, 161, -60, 0.9), (250, 311, -150, 0.92), (400, 461, -240, 0.95),
                                     (150, 700, -90, 0.93), (36, 97, -22, 0.95)]:
        for _ in range(3):
            rdc, phred, rfm = problem(rows, cols, match)
            probs.append((b.DP_EE_U8, rdc, bytes(q + 33 for q in phred), rfm, minsc))
            shapes.append(phred)
    res = run_dp_fill(ctx, probs)
    n_pass = 0
    for (kind, rdc, qa, rfm, minsc), phred, r in zip(probs, shapes, res):
        rows, cols = len(rdc), len(rfm) - 1
        Hb, Eb, Fb = (C.create_string_buffer(rows * cols) for _ in range(3))
        want = L.bt2o_sw_fill_ee_u8(C.byref(sc), rdc, phred, rows, rfm, cols, Hb, Eb, Fb)
        H = np.frombuffer(Hb.raw, dtype=np.uint8).reshape(rows, cols)
        if want >= minsc:
            n_pass += 1
            assert r["best"] == want and r["has_matrix"] == 1, (rows, cols, minsc, r["best"], want)
            for j in range(cols):
                if int(H[rows - 1, j]) - 0xff >= minsc:
                    assert int(r["lastrow"][j]) == int(H[rows - 1, j]) - 0xff, (rows, cols, j)
                else:
                    assert int(r["lastrow"][j]) < minsc, (rows, cols, j)
        else:
            assert r["best"] < minsc and r["has_matrix"] == 0, (rows, cols, minsc, r["best"], want)
    assert n_pass >= 8
    # ---- the 16-bit end-to-end kernel's arithmetic on the band (what the worker runs for minimum scores below -254 wherever the band has at most
    #      2 048 diagonals): band = the whole rectangle (every predecessor byte checked against the oracle's 16-bit matrices, as unsigned values
    #      with a bias of 0xffff), and real bands of long reads (best and every last-row score that reaches minsc)
    osc = Scoring()
    L.bt2o_scoring_default(C.byref(osc))
    probs, shapes = [], []
    for rows, cols in [(1, 1), (2, 5), (30, 61), (64, 64), (84, 100), (70, 200), (84, 400), (60, 640), (84, 900)]:
        rdc, phred, rfm = problem(rows, cols, n_at=(cols // 3 if cols > 40 else None))
        probs.append((b.DP_EE_I16_BAND, rdc, bytes(q + 33 for q in phred), rfm, -300))      # 99 reference gaps: every diagonal of a <= 84-row problem
        shapes.append((rdc, phred, rfm))
    for rows, cols, minsc, match in [(300, 361, -181, 0.93), (450, 511, -271, 0.95), (450, 511, -271, 0.7), (512, 573, -308, 0.96), (500, 1200, -301, 0.94)]:
        rdc, phred, rfm = problem(rows, cols, match)
        probs.append((b.DP_EE_I16_BAND, rdc, bytes(q + 33 for q in phred), rfm, minsc))
        shapes.append((rdc, phred, rfm))
    res = run_dp_fill(ctx, probs)
    n_full = n_pass = 0
    for (kind, rdc, qa, rfm, minsc), (rdc_, phred, _), r in zip(probs, shapes, res):
        rows, cols = len(rdc), len(rfm) - 1
        got, flag, colstop, (H, E, F) = oracle_fill_kind(L, 1, osc, rdc, phred, rfm, cols, minsc)
        want = int(H[rows - 1].max()) - 0x7fff
        Hu, Eu, Fu = H + 32768, E + 32768, F + 32768      # signed 16-bit with bias 0x7fff -> unsigned with bias 0xffff (0 = minus infinity)
        if want < minsc:
            assert r["best"] < minsc and r["has_matrix"] == 0, (rows, cols, minsc, r["best"], want)
            continue
        assert r["best"] == want and r["has_matrix"] == 1, (rows, cols, minsc, r["best"], want)
        n_pass += 1
        for j in range(cols):
            v = int(Hu[rows - 1, j]) - 0xffff
            if v >= minsc:
                assert int(r["lastrow"][j]) == v, (rows, cols, j)
            else:
                assert int(r["lastrow"][j]) < minsc, (rows, cols, j)
        if rows <= 84:
            assert r["band_lo"] == rows - 1 and r["band_w"] >= rows + cols - 1, (rows, cols, r["band_lo"], r["band_w"])
            want_pred = pred_bits_from_hef(osc, rdc, phred, rfm, Hu, Eu, Fu, L, bias=0xffff)
            gotp = np.zeros((rows, cols), dtype=np.uint8)
            for i in range(rows):
                gotp[i] = r["pred"][i, rows - 1 - i:rows - 1 - i + cols]
            assert (gotp == want_pred).all(), (rows, cols, np.argwhere(gotp != want_pred)[:5])
            n_full += 1
    assert n_full >= 8 and n_pass >= 11, (n_full, n_pass)
    # ---- 16-bit end to end (the anti-diagonal cell form) and local: every cell, on the reference-recorded problems of tests/golden/dp_kinds_golden.json + random shapes
    with open(os.path.join(GOLD, "dp_kinds_golden.json")) as f:
        gold = json.load(f)
    probs, meta = [], []
    for p in gold:
        rows, cols = p["rows"], p["cols"]
        rdc, phred = encode(p["rd"]), bytes(ord(c) - 33 for c in p["qu"])
        rfm = bytes(1 << "ACGTN".index(c) for c in p["rf"]) + b"\x10"
        kind = b.DP_EE_I16 if p["kind"] == 1 else b.DP_LOCAL
        probs.append((kind, rdc, p["qu"].encode(), rfm, p["minsc"]))
        meta.append((p["match_bonus"], phred))
    for rows in (1, 3, 64, 65, 129, 200, 330, 449, 512):      # every rows-per-lane class of the anti-diagonal fills
        cols = rows + rnd.choice([1, 30, 61])
        rdc, phred, rfm = problem(rows, cols, 0.92)
        probs.append((b.DP_EE_I16, rdc, bytes(q + 33 for q in phred), rfm, -int(0.6 * rows) - 1)); meta.append((0, phred))
        probs.append((b.DP_LOCAL, rdc, bytes(q + 33 for q in phred), rfm, 20 + int(8 * np.log(rows))) if rows > 2 else (b.DP_LOCAL, rdc, bytes(q + 33 for q in phred), rfm, 1)); meta.append((2, phred))
    by_bonus = {}
    for k, (pr, (ma, phred)) in enumerate(zip(probs, meta)):
        by_bonus.setdefault((ma, pr[0]), []).append(k)
    out = [None] * len(probs)
    for (ma, kind), ks in by_bonus.items():
        dsc = b.Scoring()
        b.lib().bt2g_scoring_default(C.byref(dsc))
        dsc.match_bonus = ma
        for k, r in zip(ks, run_dp_fill(ctx, [probs[k] for k in ks], dsc)):
            out[k] = r
    for (kind, rdc, qa, rfm, minsc), (ma, phred), r in zip(probs, meta, out):
        rows, cols = len(rdc), len(rfm) - 1
        osc = Scoring()
        L.bt2o_scoring_default(C.byref(osc))
        osc.match_bonus = ma
        if kind == b.DP_EE_I16:
            got, flag, colstop, (H, E, F) = oracle_fill_kind(L, 1, osc, rdc, phred, rfm, cols, minsc)
            assert (r["H"] == H).all() and (r["E"] == E).all() and (r["F"] == F).all(), (rows, cols)
            assert r["best"] == int(H[rows - 1].max()) - 0x7fff, (rows, cols)
        else:
            got3, flag3, colstop3, (H, E, F) = oracle_fill_kind(L, 3, osc, rdc, phred, rfm, cols, minsc)
            got2, flag2, colstop2, _ = oracle_fill_kind(L, 2, osc, rdc, phred, rfm, cols, minsc)
            n = colstop3
            # the local fill stores predecessor bits, not scores: derive them from the oracle's matrices (plain scores = the 16-bit kernel's
            # cell + 32768) with the local kernels' rule that a neighbour at the floor (0) is no predecessor
            Hs, Es, Fs = H[:, :n] + 32768, E[:, :n] + 32768, F[:, :n] + 32768
            want_pred = local_pred_bits_from_hef(osc, rdc, phred, rfm, Hs, Es, Fs, L)
            # a bit is defined where its state can be entered: H bits where H > 0, E bits where E > 0, F bits where F > 0
            care = np.where(Hs > 0, 7, 0) | np.where(Es > 0, 24, 0) | np.where(Fs > 0, 96, 0)
            got_pred = r["pred"][:, :n]
            assert ((got_pred & care) == (want_pred & care)).all(), (rows, cols, np.argwhere((got_pred & care) != (want_pred & care))[:5])
            colmax = (H[:, :n] + 32768).max(axis=0)
            assert r["best"] == int(colmax.max()), (rows, cols)
            sol = [j for j in range(n) if colmax[j] >= minsc]
            assert r["lastsolcol"] == (sol[-1] if sol else 0), (rows, cols)
            assert r["sat8"] == (1 if flag2 == -2 else 0), (rows, cols, flag2)


def test_roundtrip_property_large(case):
    """Size-independent property at a size the scalar oracle would not finish quickly: every error-free
    read sampled from the genome must be found by the exact sweep on the strand it was sampled from,
    and resolving every row of the reported range must give back (among others) the sampled position."""
    import numpy as np
    import torch
    ctx, L, idx, refs = case
    rnd = random.Random(12)
    reads, truth = [], []
    N = 20000
    while len(reads) < N:
        ti = rnd.randrange(len(refs))
        g = refs[ti][1]
        ln = rnd.choice([30, 50, 75])
        p = rnd.randrange(0, len(g) - ln)
        s = g[p:p + ln]
        if "N" in s:
            continue
        rc = rnd.random() < 0.5
        reads.append(revcomp(s) if rc else s)
        truth.append((ti, p, rc, ln))
    batch = ctx.upload_reads([encode(s) for s in reads])
    out = b.structs_from_tensor(ctx.exact_sweep(batch), b.SweepOut)
    rows, qlen, owner = [], [], []
    for k, (ti, p, rc, ln) in enumerate(truth):
        f = 1 if rc else 0
        assert out[k].hit[f] == 1 and out[k].mine[f] == 0, k
        top, bot = out[k].top[f], out[k].bot[f]
        assert bot > top
        for row in range(top, min(bot, top + 64)):
            rows.append(row)
            qlen.append(ln)
            owner.append(k)
    dev = torch.device("cuda", 0)
    res = b.structs_from_tensor(ctx.resolve_offsets(torch.tensor(rows, dtype=torch.int64, device=dev),
                                                    torch.tensor(qlen, dtype=torch.int32, device=dev), True), b.Resolved)
    found = set()
    small = set()
    for k, r in zip(owner, res):
        ti, p, rc, ln = truth[k]
        if r.tidx == ti and r.toff == p:
            found.add(k)
    for k in range(N):
        f = 1 if truth[k][2] else 0
        if out[k].bot[f] - out[k].top[f] <= 64:
            small.add(k)
    assert small <= found

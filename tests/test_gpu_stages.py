"""Parity of the HIP stage kernels (through the C ABI) against the plain-C oracle on seeded inputs,
against the committed golden vectors, and -- at larger sizes -- through round-trip properties."""
import ctypes as C
import json
import os
import random

import pytest

pytestmark = pytest.mark.gpu

import bowtie2_amd as b
from bt2test import (Index, Scoring, SeedHit, SweepOut, cached_synth_index, encode, oracle, revcomp, sha,
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


def run_dp(ctx, probs):
    """probs: list of (rd codes bytes, phred bytes, rf masks bytes[cols]) -> list of (best, H|E|F bytes)"""
    import numpy as np
    import torch
    dev = torch.device("cuda", 0)
    rd = b"".join(p[0] for p in probs)
    qu = b"".join(p[1] for p in probs)
    rf = b"".join(p[2] for p in probs)
    arr = (b.DpProblem * len(probs))()
    ro = fo = mo = 0
    for k, p in enumerate(probs):
        rows, cols = len(p[0]), len(p[2])
        arr[k] = b.DpProblem(ro, rows, fo, cols, mo)
        ro += rows
        fo += cols
        mo += 3 * rows * cols
    to = lambda x: torch.from_numpy(np.frombuffer(bytes(x) + b"\0", dtype=np.uint8).copy()).to(dev)
    d_probs = to(bytes(arr))
    mat = torch.zeros(mo + 1, dtype=torch.uint8, device=dev)
    best = torch.zeros(len(probs), dtype=torch.int32, device=dev)
    ctx.sw_fill_ee_u8(d_probs[:-1], to(rd), to(qu), to(rf), mat, best)
    m = mat.cpu().numpy().tobytes()
    bs = best.cpu().tolist()
    res = []
    for k, p in enumerate(probs):
        sz = 3 * len(p[0]) * len(p[2])
        res.append((bs[k], m[arr[k].mat_off:arr[k].mat_off + sz]))
    return res


def test_dp_fill_golden_and_random(case):
    ctx, L, idx, refs = case
    sc = Scoring()
    L.bt2o_scoring_default(C.byref(sc))
    with open(os.path.join(GOLD, "dp_golden.json")) as f:
        gold = json.load(f)
    probs = [(encode(p["rd"]), bytes(ord(c) - 33 for c in p["qu"]), bytes(1 << "ACGTN".index(c) for c in p["rf"])) for p in gold]
    res = run_dp(ctx, probs)
    for p, (best, mat) in zip(gold, res):
        assert best == p["best"], (p["rows"], p["cols"])
        assert sha(mat) == p["sha"], (p["rows"], p["cols"])   # bit-exact H, E and F vs the reference's SSE fill
    # random shapes incl. every rows-per-lane class (1..8), ragged
    rnd = random.Random(6)
    probs = []
    for rows in [1, 2, 3, 63, 64, 65, 127, 129, 150, 191, 193, 250, 256, 300, 400, 449, 512]:
        cols = rows + rnd.choice([0, 1, 30, 60])
        rdc = bytes(rnd.choice([0, 1, 2, 3, 0, 1, 2, 3, 4]) for _ in range(rows))
        rfm = bytearray(1 << rnd.randrange(4) for _ in range(cols))
        for i in range(min(rows, cols)):   # make the diagonal mostly match so scores stay off the floor
            if rnd.random() < 0.9 and rdc[i] < 4:
                rfm[i + (cols - rows) // 2] = 1 << rdc[i]
        if rnd.random() < 0.5:
            rfm[rnd.randrange(cols)] = 16
        probs.append((rdc, bytes(rnd.choice([2, 12, 20, 30, 38, 40, 41]) for _ in range(rows)), bytes(rfm)))
    res = run_dp(ctx, probs)
    for (rdc, q, rfm), (best, mat) in zip(probs, res):
        rows, cols = len(rdc), len(rfm)
        H = C.create_string_buffer(rows * cols)
        E = C.create_string_buffer(rows * cols)
        F = C.create_string_buffer(rows * cols)
        want = L.bt2o_sw_fill_ee_u8(C.byref(sc), rdc, q, rows, rfm, cols, H, E, F)
        assert best == want, (rows, cols)
        assert mat == H.raw + E.raw + F.raw, (rows, cols)


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

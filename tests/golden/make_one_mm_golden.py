#!/usr/bin/env python3
"""Regenerates tests/golden/one_mm_golden.json: reads against the tiny golden indexes (both widths) with what the REFERENCE's
SeedAligner::oneMmSearch (aligner_seed.cpp:975-1325, through oracle/ref_shim.cpp: ref_one_mm) adds to its SeedResults for them -- the
1-mismatch end-to-end hits in order (range, score, mismatch position from the 5' end, reference and read character, strand) and the exact
hits -- end to end and with the local validity rule, with / without exact hits reported, one strand switched off.  Needs oracle/_ref
(run where /root/reference exists).  tests/test_oracle_golden.py::test_one_mm_search checks bt2o_one_mm_search against it anywhere."""
import ctypes as C
import json
import math
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from bt2test import refshim, revcomp, u64  # noqa


def reads_for(g, seed):
    rnd = random.Random(seed)
    out = []
    for k in range(150):
        L = rnd.choice([2, 3, 5, 9, 10, 11, 12, 20, 21, 33, 50, 80])
        p = rnd.randrange(0, len(g) - L)
        r = list(g[p:p + L])
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
        out.append([r, "".join(rnd.choice("I5+#?") for _ in r)])
    return out


def main():
    g = "".join(l.strip() for l in open(os.path.join(HERE, "tiny.fa")) if not l.startswith(">"))
    gold = {}
    for large in (False, True):
        R = refshim(large)
        h = R.ref_open(os.path.join(HERE, "tiny_l" if large else "tiny_s").encode())
        assert h
        out = (u64 * (7 * 512))()
        ex = (u64 * 6)()
        cases = []
        for local in (0, 1):
            R.ref_set_match_bonus(h, 2 if local else 0)
            for k, (s, q) in enumerate(reads_for(g, 5 + local)):
                n = len(s)
                nceil = min(int(0 + 0.15 * n), n)
                minsc = (20 + int(8.0 * math.log(n))) if local else int(-0.6 - 0.6 * n)
                nofw, norc = int(k % 11 == 3), int(k % 13 == 5)
                for repex, rep1mm in ((0, 1), (1, 1), (1, 0)):
                    nh = R.ref_one_mm(h, s.encode(), q.encode(), minsc, nofw, norc, local, repex, rep1mm, out, 512, ex)
                    assert nh <= 512
                    hits = [[int(v) if j != 2 else int(C.c_int64(v).value) for j, v in enumerate(out[i * 7:i * 7 + 7])] for i in range(nh)]
                    cases.append({"seq": s, "qual": q, "local": local, "minsc": minsc, "nceil": nceil, "nofw": nofw, "norc": norc, "repex": repex, "rep1mm": rep1mm,
                                  "hits": hits, "exact": [int(v) for v in ex]})
        R.ref_set_match_bonus(h, 0)
        R.ref_close(h)
        gold["l" if large else "s"] = cases
    with open(os.path.join(HERE, "one_mm_golden.json"), "w") as f:
        json.dump(gold, f, separators=(",", ":"))
    print({k: (len(v), sum(len(c["hits"]) for c in v)) for k, v in gold.items()})


if __name__ == "__main__":
    main()

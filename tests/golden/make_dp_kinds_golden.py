#!/usr/bin/env python3
"""Regenerates tests/golden/dp_kinds_golden.json: DP problems for the 16-bit end-to-end fill and the two local fills, with the
sha256 of the H|E|F words the REFERENCE's SSE kernels left in their matrices (alignNucleotidesEnd2EndSseI16 / LocalSseU8 /
LocalSseI16 through oracle/ref_shim.cpp), the returned score and the flag.  Needs oracle/_ref (run where /root/reference exists).
`ncol` = columns the kernel filled before its early stop (local mode); the hash covers those columns, int32 little-endian, row-major
per matrix."""
import ctypes as C
import hashlib
import json
import math
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from bt2test import Scoring, encode, oracle, refshim  # noqa


def cells_sha(bufs, rows, cols, ncol):
    import struct
    h = hashlib.sha256()
    for b in bufs:
        for i in range(rows):
            h.update(struct.pack("<%di" % ncol, *b[i * cols:i * cols + ncol]))
    return h.hexdigest()


def main():
    R = refshim(False)
    L = oracle()
    i32p = C.POINTER(C.c_int32)
    L.bt2o_sw_fill_kind.argtypes = [C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int64, i32p, i32p, i32p,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.bt2o_sw_fill_kind.restype = C.c_int64
    h = R.ref_open(os.path.join(HERE, "tiny_s").encode())
    g = "".join(l.strip() for l in open(os.path.join(HERE, "tiny.fa")) if not l.startswith(">")).replace("N", "A")
    out = []
    for kind in (1, 2, 3):
        local = kind != 1
        sc = Scoring()
        L.bt2o_scoring_default(C.byref(sc))
        if local:
            sc.match_bonus = 2
        R.ref_set_match_bonus(h, 2 if local else 0)
        rnd = random.Random(500 + kind)
        for t in range(24):
            rows = rnd.choice([12, 33, 64, 65, 100, 150, 250] + ([400, 500] if kind != 2 else [120]))
            cols = rows + rnd.choice([0, 5, 12, 60])
            pos = rnd.randrange(0, len(g) - cols - 2)
            window = g[pos:pos + cols + 1]
            rd = list(window[(cols - rows) // 2:(cols - rows) // 2 + rows])
            mm = rnd.choice([0.0, 0.03, 0.03, 0.3, 0.75])
            for i in range(rows):
                if rnd.random() < mm:
                    rd[i] = rnd.choice("ACGTN")
            if rnd.random() < 0.4 and rows > 20:
                p = rnd.randrange(5, rows - 5)
                rd = (rd[:p] + rd[p + rnd.randrange(1, 4):] + list("ACG"))[:rows]
            rd = "".join(rd)
            rows = len(rd)
            qu = "".join(rnd.choice("GGG?5-I#") for _ in range(rows))
            if local:
                minsc = rnd.choice([0, 60, int(20 + 8.0 * math.log(rows)), 2 * rows - 10, 2 * rows + 50])
            else:
                minsc = rnd.choice([-30000, int(-0.6 - 0.6 * rows), -20, -600])
            rf = bytes(1 << "ACGTN".index(c) for c in window)
            n = rows * cols
            bufs = [(C.c_int32 * n)() for _ in range(6)]
            flag, flag2, colstop = C.c_int(), C.c_int(), C.c_int()
            want = R.ref_sw_fill_kind(h, kind, rd.encode(), qu.encode(), rf, cols, minsc, bufs[0], bufs[1], bufs[2], C.byref(flag))
            got = L.bt2o_sw_fill_kind(kind, C.byref(sc), encode(rd), bytes(ord(c) - 33 for c in qu), rows, rf, cols, minsc, bufs[3], bufs[4], bufs[5],
                                      C.byref(flag2), C.byref(colstop))
            assert (got, flag2.value) == (want, flag.value)
            ncol = colstop.value
            out.append({"kind": kind, "rows": rows, "cols": cols, "rd": rd, "qu": qu, "rf": window, "minsc": minsc, "match_bonus": 2 if local else 0,
                        "score": None if want == -2**63 else want, "flag": flag.value, "ncol": ncol, "sha": cells_sha(bufs[:3], rows, cols, ncol)})
    R.ref_set_match_bonus(h, 0)
    R.ref_close(h)
    with open(os.path.join(HERE, "dp_kinds_golden.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(len(out), "problems;", "flags:", sorted(set(p["flag"] for p in out)), "early stops:", sum(p["ncol"] < p["cols"] for p in out))


if __name__ == "__main__":
    main()

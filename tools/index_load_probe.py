#!/usr/bin/env python3
"""How long bt2g_index_load takes on an index the size of the headline's (3 100 Mbp, .bt2l), by phase (BT2G_DEBUG_LOAD=1): streamed from
the files versus through host memory (BT2G_LOAD_SERIAL=1, the path before round 5).  Builds the index of a uniform random genome with the
product's builder first (the load does not care what the text is; no torch here: importing it costs a fresh box a minute or two), runs the
product binary on 2 000 reads of that genome under either load path and compares the SAM.
Usage (GPU box):  python tools/index_load_probe.py [mbp]  > gpurun_out/<tag>/index_load.log 2>&1"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bowtie2_amd", "bin")


def main():
    mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 3100
    d = os.environ.get("BT2_BENCH_CACHE", "/tmp/bt2_amd_bench")
    os.makedirs(d, exist_ok=True)
    base = os.path.join(d, "uniform_%dmbp_bt2l" % mbp)
    fq = base + ".fq"
    if not os.path.exists(base + ".rev.2.bt2l"):
        t0 = time.time()
        rng = np.random.default_rng(5)
        fa = base + ".fa"
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        with open(fa, "wb") as f, open(fq, "wb") as q:
            for c in range(8):
                seq = lut[rng.integers(0, 4, mbp * 1000000 // 8, dtype=np.uint8)]
                f.write(b">chr%d\n" % (c + 1))
                f.write(seq.tobytes())
                f.write(b"\n")
                for i in range(250):
                    o = int(rng.integers(0, len(seq) - 150))
                    q.write(b"@r%d_%d\n" % (c, i) + seq[o:o + 150].tobytes() + b"\n+\n" + b"I" * 150 + b"\n")
        print("genome written in %.1f s" % (time.time() - t0), flush=True)
        t0 = time.time()
        p = subprocess.run([os.path.join(BIN, "bowtie2-build-l"), "-q", fa, base], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        print("index built in %.1f s (rc %d) %s" % (time.time() - t0, p.returncode, p.stdout[-300:]), flush=True)
        os.remove(fa)
    sams = {}
    for tag, env in (("streamed", {}), ("serial", {"BT2G_LOAD_SERIAL": "1"}), ("streamed", {})):
        e = dict(os.environ, BT2G_DEBUG_LOAD="1", **env)
        t0 = time.time()
        p = subprocess.run([os.path.join(BIN, "bowtie2-align-l"), "-t", "-x", base, "-U", fq], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        print("---- %s (rc %d, %.2f s wall)" % (tag, p.returncode, time.time() - t0))
        print(p.stderr, flush=True)
        sams[tag] = [l for l in p.stdout.splitlines() if not l.startswith("@PG")]
    aligned = sum(1 for l in sams["streamed"] if not l.startswith("@") and not int(l.split("\t")[1]) & 4)
    print("SAM identical between the two load paths: %s (%d lines, %d reads aligned)" % (sams["streamed"] == sams["serial"], len(sams["serial"]), aligned))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X multiseed hot path on the headline configuration of BASELINE.json:
150 bp single-end reads, --sensitive, end-to-end, against a genome-scale LARGE (.bt2l: 128-byte sides, 64-bit
offsets) index that cannot sit in the 256 MiB Infinity Cache.

Contract (DESIGN.md "Measurement"):
  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)
A "step" is one bt2g_align_batch over one resident batch of reads (inputs already in HBM).  Rank 0 prints ONE JSON
line with metric/value plus
  roofline      dominant kernel, HBM-bound, timed with HIP events on the launch stream;
  cpu_baseline  the unmodified reference (oracle/_ref/bowtie2-align-l-v256: the AVX2 build users run), all host cores,
                bounded sample, wall clock minus a measured index-load run, median of 3;
  config.parity_checked_reads   SAM of the CPU-baseline sample written by the product binary (GPU) compared with the
                reference's, byte for byte, in this same run.

hg38 itself is not available offline.  The stand-in is a deterministic synthetic genome with hg38-like repeat
content (see synth_genome_gpu): diverged interspersed-repeat families, simple repeats, low-divergence segmental
duplications and N gaps over ~45 % of the sequence.  The index is built in this run by the GPU index builder
(bt2g_index_build_mem; byte-identical to bowtie2-build's, tests/test_index_build.py).  Reads shard across ranks with
no data-path collective ("weak").
"""
import argparse
import json
import math
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def nproc():
    """Usable host cores: affinity mask, clipped by the cgroup CPU quota if there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cache_dir():
    d = os.environ.get("BT2_BENCH_CACHE", "/tmp/bt2_amd_bench")
    os.makedirs(d, exist_ok=True)
    return d


# ------------------------------------------------------------------------------------------------ genome ----
N_CHROMS = 8


def synth_genome_gpu(mbp, seed, device):
    """Deterministic hg38-like synthetic genome as one uint8 tensor of codes 0..3 (4 = N) + chromosome lengths.

    Composition (fractions of the sequence; hg38 for comparison: ~10 % Alu, ~17 % L1, ~20 % other interspersed
    repeats, ~3 % simple repeats, ~5 % segmental duplications, ~5 % N):
      * "Alu-like":  one 300 bp consensus, full-length copies, 4-16 % substitutions each          -> 10 %
      * "L1-like":   one 6 kbp consensus, 3'-anchored truncated copies of 400-6000 bp, 3-20 %     -> 17 %
      * "old" families: eight consensi of 150-1200 bp, 15-28 % substitutions                      -> 12 %
      * simple repeats: units of 1-6 bp, tracts of 20-300 bp                                      ->  2 %
      * segmental duplications: 5-40 kbp copies of earlier sequence with 0.5-2 % substitutions    ->  4 %
      * N: a 0.25 % gap at either end of every chromosome + scattered 100-5000 bp gaps            -> ~0.75 %
    Later classes overwrite earlier ones where they collide."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = mbp * 1_000_000
    per = n // N_CHROMS
    n = per * N_CHROMS
    G = torch.randint(0, 4, (n,), generator=g, device=device, dtype=torch.uint8)

    def rnd(shape):
        return torch.rand(shape, generator=g, device=device)

    def plant(consensus, frac, div_lo, div_hi, min_len=None):
        L = consensus.numel()
        mean_len = L if min_len is None else (L + min_len) / 2.0
        k = max(1, int(n * frac / mean_len))
        for c0 in range(0, k, 200_000):                     # chunks bound the index tensors
            kk = min(200_000, k - c0)
            start = (rnd((kk,)).double() * (n - L - 1)).long()
            ln = torch.full((kk,), L, device=device, dtype=torch.long) if min_len is None else \
                (min_len + (rnd((kk,)) * (L - min_len)).long())
            div = div_lo + rnd((kk,)) * (div_hi - div_lo)
            off = torch.arange(L, device=device).unsqueeze(0)                # [1, L]
            keep = off < ln.unsqueeze(1)                                     # 3'-anchored: the last ln bases of the consensus
            src = (L - ln).unsqueeze(1) + off                                # consensus position
            src = torch.where(keep, src, torch.zeros_like(src))
            vals = consensus[src]
            mut = rnd((kk, L)) < div.unsqueeze(1)
            rb = torch.randint(1, 4, (kk, L), generator=g, device=device, dtype=torch.uint8)
            vals = torch.where(mut, (vals + rb) % 4, vals)
            dst = start.unsqueeze(1) + off
            G[dst[keep]] = vals[keep]

    def cons(L):
        return torch.randint(0, 4, (L,), generator=g, device=device, dtype=torch.uint8)

    for _ in range(8):
        L = int(150 + rnd((1,)).item() * 1050)
        plant(cons(L), 0.12 / 8, 0.15, 0.28)
    plant(cons(6000), 0.17, 0.03, 0.20, min_len=400)
    plant(cons(300), 0.10, 0.04, 0.16)
    # simple repeats
    k = int(n * 0.02 / 160)
    start = (rnd((k,)).double() * (n - 400)).long()
    ln = 20 + (rnd((k,)) * 280).long()
    unit = 1 + (rnd((k,)) * 6).long().clamp(max=5)
    ub = torch.randint(0, 4, (k, 6), generator=g, device=device, dtype=torch.uint8)
    off = torch.arange(300, device=device).unsqueeze(0)
    keep = off < ln.unsqueeze(1)
    vals = torch.gather(ub, 1, off.expand(k, 300) % unit.unsqueeze(1))
    G[(start.unsqueeze(1) + off)[keep]] = vals[keep]
    # segmental duplications (copied from the current state of the sequence)
    k = max(1, int(n * 0.04 / 22_500))
    for _ in range(k):
        L = int(5000 + rnd((1,)).item() * 35_000)
        a = int(rnd((1,)).item() * (n - L - 1)); b = int(rnd((1,)).item() * (n - L - 1))
        seg = G[a:a + L].clone()
        mut = rnd((L,)) < (0.005 + rnd((1,)).item() * 0.015)
        rb = torch.randint(1, 4, (L,), generator=g, device=device, dtype=torch.uint8)
        G[b:b + L] = torch.where(mut, (seg + rb) % 4, seg)
    # N gaps
    for c in range(N_CHROMS):
        G[c * per:c * per + per // 400] = 4
        G[(c + 1) * per - per // 400:(c + 1) * per] = 4
    k = max(1, n // 1_000_000)
    start = (rnd((k,)).double() * (n - 6000)).long()
    ln = 100 + (rnd((k,)) * 4900).long()
    off = torch.arange(5000, device=device).unsqueeze(0)
    G[(start.unsqueeze(1) + off)[off < ln.unsqueeze(1)]] = 4
    return G, [per] * N_CHROMS


def synth_genome_bacterial(seed, device):
    """E. coli K-12-like stand-in (the real sequence is not available offline): one 4.64 Mbp chromosome of random sequence with what a
    bacterial genome has by way of repeats -- seven copies of a 5 kbp rRNA-operon-like element (0.1-1 % divergence) and forty copies of
    three IS-like elements of 0.8-1.4 kbp (0-2 %); no N."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = 4_640_000
    G = torch.randint(0, 4, (n,), generator=g, device=device, dtype=torch.uint8)

    def plant(L, copies, div_hi):
        cons = torch.randint(0, 4, (L,), generator=g, device=device, dtype=torch.uint8)
        for _ in range(copies):
            a = int(torch.rand(1, generator=g, device=device).item() * (n - L - 1))
            div = torch.rand(1, generator=g, device=device).item() * div_hi
            mut = torch.rand(L, generator=g, device=device) < div
            rb = torch.randint(1, 4, (L,), generator=g, device=device, dtype=torch.uint8)
            G[a:a + L] = torch.where(mut, (cons + rb) % 4, cons)
    plant(5000, 7, 0.01)
    for L, c in ((1400, 14), (1200, 13), (800, 13)):
        plant(L, c, 0.02)
    return G, [n]


def load_fasta_codes(path, device):
    """A real reference: FASTA (plain or gzip) -> codes 0..3 / 4 = anything else, as one uint8 tensor + per-sequence lengths and names.
    Sequences of length 0 are dropped (bowtie2-build keeps them out of the index as well)."""
    import gzip
    import numpy as np
    import torch
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    a = np.frombuffer(raw, dtype=np.uint8)
    lut = np.full(256, 4, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
        lut[c + 32] = i
    nl = np.flatnonzero(a == 10)
    gt = np.flatnonzero(a == ord(">"))
    heads = gt[(gt == 0) | (a[np.maximum(gt, 1) - 1] == 10)]
    names, parts, lens = [], [], []
    for k, h in enumerate(heads):
        e = nl[np.searchsorted(nl, h)] if np.searchsorted(nl, h) < nl.size else a.size
        stop = heads[k + 1] if k + 1 < heads.size else a.size
        body = a[e + 1:stop]
        body = body[(body != 10) & (body != 13) & (body != 32)]
        if body.size == 0:
            continue
        names.append(bytes(a[h + 1:e]).decode("latin-1").rstrip("\r"))
        parts.append(lut[body])
        lens.append(int(body.size))
    if not parts:
        raise SystemExit("bench.py: no sequence in %s" % path)
    G = torch.from_numpy(np.concatenate(parts)).to(device)
    return G, lens, names


def genome_from_index(base, ext, device):
    """The reference sequence back out of an existing index (<base>.3.<ext> = records of {Ns before the stretch, unambiguous bases, first-of-a-sequence},
    <base>.4.<ext> = the unambiguous bases, 2 bits each; reference.cpp:100-171) -- so that reads can be sampled from a genome given only as an index."""
    import numpy as np
    import torch
    W = np.dtype("<u8") if ext == "bt2l" else np.dtype("<u4")
    with open("%s.3.%s" % (base, ext), "rb") as f:
        raw = f.read()
    assert int(np.frombuffer(raw[:4], dtype="<i4")[0]) == 1, "endianness word of the .3 file"
    nrec = int(np.frombuffer(raw[4:4 + W.itemsize], dtype=W)[0])
    rec = np.frombuffer(raw[4 + W.itemsize:4 + W.itemsize + nrec * (2 * W.itemsize + 1)], dtype=np.dtype([("off", W), ("len", W), ("first", "u1")]))
    two = np.fromfile("%s.4.%s" % (base, ext), dtype=np.uint8)
    total = int(rec["off"].sum() + rec["len"].sum())
    out = np.full(total, 4, dtype=np.uint8)
    lens, pos, src, seq_start = [], 0, 0, 0
    for k, r in enumerate(rec):
        if r["first"] and k > 0:
            lens.append(pos - seq_start)
            seq_start = pos
        pos += int(r["off"])
        L = int(r["len"])
        if L:
            idx = np.arange(src, src + L, dtype=np.int64)
            out[pos:pos + L] = (two[idx >> 2] >> ((idx & 3) << 1).astype(np.uint8)) & 3
        pos += L
        src += L
    lens.append(pos - seq_start)
    return torch.from_numpy(out).to(device), lens, None


def build_index_gpu(base, G, chrom_lens, large, device_index, chrom_names=None):
    """The GPU index builder on the in-memory genome -> <base>.{1,2,3,4,rev.1,rev.2}.bt2[l]; returns its stats."""
    import numpy as np
    import torch
    import bowtie2_amd as b
    lut = torch.tensor([ord(c) for c in "ACGTN"], dtype=torch.uint8, device=G.device)
    asc = lut[G.long()].cpu().numpy()
    names, seqs, o = [], [], 0
    for i, L in enumerate(chrom_lens):
        names.append(chrom_names[i] if chrom_names else "chr%d" % (i + 1))
        seqs.append(asc[o:o + L])
        o += L
    t0 = time.time()
    st = b.build_index_mem(names, seqs, base, large=large, device=device_index)
    log("[bench] index built on the GPU in %.1fs (scan %.1f, forward %.1f [%d tied, %d rounds], mirror %.1f [%d rounds], files %.1f)"
        % (time.time() - t0, st.t_parse, st.t_fw, st.tied_fw, st.rounds_fw, st.t_bw, st.rounds_bw, st.t_write))
    return {"seconds": round(time.time() - t0, 2), "scan_s": round(st.t_parse, 2), "forward_s": round(st.t_fw, 2), "mirror_s": round(st.t_bw, 2),
            "files_s": round(st.t_write, 2), "tied_after_first_sort": int(st.tied_fw), "doubling_rounds": int(st.rounds_fw), "text_len": int(st.len)}


# ------------------------------------------------------------------------------------------------- reads ----
def synth_reads_gpu(G, n, length, seed, device):
    """SURVEY.md 8d generator on the GPU: uniform position over windows with at most 2 Ns, 50/50 strand, 1 % substitutions,
    0.1 % insertions + 0.1 % deletions (at most one indel per read here), Phred from {38,38,38,30,20,12}."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ntot = G.numel()
    isn = (G > 3)
    cum = torch.zeros(ntot + 1, dtype=torch.int32, device=device)
    cum[1:] = torch.cumsum(isn.to(torch.int32), 0)
    pos_parts = []
    have = 0
    while have < n:
        cand = (torch.rand(int((n - have) * 1.3) + 1024, generator=g, device=device, dtype=torch.float64) * (ntot - length - 4)).long()
        okm = (cum[cand + length + 2] - cum[cand]) <= 2
        cand = cand[okm]
        pos_parts.append(cand)
        have += cand.numel()
    gpos = torch.cat(pos_parts)[:n]
    del cum
    idx = torch.arange(length, device=device).unsqueeze(0).expand(n, length)
    has_indel = torch.rand(n, generator=g, device=device) < length * 0.002
    is_ins = torch.rand(n, generator=g, device=device) < 0.5
    k = torch.randint(5, length - 5, (n,), generator=g, device=device).unsqueeze(1)
    shift = torch.zeros(n, length, dtype=torch.int64, device=device)
    dele = (has_indel & ~is_ins).unsqueeze(1)
    ins = (has_indel & is_ins).unsqueeze(1)
    shift = torch.where(dele & (idx >= k), torch.ones_like(shift), shift)
    shift = torch.where(ins & (idx > k), -torch.ones_like(shift), shift)
    seq = G[gpos.unsqueeze(1) + idx + shift]
    del shift
    rnd_base = torch.randint(0, 4, (n, length), generator=g, device=device, dtype=torch.uint8)
    seq = torch.where(ins & (idx == k), rnd_base, seq)
    sub = torch.rand(n, length, generator=g, device=device) < 0.01
    seq = torch.where(sub & (seq < 4), (seq + 1 + rnd_base % 3) % 4, seq)
    seq = torch.where(seq > 3, rnd_base, seq)   # genome N -> random base, like a sequencer would call it
    rc = torch.rand(n, generator=g, device=device) < 0.5
    seq = torch.where(rc.unsqueeze(1), (3 - seq).flip(1), seq)
    qtab = torch.tensor([ord(c) for c in "GGG?5-"], dtype=torch.uint8, device=device)
    qual = qtab[torch.randint(0, 6, (n, length), generator=g, device=device)]
    return seq.contiguous(), qual.contiguous()


def reads_from_fastq(path, n, rank, device):
    """--reads-fastq: records [rank*n, (rank+1)*n) of a 4-line FASTQ file whose reads all have one length -> (seq codes, qualities, names padded with NUL).
    (An N in a read becomes code 4, as the drop-in binary's parser makes it.)"""
    import gzip
    import numpy as np
    import torch
    op = gzip.open if open(path, "rb").read(2) == b"\x1f\x8b" else open
    seqs, quals, nms = [], [], []
    with op(path, "rb") as f:
        for k in range((rank + 1) * n):
            rec = [f.readline() for _ in range(4)]
            if not rec[3]:
                break
            if k < rank * n:
                continue
            nms.append(rec[0][1:].rstrip(b"\r\n")); seqs.append(rec[1].rstrip(b"\r\n")); quals.append(rec[3].rstrip(b"\r\n"))
    if not seqs:
        raise SystemExit("bench.py: %s holds no reads for rank %d" % (path, rank))
    L = len(seqs[0])
    if any(len(x) != L for x in seqs) or any(len(x) != L for x in quals):
        raise SystemExit("bench.py: --reads-fastq needs reads of one length (the resident batch is rectangular)")
    lut = np.full(256, 4, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
        lut[c + 32] = i
    sq = lut[np.frombuffer(b"".join(seqs), dtype=np.uint8)].reshape(len(seqs), L)
    ql = np.frombuffer(b"".join(quals), dtype=np.uint8).reshape(len(seqs), L).copy()
    w = max(len(x) for x in nms)
    nm = np.zeros((len(nms), w), dtype=np.uint8)
    for i, x in enumerate(nms):
        nm[i, :len(x)] = np.frombuffer(x, dtype=np.uint8)
    return torch.from_numpy(sq).to(device), torch.from_numpy(ql).to(device), nm


def read_names(n0, n, width=9):
    """Fixed-width names r000000123 as a [n, width+1] uint8 array."""
    import numpy as np
    ids = np.arange(n0, n0 + n, dtype=np.int64)
    out = np.empty((n, width + 1), dtype=np.uint8)
    out[:, 0] = ord("r")
    for d in range(width):
        out[:, width - d] = ord("0") + (ids // (10 ** d)) % 10
    return out


def gen_rand_seeds(seq, qual, names_t):
    """genRandSeed (pat.cpp:45-84) with the default --seed 0, vectorised: what the drop-in binary derives per read."""
    import torch
    n, L = seq.shape
    dev = seq.device
    base = 101 * 59 * 61 * 67 * 71 * 73 * 79 * 83 % (1 << 32)
    acc = torch.full((n,), base, dtype=torch.int64, device=dev)
    for j in range(L):      # XOR-reduction over columns (no xor-reduce primitive); L is 150
        acc ^= (seq[:, j].long() << ((j & 15) << 1)) & 0xffffffff
        acc ^= (qual[:, j].long() << ((j & 3) << 3)) & 0xffffffff
    names_t = names_t.masked_fill((names_t == ord("/")).cumsum(1) > 0, 0)      # the name up to the first '/' (pat.cpp:76-80); a NUL (padding) changes nothing
    for j in range(names_t.shape[1]):
        acc ^= (names_t[:, j].long() << ((j & 3) << 3)) & 0xffffffff
    return (acc & 0xffffffff)


def write_fastq_fixed(path, seq, qual, names, append=False, at=None):
    """FASTQ of equal-length reads, assembled as one 2-D byte array (`at`: written at this byte offset of an existing file instead)."""
    import numpy as np
    s = np.frombuffer(b"ACGTN", dtype=np.uint8)[seq.cpu().numpy()]
    q = qual.cpu().numpy()
    n, L = s.shape
    w = names.shape[1]
    if (names == 0).any():
        # names of several widths (--reads-fastq): NUL-padded rows, written record by record
        with open(path, "ab" if append else "wb") as f:
            for i in range(n):
                f.write(b"@" + names[i].tobytes().rstrip(b"\0") + b"\n" + s[i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n")
        return
    rec = np.empty((n, 1 + w + 1 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1:1 + w] = names; rec[:, 1 + w] = 10
    o = 2 + w
    rec[:, o:o + L] = s; rec[:, o + L] = 10; rec[:, o + L + 1] = ord("+"); rec[:, o + L + 2] = 10
    o2 = o + L + 3
    rec[:, o2:o2 + L] = q; rec[:, o2 + L] = 10
    if at is not None:
        fd = os.open(path, os.O_WRONLY)
        try:
            buf, o = memoryview(rec.tobytes()), 0
            while o < len(buf):
                o += os.pwrite(fd, buf[o:o + (1 << 30)], at + o)
        finally:
            os.close(fd)
        return
    with open(path, "ab" if append else "wb") as f:
        f.write(rec.tobytes())


# BASELINE.json configs[2..4] (+ the round-2 paired line).  `args` is the bowtie2-align command line of the configuration: the timed
# batches take their parameters from it through bt2g_cli_params (the drop-in binary's own option parser), the reference and the product
# binary of the parity check are run with it.
CONFIGS = {
    "se150":    {"args": ["--sensitive"], "paired": False, "readlen": 150, "reads": 2_000_000, "cpu_sample": 1_000_000, "pipeline": 2,
                 "what": "--sensitive (-D 15 -R 2 -N 0 -L 22 -i S,1,1.15), end-to-end"},
    # "pipeline": steps in flight.  A batch of pairs or of long --local reads ends in a tail -- a few pathological reads keep a handful of the
    # 4096 waves busy for hundreds of ms after the rest are done -- which the next batch's waves fill when two batches are in flight (as in
    # the product driver, whose device-stage threads each issue their batch on their own stream).  The headline has no such tail, but since round 6
    # its worker launches 18 of the 20 waves a CU holds (the kernel's throughput saturates at 16), and the lane-per-task FM kernels of the next
    # batch -- random index reads, no LDS -- run in the room that leaves: 330.6 -> 306.9 ms per step with two in flight (profiles/r06x_*).
    "pe-sens":  {"args": ["--sensitive"], "paired": True, "readlen": 150, "reads": 400_000, "cpu_sample": 400_000, "pipeline": 3,
                 "what": "pairs, --sensitive, --fr -I 0 -X 500"},
    "pe-vsens": {"args": ["--very-sensitive", "-X", "500"], "paired": True, "readlen": 150, "reads": 400_000, "cpu_sample": 200_000, "pipeline": 3, "reps": 3,
                 "what": "pairs, --very-sensitive (-D 20 -R 3 -N 0 -L 20 -i S,1,0.50), --fr -I 0 -X 500 (mate rescue)"},
    "local400": {"args": ["--local"], "paired": False, "readlen": 400, "reads": 200_000, "cpu_sample": 100_000, "pipeline": 2,
                 "what": "--local = --sensitive-local (-D 15 -R 2 -N 0 -L 20 -i S,1,0.75, --ma 2, --score-min G,20,8)"},
    # BASELINE.json configs[1]: a bacterial genome behind a small (.bt2: 64-byte sides, 32-bit offsets) index -- the uint32_t instantiations.  Three steps in
    # flight: its launches are short (1 M reads, 130 ms alone) and a launch's ramp and tail are a sixth of it -- 7.28 / 7.29 M reads/s with two in flight,
    # 8.35 / 8.84 / 9.11 M with three (profiles/r06bc_*); the median of three timed regions is reported.
    "ecoli100": {"args": ["--sensitive"], "paired": False, "readlen": 100, "reads": 1_000_000, "cpu_sample": 1_000_000, "genome": "ecoli", "pipeline": 3, "reps": 3,
                 "what": "default preset = --sensitive (-D 15 -R 2 -N 0 -L 22 -i S,1,1.15), end-to-end"},
}


# ---------------------------------------------------------------------------------- CPU baseline + parity ----
def run_timed(cmd):
    t0 = time.perf_counter()
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    return time.perf_counter() - t0, p


def sam_body(path):
    with open(path, "rb") as f:
        return [l for l in f if not l.startswith(b"@PG")]


# ------------------------------------------------------------------------------- --dry-ranks stand-ins ----
class _DryCuda:
    """torch.cuda as far as main() uses it, without a device."""
    class Event:
        def __init__(self, enable_timing=True):
            self.t = 0.0

        def record(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3

    class Stream:
        def __init__(self, device=None):
            pass

        def wait_stream(self, s):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    def current_stream(self):
        return self.Stream()

    def stream(self, s):
        return s

    def synchronize(self):
        pass

    def empty_cache(self):
        pass


def _dry_build_index(base, ext):
    """rank 0 'builds' the index: the files the other ranks look for after the barrier, written the way the builder publishes them (last file last)"""
    time.sleep(0.5)
    for suf in ("1", "2", "3", "4", "rev.1", "rev.2"):
        with open("%s.%s.%s" % (base, suf, ext), "wb") as f:
            f.write(b"dry")
    return {"seconds": 0.5, "dry": True}


class _DryContext:
    """bowtie2_amd.Context as far as main() uses it: records of the right shape, packed sizes of the right order (0.15 KB per read), no alignment."""
    class _Info:
        pass

    def __init__(self, large):
        self.large = large
        self.n = 0

    def load_index(self, base):
        i = self._Info()
        i.side_sz, i.off_size, i.hbm_bytes, i.len = (128, 8, 0, 0) if self.large else (64, 4, 0, 0)
        return i

    def align_batch(self, batch, rp_t, P, readlen):
        import ctypes as C
        import torch
        import bowtie2_amd as b
        n = batch.n
        stride = (C.sizeof(b.ReadResult) + (max(1, P.khits) - 1) * C.sizeof(b.Aln) + 15) & ~15
        res = torch.zeros(n * stride, dtype=torch.uint8)
        res.view(n, stride)[:, 1] = 1          # "aligned"
        self.n = n
        time.sleep(0.01)
        return res, stride

    def results_pack(self, res, n, khits):
        import torch
        offs = torch.arange(n + 1, dtype=torch.int64) * 152
        return torch.zeros(int(offs[n]), dtype=torch.uint8), offs

    def align_timing(self, on_current_stream=False):
        return {"k_exact_sweep": 0.0, "k_one_mm": 0.0, "k_seed_search_exact": 0.0, "k_extend_hits": 0.0, "k_align_reads": 10.0}

    def align_profile(self, reset=False):
        p = [0] * 32
        p[9] = max(1, self.n)
        return p

    def counters(self, reset=False):
        c = self._Info()
        c.rank_queries = c.ftab_lookups = c.sa_lookups = 0
        return c

    def close(self):
        pass


def write_e2e_fastq(G, seq, qual, names, n, args, dev, rank):
    """FASTQ input of the end-to-end leg: the timed batch (whose first reads are the ones compared with the reference) followed by further batches of
    distinct reads from the same generator, other seeds, names numbered on.  Returns the path (a pair of paths for pairs) and the read count."""
    import shutil
    from bowtie2_amd import shard
    work = cache_dir()
    per = 2 * args.readlen + 16
    total = max(n, (args.e2e_reads // n) * n)
    free = shutil.disk_usage(work).free
    while total > n and total * per * 1.1 > free * 0.9:      # the input has to fit (the SAM goes to /dev/null when only the input fits)
        total -= n
    paths = (os.path.join(work, "e2e_1.fq"), os.path.join(work, "e2e_2.fq")) if args.paired else (os.path.join(work, "e2e.fq"),)
    for p in paths:
        if os.path.exists(p):
            os.remove(p)
    t0 = time.time()
    done = 0
    k = 0
    while done < total:
        if k == 0:
            sq, ql, nm = seq, qual, names
        elif args.paired:
            sq, ql = synth_pairs_gpu(G, n // 2, args.readlen, shard.shard_seed(7000 + k, rank), dev)
            nm = read_names(done, n)
        else:
            sq, ql = synth_reads_gpu(G, n, args.readlen, shard.shard_seed(7000 + k, rank), dev)
            nm = read_names(done, n)
        if args.paired:
            # mates interleaved in a batch (named as the timed batch's FASTQ names them)
            for m in (0, 1):
                write_fastq_fixed(paths[m], sq[m::2], ql[m::2], nm[m::2], append=True)
        else:
            write_fastq_fixed(paths[0], sq, ql, nm, append=True)
        done += n
        k += 1
    os.sync()      # (the input's dirty pages reach the disk now, not while the timed run reads it)
    log("[bench] e2e input: %d distinct reads written as FASTQ in %.1fs" % (done, time.time() - t0))
    return {"paths": paths, "reads": done, "batches": k, "first_batch_is_timed_batch": True}


def write_e2e_fastq_ranks(G, n, args, dev, rank, world, dist):
    """The end-to-end input of an N-GPU run: ONE FASTQ file of world x (e2e_reads per rank) distinct reads, which bowtie2_amd.mgpu then shards by byte range
    (the product's N-GPU path).  Records have one size (fixed-width names), so rank r writes its batches straight to their place in the file: rank 0 sizes
    the file, everybody writes in parallel.  Unpaired configurations only (the paired e2e leg stays an N = 1 measurement)."""
    import shutil
    import torch
    from bowtie2_amd import shard
    work = cache_dir()
    rec = 1 + 10 + 1 + args.readlen + 3 + args.readlen + 1
    total = max(n, (args.e2e_reads // n) * n)
    free = shutil.disk_usage(work).free
    while total > n and world * total * rec * 1.1 > free * 0.9:
        total -= n
    if dist is not None:       # every rank must agree on the size of a rank's section
        t = torch.tensor([total], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        total = int(t.item())
    path = os.path.join(work, "e2e_n%d.fq" % world)
    if rank == 0:
        with open(path, "wb") as f:
            f.truncate(world * total * rec)
    if dist is not None:
        dist.barrier()
    t0 = time.time()
    done, k = 0, 0
    while done < total:
        sq, ql = synth_reads_gpu(G, n, args.readlen, shard.shard_seed(7000 + k, rank) + 100 * rank, dev)
        nm = read_names(rank * total + done, n)
        write_fastq_fixed(path, sq, ql, nm, at=(rank * total + done) * rec)
        done += n
        k += 1
    if dist is not None:
        dist.barrier()
    if rank == 0:
        os.sync()
        log("[bench] e2e input: %d ranks x %d distinct reads written into one FASTQ file in %.1fs" % (world, total, time.time() - t0))
    return {"paths": (path,), "reads": world * total, "reads_per_rank": total, "batches": k, "first_batch_is_timed_batch": False}


def _clean_launcher_env():
    """the environment for a job launched from inside a torch.distributed.run worker: without the outer job's rendezvous"""
    drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT",
            "BT2_BENCH_CHILD", "OMP_NUM_THREADS")
    return {k: v for k, v in os.environ.items() if k not in drop and not k.startswith("TORCHELASTIC_") and not k.startswith("TORCH_NCCL_ASYNC")}


def _free_port():
    import socket
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    return port


def e2e_leg_ranks(pend, world, resident_rate):
    """The N-GPU end-to-end leg: `python -m torch.distributed.run ... -m bowtie2_amd.mgpu -- <options> -U e2e.fq -S e2e.sam` -- the product's own N-GPU
    driver: one drop-in executable per GPU on its byte range of the one input file, pieces concatenated by rank 0, counters all-reduced -- after every process of
    the resident-batch measurement has exited.  reads/s after the load = all reads / the slowest rank's search wall time (each rank's executable prints it with -t)."""
    import shutil
    work, fq = pend["work"], pend["fq"]
    path = fq["paths"][0]
    out = os.path.join(work, "e2e_n%d.sam" % world)
    in_bytes = os.path.getsize(path)
    # the other ranks' pieces are files next to the merged output until rank 0 has appended them (bowtie2_amd.mgpu): room for both
    to_file = shutil.disk_usage(work).free > in_bytes * 1.3 * 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           "-m", "bowtie2_amd.mgpu"] + list(pend.get("mgpu_args", [])) + ["--"] + list(pend["preset"]) + ["-t", "-x", pend["base"], "-U", path, "-S", out if to_file else "/dev/null"]
    env = _clean_launcher_env()
    env["TMPDIR"] = work       # (mgpu keeps the pieces in a temporary directory: on the disk the bench's cache lives on)
    time.sleep(float(os.environ.get("BT2_BENCH_E2E_SETTLE_S", "8")))
    runs = []
    for _ in range(int(os.environ.get("BT2_BENCH_E2E_RUNS_N", "1"))):
        if os.path.exists(out):
            os.remove(out)
        os.sync()
        t0 = time.perf_counter()
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT)
        t = time.perf_counter() - t0
        per = [(float(a), float(b_), int(c)) for a, b_, c in re.findall(r"index load ([\d.]+) s; search ([\d.]+) s wall, (\d+) reads", p.stderr)]
        runs.append((t, p, per))
    t, p, per = sorted(runs, key=lambda r: r[0])[len(runs) // 2]
    e = {"n_gpus": world, "reads": fq["reads"], "distinct_reads": True, "input": "one FASTQ file, %d bytes, sharded by byte range (bowtie2_amd.mgpu)" % in_bytes,
         "output": "one SAM file (rank 0's executable writes into it, the other ranks' pieces are appended with sendfile)" if to_file else "/dev/null",
         "command": " ".join(os.path.basename(c) if os.sep in c else c for c in cmd), "returncode": p.returncode, "wall_s_job": round(t, 2),
         "host_threads_per_rank": max(1, pend["threads"] // world), "protocol": "%d run(s) of the command; search time = the slowest rank's" % len(runs)}
    if p.returncode != 0 or len(per) != world:
        e["error"] = p.stderr[-600:]
    else:
        nreads = sum(x[2] for x in per)
        slow = max(x[1] for x in per)
        e.update({"reads_aligned_by_the_ranks": nreads, "search_s_per_rank": [x[1] for x in per], "index_load_s_per_rank": [x[0] for x in per],
                  "reads_per_s_after_load": nreads / slow, "frac_of_resident": (nreads / slow) / resident_rate if resident_rate else None,
                  "reads_per_s_whole_job": nreads / t})
        if to_file:
            e["sam_bytes"] = os.path.getsize(out)
            # the merged file holds every read once, in input order: count the records, look at the first and the last name
            nrec, first, last = 0, None, None
            with open(out, "rb") as f:
                for l in f:
                    if l[:1] == b"@":
                        continue
                    nrec += 1
                    if first is None:
                        first = l.split(b"\t", 1)[0]
                    last = l
            e["sam_records"] = nrec
            e["sam_complete_and_in_input_order"] = bool(nrec == fq["reads"] and first == b"r%09d" % 0 and last.split(b"\t", 1)[0] == b"r%09d" % (fq["reads"] - 1))
    for f_ in (out,) + (() if os.environ.get("BT2_BENCH_KEEP_E2E") else (path,)):
        try:
            os.remove(f_)
        except OSError:
            pass
    return e


def dry_engine(argv):
    """--dry-ranks stand-in for the drop-in executable as bowtie2_amd.mgpu drives it (no device here): takes its byte range of the reads file, writes one
    SAM-like line per record, the shard index and the -t line.  Nothing in it aligns anything; it lets the CPU suite run the N-GPU e2e leg's plumbing."""
    a = {"--shard-bytes": None, "--shard-first-read": "0", "-S": None, "--shard-index": None, "-U": None}
    i = 0
    while i < len(argv):
        if argv[i] in a and i + 1 < len(argv):
            a[argv[i]] = argv[i + 1]
            i += 1
        i += 1
    t0 = time.perf_counter()
    lo, hi = (int(x) for x in a["--shard-bytes"].split(",")[0].split(":"))
    with open(a["-U"], "rb") as f:
        f.seek(lo)
        data = f.read(hi - lo)
    lines = data.split(b"\n")
    names = [l[1:] for l in lines[0::4] if l]
    with open(a["-S"], "wb") as f:
        if "--no-hd" not in argv:
            f.write(b"@HD\tVN:1.5\tSO:unsorted\n")
        for nm in names:
            f.write(nm + b"\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*\n")
    with open(a["--shard-index"], "w") as f:
        f.write("S %d %d 0 0\nF 0\nR %d\n" % (len(names), len(names), hi - lo))
    dt = max(1e-6, time.perf_counter() - t0)
    sys.stderr.write("[bt2g] index load 0.000 s; search %.3f s wall, %d reads -> %d reads/s after the load\n" % (dt, len(names), len(names) / dt))
    return 0


def e2e_leg(base, large, fq, preset, threads, resident_rate, work, par):
    """The drop-in binary, its own process, FASTQ file -> SAM file on the e2e input; -t prints the index load, the wall time of the search after it
    and the rate.  The SAM records of the reads that were compared with the reference (the head of the file) are compared with the reference's again."""
    import shutil
    sfx = "l" if large else "s"
    exe = os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-align-%s" % sfx)
    paths = fq["paths"]
    out = os.path.join(work, "e2e.sam")
    in_bytes = sum(os.path.getsize(p) for p in paths)
    to_file = shutil.disk_usage(work).free > in_bytes * 1.3
    rd = ["-U", paths[0]] if len(paths) == 1 else ["-1", paths[0], "-2", paths[1]]
    cmd = [exe] + list(preset) + ["-t", "-p", str(threads), "-x", base] + rd + ["-S", out if to_file else "/dev/null"]
    # The process that measured the resident batches has just exited and the driver is still wiping the ~100 GB of HBM it held; the wipe runs
    # on the copy engines this binary's uploads and downloads need (session r05h: the first batch came back after 3.8 s instead of 0.25 s,
    # with the time booked under "download").  A user's GPU is not being wiped: give it a moment, then take the median of three runs.
    time.sleep(float(os.environ.get("BT2_BENCH_E2E_SETTLE_S", "8")))
    runs = []
    for _ in range(int(os.environ.get("BT2_BENCH_E2E_RUNS", "3"))):
        # every run writes a fresh file onto a quiet disk: truncating the previous run's 10 GB output, and its dirty pages still on their way
        # to the disk, cost the next run a second at either end (session r05z: 5.0 M reads/s for the first run, 3.5 M for the two behind it)
        if os.path.exists(out):
            os.remove(out)
        os.sync()
        t, p = run_timed(cmd)
        mm = re.search(r"index load ([\d.]+) s; search ([\d.]+) s wall, (\d+) reads -> (\d+) reads/s after the load", p.stderr)
        runs.append((int(mm.group(4)) if mm else -1, t, p, mm))
    e = {"reads": fq["reads"], "distinct_reads": True, "input": "FASTQ file%s, %d bytes" % ("s (-1/-2)" if len(paths) == 2 else "", in_bytes),
         "output": "SAM file" if to_file else "/dev/null (no room for the SAM file next to the input)",
         "command": " ".join(os.path.basename(c) if os.sep in c else c for c in cmd),
         "runs_reads_per_s_after_load": [r[0] for r in runs], "protocol": "median of %d runs of the same command, in order" % len(runs)}
    _, t, p, m = sorted(runs, key=lambda r: r[0])[len(runs) // 2]
    e["wall_s_process"] = round(t, 2)
    e["returncode"] = p.returncode
    flagged = re.search(r"Error: (\d+) read\(s\) exceeded a limit of this build", p.stderr)
    if flagged:
        # (flagged, never approximated: the binary exits 1 and says which reads)
        e["reads_over_a_capacity_limit"] = int(flagged.group(1))
        e["capacity_warnings"] = [l for l in p.stderr.splitlines() if l.startswith("Warning")][:4]
    if not m or (p.returncode != 0 and not flagged):
        e["error"] = p.stderr[-400:]
    else:
        e.update({"index_load_s": float(m.group(1)), "search_s": float(m.group(2)), "reads_per_s_after_load": int(m.group(4)),
                  "frac_of_resident": int(m.group(4)) / resident_rate, "reads_per_s_whole_process": fq["reads"] / t,
                  "stages": [l.strip() for l in p.stderr.splitlines() if l.startswith("[bt2g]")]})
        if to_file:
            e["sam_bytes"] = os.path.getsize(out)
            ref = os.path.join(work, "sample.ref.sam")
            if par and par.get("parity_identical") is not None and os.path.exists(ref):
                # the head of the e2e input is the parity sample: same reads, same names -> the same SAM records, whatever batch they travelled in
                a = [l for l in sam_body(ref) if not l.startswith(b"@")]
                nd, k = 0, 0
                with open(out, "rb") as f:
                    for l in f:
                        if l.startswith(b"@"):
                            continue
                        if k >= len(a):
                            break
                        nd += l != a[k]
                        k += 1
                e["head_records_compared_with_reference"] = k
                e["head_records_differing"] = nd + (len(a) - k)
    for f_ in (out, os.path.join(work, "sample.ref.sam")) + (() if os.environ.get("BT2_BENCH_KEEP_E2E") else tuple(paths)):
        try:
            os.remove(f_)
        except OSError:
            pass
    return e


def cpu_baseline_and_parity(base, large, fq_all, fq_tiny, n_sample, n_tiny, threads, preset, work, readlen=150, parity_only=False, keep_ref_sam=False):
    # fq_all / fq_tiny: one FASTQ path (unpaired) or a pair of paths (mate 1, mate 2)
    def rd_args(fq):
        return ["-U", fq] if isinstance(fq, str) else ["-1", fq[0], "-2", fq[1]]
    paired = not isinstance(fq_all, str)
    """Reference bowtie2-align (AVX2 build when present) on the sample: reads/s from wall clock minus the wall clock of
    an index-load-dominated run (n_tiny reads), median of 3.  The first pass writes SAM; the product binary aligns the
    same FASTQ on the GPU and the two SAM files are compared byte for byte (minus @PG)."""
    sfx = "l" if large else "s"
    exe = os.path.join(ROOT, "oracle", "_ref", "bowtie2-align-%s-v256" % sfx)
    simd = "AVX2 (-march=x86-64-v3 -DSSE_AVX2)"
    if not os.path.exists(exe):
        exe = os.path.join(ROOT, "oracle", "_ref", "bowtie2-align-%s" % sfx)
        simd = "SSE2"
    if not os.path.exists(exe):
        return None, None
    common = list(preset) + ["-p", str(threads), "--reorder", "-x", base]
    preset = " ".join(preset)
    ref_sam = os.path.join(work, "sample.ref.sam")
    t_load = []
    for _ in range(2):
        t, p = run_timed([exe] + common + rd_args(fq_tiny) + ["-S", "/dev/null"])
        if p.returncode != 0:
            log("[bench] cpu baseline failed:", p.stderr[-500:]); return None, None
        t_load.append(t)
    t_full = []
    al = None
    for k in range(1 if parity_only else 3):
        t, p = run_timed([exe] + common + rd_args(fq_all) + ["-S", ref_sam if k == 0 else "/dev/null"])
        if p.returncode != 0:
            log("[bench] cpu baseline failed:", p.stderr[-500:]); return None, None
        t_full.append(t)
        al = re.search(r"([\d.]+)% overall alignment rate", p.stderr)
    t_full.sort(); tl = min(t_load)
    search = t_full[len(t_full) // 2] - tl
    cb = {"value": (n_sample - n_tiny) / search, "unit": "reads/s", "cores": threads, "kind": "reference",
          "sample": ("%d of the same synthetic %d bp reads" + (" (as pairs, -1/-2)" if paired else "") + ", unmodified bowtie2-align-%s v2.5.5 (oracle/_ref, -O3, %s) %s -p %d --reorder; "
                     "time = wall clock (median of %d: %s s) minus the wall clock of a %d-read run that is all index load (%.2f s); CPU: %s; "
                     "overall alignment rate %s%%") % (n_sample, readlen, sfx, simd, preset, threads, len(t_full), "/".join("%.2f" % x for x in t_full), n_tiny, tl, cpu_model(),
                                                      al.group(1) if al else "?")}
    # parity: the product binary (GPU) on the same FASTQ
    ours = os.path.join(ROOT, "bowtie2_amd", "bin", "bowtie2-align-%s" % sfx)
    our_sam = os.path.join(work, "sample.gpu.sam")
    t, p = run_timed([ours] + common + rd_args(fq_all) + ["-S", our_sam])
    par = {"parity_checked_reads": 0, "parity_identical": False}
    if p.returncode != 0:
        log("[bench] product binary failed on the parity sample:", p.stderr[-800:])
        par["parity_error"] = p.stderr[-300:]
    else:
        a, b_ = sam_body(ref_sam), sam_body(our_sam)
        ndiff = sum(1 for x, y in zip(a, b_) if x != y) + abs(len(a) - len(b_))
        if ndiff:
            # keep the evidence: differing lines (reference first) and the reads behind them, next to the bench output
            dd = os.path.join(ROOT, "gpurun_out", "parity_diff")
            os.makedirs(dd, exist_ok=True)
            names = set()
            with open(os.path.join(dd, "diff.txt"), "wb") as f:
                for x, y in zip(a, b_):
                    if x != y and len(names) < 200:
                        f.write(b"REF " + x + b"GPU " + y)
                        names.add(x.split(b"\t", 1)[0]); names.add(y.split(b"\t", 1)[0])
            with open(fq_all if not paired else fq_all[0], "rb") as f, open(os.path.join(dd, "reads.fq"), "wb") as g:
                while True:
                    rec = [f.readline() for _ in range(4)]
                    if not rec[0]:
                        break
                    if rec[0][1:].strip() in names:
                        g.write(b"".join(rec))
        par = {"parity_checked_reads": n_sample, "parity_identical": ndiff == 0, "parity_differing_sam_lines": ndiff,
               "parity_product_binary_wall_s": round(t, 2)}
        aligned_ref = sum(1 for l in a if not l.startswith(b"@") and not (int(l.split(b"\t", 2)[1]) & 4))
        par["parity_sample_aligned_reads"] = aligned_ref
    for f in (our_sam,) if keep_ref_sam else (ref_sam, our_sam):
        try:
            os.remove(f)
        except OSError:
            pass
    return cb, par


def pmc_files(config):
    """committed PMC summaries of this configuration, oldest first: profiles/*pmc_traffic.json is the headline's (se150), the others carry
    the configuration's name (profiles/*pmc_traffic_<config>.json) -- per-read traffic and instruction counts do not transfer between them"""
    import glob
    pat = "*pmc_traffic.json" if config == "se150" else "*pmc_traffic_%s.json" % config
    return sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))


def lib_sha256():
    """SHA-256 of the library this run loads (bowtie2_amd/libbt2g.so): what a committed PMC summary must have been measured on to be quoted."""
    import hashlib
    h = hashlib.sha256()
    try:
        with open(os.path.join(ROOT, "bowtie2_amd", "libbt2g.so"), "rb") as f:
            for blk in iter(lambda: f.read(1 << 22), b""):
                h.update(blk)
    except OSError:
        return None
    return h.hexdigest()


def pmc_traffic(kernel, reads_per_launch, config="se150"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE + WRITE_SIZE, KiB), scaled from the
    reads-per-launch of that profile run to this run's -- ONLY when that file was measured on the library this run loads (its `lib_sha256`);
    otherwise None.  The default N = 1 run replaces it with a live measurement (live_pmc_traffic: rocprofv3 needs to sit around the process)."""
    files = pmc_files(config)
    if not files:
        return None, "no committed PMC summary for %s" % config
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        have, want = d.get("lib_sha256"), lib_sha256()
        if not have or have != want:
            return None, "profiles/%s was measured on another build of libbt2g.so (sha256 %s, loaded %s): not quoted" % (os.path.basename(files[-1]), (have or "unrecorded")[:16], (want or "?")[:16])
        k = d["kernels"][kernel]
        # MI355X_MICROARCH.md, HBM section: rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests as 64 bytes
        # (calibrated on wide coalesced reads) -> doubled; WRITE_SIZE is taken as reported (uncalibrated there)
        per_read = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0 / (d["reads_per_launch"] * d["launches"])
        return int(per_read * reads_per_launch), "profiles/%s, measured on this libbt2g.so (sha256 %s): (2 x FETCH_SIZE + WRITE_SIZE) per read x %d reads; FETCH_SIZE doubled as the guide prescribes for gfx950" % (os.path.basename(files[-1]), want[:16], reads_per_launch)
    except Exception as e:
        return None, "unreadable PMC summary: %s" % e


def live_pmc_traffic(config, kernel, argv_extra, timeout_s=300):
    """roofline.traffic as a LIVE number: this script's own measuring process once more under `rocprofv3 --kernel-trace --pmc X`,
    one counter per pass as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass), 1 warm-up + 1 timed + 1 profiled launch of
    the same resident batch.  Returns {"fetch_kib", "write_kib", "dispatches", "traffic_bytes_per_launch"} or {"error": ...}."""
    import csv
    import glob
    import shutil
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not rp:
        return {"error": "rocprofv3 not found"}
    out = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pmc_", dir=cache_dir())
        cmd = [rp, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--config", config, "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--e2e-reads", "0"] + list(argv_extra)
        env = dict(os.environ, BT2_BENCH_CHILD="1", TMPDIR="/tmp")
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd="/tmp", timeout=timeout_s)
            tot, ids = 0.0, set()
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if kernel + "<" in r["Kernel_Name"] or r["Kernel_Name"].split("(")[0].endswith(kernel):
                            if r["Counter_Name"] == ctr:
                                tot += float(r["Counter_Value"]); ids.add(r["Dispatch_Id"])
            if not ids:
                return {"error": "no %s rows for %s (rc %d): %s" % (ctr, kernel, p.returncode, p.stderr[-300:])}
            out[ctr] = tot / len(ids)
            out["dispatches"] = len(ids)
        except subprocess.TimeoutExpired:
            return {"error": "rocprofv3 --pmc %s pass timed out after %d s" % (ctr, timeout_s)}
        except Exception as e:
            return {"error": "rocprofv3 --pmc %s pass: %s" % (ctr, e)}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"fetch_kib_per_launch": out["FETCH_SIZE"], "write_kib_per_launch": out["WRITE_SIZE"], "dispatches": out["dispatches"],
            "traffic_bytes_per_launch": int((2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0)}


def pmc_issue(kernel, kern_ms, reads_per_launch, n_cu, config="se150"):
    """What actually bounds the worker kernel: instructions issued per read (committed --pmc pass) against the SIMD-cycles the
    launch had (4 SIMDs per CU, a wave64 vector instruction occupies its SIMD for 4 cycles; 2.4 GHz engine clock)."""
    files = pmc_files(config)
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        k = d["kernels"][kernel]
        per = float(d["reads_per_launch"] * d["launches"])
        valu, salu, lds = k["SQ_INSTS_VALU"] / per, k["SQ_INSTS_SALU"] / per, k["SQ_INSTS_LDS"] / per
        vmem = (k["SQ_INSTS_VMEM_RD"] + k["SQ_INSTS_VMEM_WR"]) / per
        simd_cycles_per_read = kern_ms * 1e-3 * 2.4e9 * n_cu * 4 / reads_per_launch
        return {"source": "profiles/" + os.path.basename(files[-1]), "measured_on_loaded_library": bool(d.get("lib_sha256")) and d.get("lib_sha256") == lib_sha256(), "valu_per_read": round(valu), "salu_per_read": round(salu), "lds_per_read": round(lds),
                "vmem_per_read": round(vmem), "simd_cycles_per_read": round(simd_cycles_per_read), "valu_busy_frac": round(valu * 4 / simd_cycles_per_read, 3),
                "note": "k_align_reads is one serial instruction stream per read (one wavefront each, 4 or 5 per SIMD): not HBM-bandwidth work; round 6 measured 63 % of "
                        "its wave cycles parked at s_waitcnt and a throughput that saturates at 16 waves per CU -- it is bound by the latency of ~800 dependent L2 misses "
                        "per read, not by the issue rate (19 % fewer vector instructions changed nothing, DESIGN section 6); the hbm fraction above is reported because the contract asks for it"}
    except Exception:
        return None


def synth_pairs_gpu(G, npairs, length, seed, device):
    """Pairs for --paired: fragment length N(300,30) clipped to [length+1, 450], mate 1 = fragment start (forward), mate 2 = reverse
    complement of the fragment end, 1 % substitutions, half of the pairs with the roles of the mates swapped.  Returns [2*npairs, length]."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ntot = G.numel()
    frag = (300.0 + 30.0 * torch.randn(npairs, generator=g, device=device)).long().clamp(length + 1, 450)
    gpos = (torch.rand(npairs, generator=g, device=device, dtype=torch.float64) * (ntot - 460)).long()
    idx = torch.arange(length, device=device).unsqueeze(0).expand(npairs, length)
    m1 = G[gpos.unsqueeze(1) + idx]
    m2 = (3 - G[(gpos + frag - length).unsqueeze(1) + idx].clamp(max=3)).flip(1)
    both = torch.stack([m1, m2], dim=1)                               # [npairs, 2, length]
    swap = torch.rand(npairs, generator=g, device=device) < 0.5
    both = torch.where(swap.view(-1, 1, 1), both.flip(1), both)
    seq = both.reshape(2 * npairs, length).to(torch.uint8)
    rnd_base = torch.randint(0, 4, seq.shape, generator=g, device=device, dtype=torch.uint8)
    sub = torch.rand(seq.shape, generator=g, device=device) < 0.01
    seq = torch.where(sub & (seq < 4), (seq + 1 + rnd_base % 3) % 4, seq)
    seq = torch.where(seq > 3, rnd_base, seq)
    qtab = torch.tensor([ord(c) for c in "GGG?5-"], dtype=torch.uint8, device=device)
    qual = qtab[torch.randint(0, 6, seq.shape, generator=g, device=device)]
    return seq.contiguous(), qual.contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="se150",
                    help="se150 = the headline (BASELINE.json configs[2]); ecoli100 = configs[1]; pe-vsens = configs[3]; local400 = configs[4]")
    ap.add_argument("--genome-mbp", type=int, default=int(os.environ.get("BT2_BENCH_MBP", "3100")), help="3100 = hg38 scale")
    ap.add_argument("--small-index", action="store_true", help="build a .bt2 (32-bit) index instead of the headline .bt2l")
    ap.add_argument("--reads", type=int, default=int(os.environ.get("BT2_BENCH_READS", "0")), help="reads per GPU per step (0: the config's default)")
    ap.add_argument("--readlen", type=int, default=0, help="0: the config's read length")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("BT2_BENCH_CPU_SAMPLE", "0")))
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the reference run (no cpu_baseline, no SAM parity)")
    ap.add_argument("--parity-only", action="store_true", help="run the reference once, for the SAM comparison only (cpu_baseline then comes from one run)")
    ap.add_argument("--paired", action="store_true", help="same as --config pe-sens: the paired kernel at --sensitive (round-2 line)")
    ap.add_argument("--e2e-reads", type=int, default=-1, help="distinct reads the product binary aligns FASTQ file -> SAM file after the timed steps (N = 1 only); "
                    "-1: 12 batches of --reads (24 M for the headline) in the default run, none with --parity-only / --no-cpu-baseline; 0: none")
    ap.add_argument("--dry-ranks", action="store_true", help="no GPU: every rank runs this script's N-GPU plumbing -- rendezvous (gloo), rank 0 filling the "
                    "index cache while the others wait, per-rank read shards, the step loop with its per-step gather of packed records to rank 0, "
                    "barrier + max-over-ranks timing, the summed counters, rank 0's JSON line -- around a stand-in for the device (tests/test_bench_dry_ranks.py)")
    ap.add_argument("--pipeline", type=int, default=0, help="steps in flight (on that many streams; the context keeps a working set per stream): 0 = the config's default")
    ap.add_argument("--reps", type=int, default=0, help="repetitions of the timed region (K steps each, barrier + synchronize around every one); value = the median, the spread is "
                    "reported; 0: the config's default (3 for pe-vsens, whose step time is tail-dominated; 1 otherwise)")
    ap.add_argument("--extra-configs", default=None, help="comma-separated configurations run after the headline, each in a process of its own, and reported under `configs` of the "
                    "one JSON line (default at N = 1 for the full headline run: ecoli100,pe-vsens,local400 = BASELINE.json configs[1], [3], [4]; '' = none)")
    ap.add_argument("--extra-steps", type=int, default=10, help="timed steps of each extra configuration")
    ap.add_argument("--no-pmc-pass", action="store_true", help="do not spawn the rocprofv3 --pmc passes that make roofline.traffic a live number")
    ap.add_argument("--genome-fasta", default=None, help="a real reference (FASTA, plain or gzip) instead of the synthetic genome: indexed by the GPU builder, reads sampled from it by the "
                    "same generator (env BT2_BENCH_HG38 for the hg38 configurations, BT2_BENCH_ECOLI for ecoli100)")
    ap.add_argument("--index-base", default=None, help="an existing bowtie2 index (<base>.1.bt2[l] ...): used as it is; the sequence reads are sampled from is recovered from its .3/.4 files")
    ap.add_argument("--check-determinism", action="store_true", help="diagnostic: align the resident batch twice more after the timed region and compare the result records byte for byte (config.determinism)")
    ap.add_argument("--reads-fastq", default=None, help="reads of one length from this FASTQ file (names, qualities and all) instead of the generator")
    args = ap.parse_args()
    if args.paired:
        args.config = "pe-sens"
    cfg = CONFIGS[args.config]
    args.paired = cfg["paired"]
    if not args.reads:
        args.reads = cfg["reads"]
    if not args.readlen:
        args.readlen = cfg["readlen"]
    if not args.cpu_sample:
        args.cpu_sample = cfg["cpu_sample"]
    if not args.pipeline:
        args.pipeline = cfg.get("pipeline", 1)
    if not args.reps:
        args.reps = cfg.get("reps", 1)
    if not args.genome_fasta and not args.index_base:
        args.genome_fasta = os.environ.get("BT2_BENCH_ECOLI" if cfg.get("genome") == "ecoli" else "BT2_BENCH_HG38") or None

    import numpy as np
    import torch
    import bowtie2_amd as b

    from bowtie2_amd import shard
    rank, local_rank, world = shard.env_rank()
    dry = args.dry_ranks
    if dry:
        # the plumbing of the N-GPU run without the device: nothing measured here is a result (data says so), everything exchanged is
        dev = torch.device("cpu")
        dist = shard.init("gloo")
        cuda = _DryCuda()
        args.genome_mbp = min(args.genome_mbp, 1)
        args.no_cpu_baseline = True
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        dist = shard.init("nccl", dev)      # RCCL; None when WORLD_SIZE == 1
        cuda = torch.cuda

    threads = nproc()
    bacterial = cfg.get("genome") == "ecoli"
    large = not (args.small_index or bacterial)
    real, chrom_names = None, None      # a real reference when the box has one (--genome-fasta / --index-base / BT2_BENCH_HG38 / BT2_BENCH_ECOLI)
    if args.index_base:
        if os.path.exists(args.index_base + ".1.bt2l"):
            large = True
        elif os.path.exists(args.index_base + ".1.bt2"):
            large = False
        else:
            raise SystemExit("bench.py: no index at %s" % args.index_base)
    ext = "bt2l" if large else "bt2"
    t0 = time.time()
    if args.index_base:
        base = args.index_base
        G, chrom_lens, chrom_names = genome_from_index(base, ext, dev)
        real = "index %s (%d sequences, %.1f Mbp incl. N)" % (os.path.basename(base), len(chrom_lens), G.numel() / 1e6)
        args.genome_mbp = G.numel() / 1e6
    elif args.genome_fasta:
        G, chrom_lens, chrom_names = load_fasta_codes(args.genome_fasta, dev)
        st_ = os.stat(args.genome_fasta)
        tag = re.sub(r"[^A-Za-z0-9]+", "_", os.path.basename(args.genome_fasta))[:40]
        base = os.path.join(cache_dir(), "real_%s_%d_%s" % (tag, st_.st_size, ext))
        real = "%s (%d sequences, %.1f Mbp incl. N)" % (os.path.basename(args.genome_fasta), len(chrom_lens), G.numel() / 1e6)
        args.genome_mbp = G.numel() / 1e6
    else:
        base = os.path.join(cache_dir(), "ecolilike_s3_%s" % ext if bacterial else "hg38like_%dmbp_s2_%s" % (args.genome_mbp, ext))
        # ---- workload: genome (every rank, same seed) + index (rank 0 builds it on its GPU, the others wait) ----
        if bacterial:
            G, chrom_lens = synth_genome_bacterial(3, dev)
            args.genome_mbp = G.numel() / 1e6
        else:
            G, chrom_lens = synth_genome_gpu(args.genome_mbp, 2, dev)
    if dry:
        base += "_dry"
    cuda.synchronize()
    log("[bench] genome: %.4g Mbp %s in %.1fs" % (args.genome_mbp, "read from " + real if real else "generated", time.time() - t0))
    build_info = None
    if rank == 0 and not os.path.exists(base + ".rev.2." + ext):
        cuda.empty_cache()       # the builder allocates ~30 bytes per base with hipMalloc, next to torch's caching allocator
        build_info = _dry_build_index(base, ext) if dry else build_index_gpu(base, G, chrom_lens, large, local_rank, chrom_names)
    if dist is not None:
        dist.barrier()
    if not os.path.exists(base + ".rev.2." + ext):
        raise SystemExit("rank %d: the index rank 0 built is not visible at %s" % (rank, base))

    ctx = _DryContext(large) if dry else b.Context(local_rank)
    t0 = time.time()
    info = ctx.load_index(base)
    index_load_s = time.time() - t0
    log("[bench] index loaded into HBM in %.2fs (%.2f GB)" % (index_load_s, info.hbm_bytes / 1e9))
    # per-rank shard of reads (weak scaling: fixed reads per GPU)
    n = args.reads
    if args.reads_fastq:
        seq, qual, names = reads_from_fastq(args.reads_fastq, n, rank, dev)
        n = seq.shape[0]
        args.readlen = seq.shape[1]
    elif args.paired:
        # mates interleaved: read 2i = forward mate at the fragment start, read 2i+1 = reverse-complemented mate at its end
        seq, qual = synth_pairs_gpu(G, n // 2, args.readlen, shard.shard_seed(2000, rank), dev)
        n = seq.shape[0]
    else:
        seq, qual = synth_reads_gpu(G, n, args.readlen, shard.shard_seed(1000, rank), dev)
    if not args.reads_fastq:
        names = read_names(rank * n, n)
    # ---- the end-to-end leg's input (N = 1): the timed batch followed by further batches of DISTINCT reads of the same generator, as FASTQ ----
    if args.e2e_reads < 0:
        args.e2e_reads = 0 if (args.no_cpu_baseline or args.parity_only) else 12 * n
    e2e_fq = None
    if args.e2e_reads > 0 and not args.reads_fastq:
        if world == 1:
            e2e_fq = write_e2e_fastq(G, seq, qual, names, n, args, dev, rank)
        elif not args.paired:
            # N GPUs: one input file of world x e2e_reads distinct reads for the product's N-GPU driver (bowtie2_amd.mgpu, byte-range sharding)
            e2e_fq = write_e2e_fastq_ranks(G, n, args, dev, rank, world, dist)
    del G
    cuda.empty_cache()
    names_t = torch.from_numpy(names).to(dev)
    off = (torch.arange(n + 1, device=dev, dtype=torch.int64) * args.readlen)
    batch = b.ReadBatch(seq.view(-1), qual.view(-1), off, n)

    # batch and per-read parameters exactly as the drop-in binary derives them from this command line (bt2g_cli_params = its own parser):
    # --sensitive at 150 bp gives -L 22, interval 1 + 1.15*sqrt(150) = 15, minsc = (long)(-0.6 + -0.6*len), nceil = 0.15*len;
    # the per-read seed is genRandSeed(name, seq, qual) -- so the timed batch is the verified configuration
    import ctypes as C
    cli = list(cfg["args"]) + (["-1", "a", "-2", "b"] if args.paired else ["-U", "a"])
    P, rp1 = b.cli_params(cli, args.readlen, large_index=info.off_size == 8, both_mates_pass=args.paired)
    P.max_seeds = 1 + max(0, args.readlen - rp1.seedlen) // rp1.interval      # every read has this many seed positions per strand: with the bound given, bt2g_align_batch does not synchronise
    rp = np.zeros(n, dtype=[("minsc", "<i4"), ("interval", "<i4"), ("nceil", "<i4"), ("seedlen", "<i4"), ("seed", "<u4"), ("filt", "<u4")])
    rp["minsc"] = rp1.minsc; rp["interval"] = rp1.interval; rp["nceil"] = rp1.nceil; rp["seedlen"] = rp1.seedlen; rp["filt"] = rp1.filt
    rp["seed"] = gen_rand_seeds(seq, qual, names_t).cpu().numpy().astype(np.uint32)
    rp_t = torch.from_numpy(rp.view(np.uint8).copy()).to(dev)

    ev = lambda: cuda.Event(enable_timing=True)
    stage_events = []
    last = {}
    kern_times = []
    depth = max(1, args.pipeline)      # (N-GPU path too, round 6: a batch's records are packed and gathered over RCCL after the NEXT batch has been issued)
    if args.warmup < depth:
        # every stream's working set is created (its arena allocated and zeroed) by its first batch: one untimed step per stream in flight
        log("[bench] --warmup raised from %d to %d: one untimed step per stream in flight" % (args.warmup, depth))
        args.warmup = depth
    streams = [cuda.Stream(device=dev) for _ in range(depth)] if depth > 1 else [cuda.current_stream()]
    for s_ in streams:
        s_.wait_stream(cuda.current_stream())
    issued = [0] * depth
    step_no = [0]
    pending = []      # N-GPU path: batches issued whose records are not packed and gathered yet (oldest first)

    def gather_oldest():
        # N-GPU path: the per-GPU result records are cut to size and merged on rank 0 over RCCL (the only exchange of the job) -- on the batch's own
        # stream; the size read-back waits for that batch alone, the batch issued after it keeps the device busy meanwhile
        k, res, e0, record = pending.pop(0)
        with cuda.stream(streams[k]):
            packed, offs = ctx.results_pack(res, n, P.khits)
            last["gathered"] = shard.gather_packed(dist, packed, int(offs[n].item()), dev)
            e1 = ev()
            e1.record()
            if record:
                stage_events.append((e0, e1))

    def step(record):
        k = step_no[0] % depth
        step_no[0] += 1
        with cuda.stream(streams[k]):
            if depth > 1 and record and issued[k]:
                # the batch issued `depth` steps ago on this stream: its kernel times (blocks until it is done -- the pipeline's backpressure)
                kern_times.append(ctx.align_timing(on_current_stream=True))
            e0 = ev()
            e0.record()
            res, stride = ctx.align_batch(batch, rp_t, P, args.readlen)
            issued[k] = 1 if record else 0
            if dist is not None:
                pending.append((k, res, e0, record))
            else:
                e1 = ev()
                e1.record()
                if record:
                    stage_events.append((e0, e1))
            if record and depth == 1:
                kern_times.append(ctx.align_timing())      # HIP events recorded by the library around each kernel
            last["res"], last["stride"] = res, stride
        while len(pending) > depth - 1:
            gather_oldest()

    def sync_all():
        while pending:
            gather_oldest()
        cuda.synchronize()
        if dist is not None:
            dist.barrier()
            cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    sync_all()
    ctx.align_profile(reset=True)
    ctx.counters(reset=True)
    # the timed region: K steps between barrier + synchronize, `reps` times over (value = the median region; pe-vsens: its step time is the
    # tail of its slowest pairs, and one region of 10 steps ranged 1.03-1.49 M reads/s over four runs on one box in round 5)
    rep_dts = []
    for _rep in range(max(1, args.reps)):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(True)
        sync_all()
        rep_dts.append(shard.reduce_max(dist, time.perf_counter() - t0, dev))      # the slowest rank defines the step time
        if depth > 1:
            for k in range(depth):                # the last batch of every stream
                if issued[k]:
                    with cuda.stream(streams[k]):
                        kern_times.append(ctx.align_timing(on_current_stream=True))
                    issued[k] = 0
    dt = sorted(rep_dts)[len(rep_dts) // 2]
    batch_ms = sum(a.elapsed_time(bb) for a, bb in stage_events) / len(stage_events)
    kavg = {k: sum(t[k] for t in kern_times) / len(kern_times) for k in kern_times[0]}
    cnt = ctx.counters()
    n_timed_steps = args.steps * max(1, args.reps)
    # With several steps in flight the kernels of different batches overlap and their HIP-event durations are not additive: the roofline block
    # of such a configuration is taken from extra steps issued ONE AT A TIME on one stream, outside the timed region.
    kavg_serial = None
    if depth > 1:
        ks = []
        with cuda.stream(streams[0]):
            for _ in range(2):
                ctx.align_batch(batch, rp_t, P, args.readlen)
                cuda.synchronize()
                ks.append(ctx.align_timing(on_current_stream=True))
        kavg_serial = {k: sum(t[k] for t in ks) / len(ks) for k in ks[0]}
    kroof = kavg_serial or kavg
    kern_ms = kroof["k_align_reads"]
    determinism = None
    if args.check_determinism and not dry:
        with cuda.stream(streams[0]):
            ra, st_ = ctx.align_batch(batch, rp_t, P, args.readlen); cuda.synchronize(); ra = ra.view(n, st_).clone()
            rb, _ = ctx.align_batch(batch, rp_t, P, args.readlen); cuda.synchronize(); rb = rb.view(n, st_)
            diff = (ra != rb).any(dim=1)
            idx = diff.nonzero().flatten()
            first = []
            for i in idx[:8].tolist():
                cols = (ra[i] != rb[i]).nonzero().flatten()
                first.append({"read": i, "first_byte": int(cols[0]), "bytes_differing": int(cols.numel()), "aligned": [int(ra[i, 1]), int(rb[i, 1])]})
            hdr_diff = (ra[:, :135] != rb[:, :135]).any(dim=1)      # the record's header (status, counters, scores) and the first alignment's fields up to its edits
            colhist = (ra != rb).sum(dim=0)
            cols_ = colhist.nonzero().flatten().tolist()
            determinism = {"records": n, "records_differing_between_two_runs": int(idx.numel()), "records_differing_in_header_or_first_alignment_fields": int(hdr_diff.sum()),
                           "header_diff_reads": hdr_diff.nonzero().flatten()[:16].tolist(), "first": first,
                           "differing_byte_offsets": {int(c_): int(colhist[c_]) for c_ in cols_[:64]}, "n_differing_offsets": len(cols_)}
            log("[bench] determinism: %d of %d result records differ between two runs of the same batch" % (int(idx.numel()), n))
    # the worker's phase timers are off in the timed steps (measured: they cost nothing beyond run-to-run noise, but the timed
    # region is the product configuration); one more, untimed, pass over the same batch with them on gives the per-phase breakdown
    P.profile = 1
    ctx.align_profile(reset=True)
    step(False)
    sync_all()
    prof = ctx.align_profile()
    P.profile = 0

    # per-read work counters come back in the result records
    stride = last["stride"]
    rec = last["res"].view(n, stride)[:, :C.sizeof(b.ReadResult) - C.sizeof(b.Aln)].cpu().numpy()
    hdr = np.dtype([("status", "u1"), ("aligned", "u1"), ("maxed", "u1"), ("filt", "u1"), ("exhausted", "u1"), ("has_secbest", "u1"),
                    ("pad", "u1", 2), ("secbest", "<i4"), ("best", "<i4"), ("nalns", "<u4"), ("nreport", "<u4"),
                    ("n_ex_iters", "<u4"), ("n_ex_dps", "<u4"), ("n_ex_ugs", "<u4"), ("n_dp_fail_streak_max", "<u4"),
                    ("n_bwops_seed", "<u4"), ("n_bwops_ext", "<u4"), ("n_redundants", "<u4"), ("n_bt_attempts", "<u4"),
                    ("n_ext_left", "<u4"), ("n_ext_right", "<u4"), ("n_resolve_steps", "<u4"), ("n_sides", "<u4"),
                    ("pair_best", "<i4"), ("pair_secbest", "<i4"), ("n_mate_dps", "<u4"), ("pad2", "<u4")])
    h = np.frombuffer(rec.tobytes(), dtype=hdr)
    aligned = int(h["aligned"].sum())
    all_aligned = shard.reduce_sum(dist, [aligned], dev)[0]

    if rank == 0:
        steps = args.steps
        side = info.side_sz
        off_sz = info.off_size
        rankq = float(h["n_bwops_seed"].sum() + h["n_bwops_ext"].sum())
        sides_per_launch = float(prof[8])        # the profiled pass is ONE launch over the same batch (the profile was reset before it)
        dp_cells = float(prof[24] + prof[25])   # DP cells actually computed in that launch: band cells of the score-only passes (up to their early exit) + of the fills that store a matrix
        # Algorithmic bytes.  k_align_reads, the dominant kernel: the rank queries it still issues itself
        # (offset resolution, re-seeding rounds) * side_sz + DP reference windows + reads in + result records out.
        dp_windows = float(h["n_ex_dps"].sum() + h["n_mate_dps"].sum())
        win_cols = args.readlen + 4 * 15 + 1     # seed-extension windows; opposite-mate windows are wider (counted at the same size: a lower bound)
        # (result records: header + khits alignment slots of bt2g_aln -- the record as SURVEY 8d priced it in every round; the stride of the result buffer
        # also leaves room for the long-read class's larger slots since round 5, which is not traffic)
        rec_bytes = (C.sizeof(b.ReadResult) + (max(1, P.khits) - 1) * C.sizeof(b.Aln) + 15) & ~15
        alg_bytes = sides_per_launch * side + dp_windows * ((win_cols + 3) // 4) + n * args.readlen * 2 + n * rec_bytes
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        # The four lane-per-task FM kernels in front of it carry most of the rank queries of the path.
        fm_ms = sum(v for k, v in kroof.items() if k != "k_align_reads")
        fm_bytes = (cnt.rank_queries * side + cnt.ftab_lookups * 2 * off_sz + cnt.sa_lookups * off_sz) / float(n_timed_steps) + n * args.readlen * 2
        fm_achieved = fm_bytes / (fm_ms * 1e-3) / 1e9 if fm_ms > 0 else 0.0
        # the same requests priced at what this layout moves for them: a rank query reads ONE 64-byte block of the device rank index (occ words +
        # both bit planes, bt2g_device.hpp Blk) instead of a reference side, an offset lookup one 8-byte entry of the full suffix array
        fm_phys = (cnt.rank_queries * 64 + cnt.ftab_lookups * 2 * off_sz + cnt.sa_lookups * 8) / float(n_timed_steps) + n * args.readlen * 2
        fm_phys_achieved = fm_phys / (fm_ms * 1e-3) / 1e9 if fm_ms > 0 else 0.0
        kname = "k_align_pairs" if args.paired else "k_align_reads"
        traffic, traffic_src = pmc_traffic(kname, n, args.config)
        cb, par = None, None
        if not args.no_cpu_baseline and world == 1:      # reported at N=1 only
            ns = min(args.cpu_sample, n) & ~1
            ntiny = 1000
            work = cache_dir()
            if args.paired:
                # mates interleaved in the batch: read 2i = mate 1, 2i+1 = mate 2 of pair i
                fq_all = (os.path.join(work, "sample_1.fq"), os.path.join(work, "sample_2.fq"))
                fq_tiny = (os.path.join(work, "tiny_1.fq"), os.path.join(work, "tiny_2.fq"))
                for m in (0, 1):
                    write_fastq_fixed(fq_all[m], seq[m:ns:2], qual[m:ns:2], names[m:ns:2])
                    write_fastq_fixed(fq_tiny[m], seq[m:ntiny:2], qual[m:ntiny:2], names[m:ntiny:2])
            else:
                fq_all, fq_tiny = os.path.join(work, "sample.fq"), os.path.join(work, "tiny.fq")
                write_fastq_fixed(fq_all, seq[:ns], qual[:ns], names[:ns])
                write_fastq_fixed(fq_tiny, seq[:ntiny], qual[:ntiny], names[:ntiny])
            cb, par = cpu_baseline_and_parity(base, large, fq_all, fq_tiny, ns, ntiny, threads, cfg["args"], work, args.readlen, args.parity_only, keep_ref_sam=e2e_fq is not None)
            if par is not None and "parity_sample_aligned_reads" in par and not args.paired:
                # the timed batch starts with the same reads, same parameters, same per-read seeds: its records must agree
                par["timed_batch_aligned_reads_same_sample"] = int(h["aligned"][:ns].sum())
        gname = ("REAL reference " + real) if real else ("E. coli K-12-like synthetic genome" if bacterial else "hg38-like synthetic genome (hg38 unavailable offline)")
        cfg_no = {"ecoli100": "configs[1]", "se150": "configs[2]", "pe-vsens": "configs[3]", "local400": "configs[4]"}.get(args.config, "(extra) " + args.config)
        what_reads = "%d x %d bp %s per GPU per step, %s" % (n, args.readlen, "reads as %d pairs" % (n // 2) if args.paired else "SE reads", cfg["what"])
        idx_txt = ".%s index (side %d B, %d-byte offsets) %s" % (ext, side, off_sz, "as given (--index-base)" if args.index_base else "built by the GPU index builder")
        if real:
            genome_txt = "REAL reference %s" % real
        elif bacterial:
            genome_txt = ("E. coli K-12-like synthetic %.2f Mbp genome (one chromosome; 7 rRNA-operon-like and 40 IS-like repeat copies; the real sequence is not available offline)" % args.genome_mbp)
        else:
            genome_txt = ("hg38-like synthetic %d Mbp genome%s (%d chromosomes; ~45 %% repeats: Alu-/L1-like and older diverged families, simple repeats, segmental duplications; N gaps)"
                          % (args.genome_mbp, " = hg38 scale" if args.genome_mbp >= 3000 else "", N_CHROMS))
        timed_volume = "timed region: %d x the same resident batch of %d reads per GPU%s" % (steps, n, " (x %d regions, median)" % len(rep_dts) if len(rep_dts) > 1 else "")
        workload = "BASELINE.json %s: %s, %s, %s; %s" % (cfg_no, genome_txt, idx_txt, what_reads, timed_volume)
        res = {
            "metric": ("aligned reads/sec (whole node), 2 x %d bp PE (mates counted as reads), %s" % (args.readlen, gname) if args.paired else
                       "aligned reads/sec (whole node), %d bp SE vs %s, %s index" % (args.readlen, gname, "large" if large else "small")),
            "value": shard.throughput(world, n, steps, dt),
            "unit": "reads/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3,
            "reps": {"timed_regions": len(rep_dts), "reads_per_s_each": [round(shard.throughput(world, n, steps, x)) for x in rep_dts],
                     "spread_frac": (max(rep_dts) - min(rep_dts)) / dt, "value_is": "the median region"} if len(rep_dts) > 1 else None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u32" if off_sz == 4 else "u8/u64", "data": "DRY RUN (--dry-ranks): no device, no alignment; only the N-rank plumbing is real" if dry else "synthetic",
            "config": {
                "workload": workload,
                "genome": real or "synthetic", "reads_source": args.reads_fastq or "SURVEY 8d generator",
                "config_name": args.config, "command_line": " ".join(cfg["args"]), "determinism": determinism,
                "steps_in_flight": depth,
                "steps_in_flight_note": None if depth == 1 else "the timed steps are issued on %d alternating streams (one working set of the context each), so that a batch's tail is filled by the next batch, as in the product driver; kernel_ms_per_step are the HIP-event durations of the kernels on their own stream and overlap in time" % depth,
                "stages_timed": "one bt2g_align_batch per step = the whole per-read worker: k_exact_sweep, k_one_mm, k_seed_search_exact, k_extend_hits (lane-per-task FM kernels) then k_align_reads "
                                "(rank+prioritise, offset resolution, re-seeding, SW fill + backtrace, -M reporting)",
                "not_in_timed_region": "FASTQ parse and SAM text formatting (host side, SURVEY.md 8f)",
                "n_gpu_merge": None if world == 1 else "every step also packs the result records (bt2g_results_pack) and gathers them to rank 0 over RCCL: %d bytes arrived on rank 0 in the last step" % sum(int(t.numel()) for t in last["gathered"]),
                "fraction_aligned": all_aligned / float(world * n),
                "index_bytes_hbm": int(info.hbm_bytes), "side_sz": int(side), "off_size": int(off_sz),
                "index_build": build_info, "index_load_s": round(index_load_s, 3),
                "bw_ops_per_read": rankq / n, "dp_fills_per_read": float(h["n_ex_dps"].sum()) / n,
                "backtraces_per_read": float(h["n_bt_attempts"].sum()) / n,
                "reads_overflowed": int((h["status"] != 0).sum()),
                "worker_phase_us_per_read_profiled_pass": dict(zip(["sweep", "mm1", "seeds", "rank_prioritise", "resolve", "dp_fill", "backtrace", "whole_read", "gather_cells", "report", "ungapped",
                                                      "bt_tile_fetch", "gather_lastrow", "gather_zero_masks", "prioritize_collect_extend", "prioritize_row_sampling", "sink_report", "opposite_mate_total"],
                                                     [round(prof[i] / 100.0 / max(1, prof[9]), 1) for i in (0, 1, 2, 3, 4, 5, 6, 7, 10, 11, 12, 16, 17, 18, 19, 20, 21, 22)])),
                "backtrace_profile_per_read": {"walk_us": round(prof[27] / 100.0 / max(1, prof[9]), 1), "walk_us_successful": round(prof[28] / 100.0 / max(1, prof[9]), 1),
                                               "successful_walks": prof[29] / max(1, prof[9]), "tail_us_after_successful_trace": round(prof[30] / 100.0 / max(1, prof[9]), 1),
                                               "scalar_steps": prof[31] / max(1, prof[9])},
                "worker_counts_per_read": {"bt_steps": prof[13] / max(1, prof[9]), "bt_tiles": prof[14] / max(1, prof[9]), "cand_cells": prof[15] / max(1, prof[9]),
                                           "sampled_rows": (prof[23] & 0xffffffff) / max(1, prof[9]), "sampled_rows_in_batches": (prof[23] >> 32) / max(1, prof[9]),
                                           "sampler_batches": (prof[26] >> 32) / max(1, prof[9])},
                "kernel_ms_per_step": {("k_align_pairs" if args.paired and k == "k_align_reads" else k): round(v, 3) for k, v in kavg.items()}, "batch_ms_events": round(batch_ms, 3),
                "kernel_ms_per_step_one_at_a_time": None if kavg_serial is None else {("k_align_pairs" if args.paired and k == "k_align_reads" else k): round(v, 3) for k, v in kavg_serial.items()},
                "kernel_ms_note": None if kavg_serial is None else "kernel_ms_per_step are HIP-event durations inside the timed region, where the kernels of %d batches in flight overlap (not additive); "
                                  "the roofline blocks use kernel_ms_per_step_one_at_a_time: two extra steps issued alone on one stream after the timed region" % depth,
                "fm_kernels_sides_per_read": cnt.rank_queries / float(n * n_timed_steps),
                "sides_per_read": prof[8] / max(1, prof[9]),
            },
            "roofline": {"bound": "hbm", "kernel": kname, "sides_per_launch": sides_per_launch, "dp_windows_per_launch": dp_windows, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": kern_ms,
                         "dp_gcups": dp_cells / (kern_ms * 1e-3) / 1e9,
                         "dp_cells_note": "cells computed per launch (band cells; score-only passes up to their early exit: %d, matrix-storing fills: %d), over the whole kernel time"
                                          % (prof[24], prof[25]),
                         "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
                         "instruction_issue": pmc_issue(kname, kern_ms, n, 256 if dry else torch.cuda.get_device_properties(dev).multi_processor_count, args.config),
                         "fm_kernels": {"kernels": ["k_exact_sweep", "k_one_mm", "k_seed_search_exact", "k_extend_hits"], "bound": "hbm",
                                        "ms_per_launch_sum": fm_ms, "algorithmic_bytes_per_launch": int(fm_bytes),
                                        "achieved": fm_achieved, "unit": "GB/s", "frac": fm_achieved / HBM_PEAK_GBS,
                                        "physical_bytes": int(fm_phys), "physical_achieved": fm_phys_achieved, "physical_frac": fm_phys_achieved / HBM_PEAK_GBS,
                                        "physical_note": "64-B rank blocks, 8-B suffix-array entries, ftab pairs, reads; the k_extend_hits interval also holds the re-seeding rounds' seed search"}},
        }
        if par:
            res["config"].update(par)
        res["cpu_baseline"] = cb
        if e2e_fq is not None:
            # the end-to-end leg runs once this process is gone (outer()): what it needs travels in the line
            res["e2e_pending"] = {"base": base, "large": large, "fq": e2e_fq, "preset": cfg["args"], "threads": threads, "work": cache_dir(), "world": world,
                                  "par": {"parity_identical": par.get("parity_identical")} if par else None}
            if dry:
                # no device: the stand-in engine (dry_engine) behind the real bowtie2_amd.mgpu, over gloo
                eng = os.path.join(cache_dir(), "dry_engine.sh")
                with open(eng, "w") as f:
                    f.write("#!/bin/sh\nexec %s %s --dry-engine \"$@\"\n" % (sys.executable, os.path.abspath(__file__)))
                os.chmod(eng, 0o755)
                res["e2e_pending"]["mgpu_args"] = ["--engine", eng, "--backend", "gloo"]
        # what outer() does after this process: the live PMC passes and the other BASELINE configurations belong to the default N = 1 headline run only
        full = world == 1 and not dry and not args.no_cpu_baseline and not args.parity_only
        extra = args.extra_configs
        if extra is None:
            extra = "ecoli100,pe-vsens,local400" if (full and args.config == "se150") else ""
        fwd = []
        if "--genome-mbp" in sys.argv:
            fwd += ["--genome-mbp", str(int(args.genome_mbp))]
        res["_outer"] = {"pmc": full and not args.no_pmc_pass, "kernel": kname, "config": args.config, "reads": n,
                         "extra_configs": [c for c in extra.split(",") if c and c != args.config], "extra_steps": args.extra_steps, "warmup": args.warmup,
                         "forward": fwd, "genome_fasta": args.genome_fasta if "--genome-fasta" in sys.argv else None}
        print_line(res)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def summarize_config(d):
    """One extra configuration's line, cut to what the headline's `configs` block carries."""
    c, r, cb = d.get("config", {}), d.get("roofline", {}), d.get("cpu_baseline") or {}
    return {"metric": d.get("metric"), "value": d.get("value"), "unit": d.get("unit"), "ms_per_step": d.get("ms_per_step"), "steps": d.get("steps"), "reps": d.get("reps"),
            "workload": c.get("workload"), "steps_in_flight": c.get("steps_in_flight"),
            "parity_identical": c.get("parity_identical"), "parity_checked_reads": c.get("parity_checked_reads"), "parity_differing_sam_lines": c.get("parity_differing_sam_lines"),
            "reads_overflowed": c.get("reads_overflowed"), "fraction_aligned": c.get("fraction_aligned"),
            "kernel_ms_per_step": c.get("kernel_ms_per_step"), "kernel_ms_per_step_one_at_a_time": c.get("kernel_ms_per_step_one_at_a_time"),
            "roofline": {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "algorithmic_bytes_per_launch", "traffic", "traffic_source",
                                               "traffic_over_algorithmic", "dp_gcups")},
            "fm_kernels_physical_frac": (r.get("fm_kernels") or {}).get("physical_frac"),
            "cpu_baseline": {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample")} if cb else None,
            "speedup_vs_cpu_baseline": (d.get("value") / cb["value"]) if cb.get("value") else None}


def apply_live_traffic(res, lp):
    r = res.get("roofline")
    if not r:
        return
    if lp and "traffic_bytes_per_launch" in lp:
        r["traffic"] = lp["traffic_bytes_per_launch"]
        r["traffic_source"] = ("live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (one pass each) around this script's own measuring process in this run, %d dispatches of %s on the same "
                               "resident batch; 2 x FETCH_SIZE + WRITE_SIZE (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); libbt2g.so sha256 %s"
                               % (lp["dispatches"], r.get("kernel"), (lib_sha256() or "?")[:16]))
        r["traffic_live"] = lp
        if r.get("algorithmic_bytes_per_launch"):
            r["traffic_over_algorithmic"] = r["traffic"] / r["algorithmic_bytes_per_launch"]
    elif lp:
        r["traffic_live"] = lp      # {"error": ...}: the committed summary (if it is this library's) or null stays


def print_line(res):
    """The ONE JSON line, as the last thing on stdout: whatever native libraries left in the C stdio buffer (RCCL prints a version banner there when
    the process group is created, and it would otherwise surface at exit, behind the line) is flushed first."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.write(json.dumps(res) + "\n")
    sys.stdout.flush()


def run_child(argv, timeout_s=None):
    env = dict(os.environ, BT2_BENCH_CHILD="1")
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__)] + list(argv), stdout=subprocess.PIPE, env=env, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return None, "timed out after %s s" % timeout_s, 124
    js = [l for l in p.stdout.splitlines() if l.startswith("{")]      # (the line; anything a native library printed around it is not)
    line = js[-1] if js else ""
    try:
        return json.loads(line), None, p.returncode
    except ValueError:
        return None, p.stdout[-400:], p.returncode      # (ranks other than 0 print no line)


def outer():
    """The measurement runs in a child process; what needs the GPU(s) to itself runs after the child has exited: the end-to-end leg -- the drop-in binary
    (N = 1) or the product's N-GPU driver bowtie2_amd.mgpu (N > 1), FASTQ file -> SAM file, the reference's own timed quantity (bt2_search.cpp:4863
    "Multiseed full-index search") -- then, in the default N = 1 run, the live rocprofv3 --pmc passes and the other BASELINE configurations, each a process of
    its own.  Measured (profiles/r05g_*): next to a second process that merely holds a context on the same GPU (this script with its torch runtime, idle) the
    binary runs at 0.55 x its rate on a GPU of its own; a user's run has the GPU to itself."""
    if len(sys.argv) > 1 and sys.argv[1] == "--dry-engine":
        raise SystemExit(dry_engine(sys.argv[2:]))
    if os.environ.get("BT2_BENCH_CHILD"):
        return main()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    res, err, rc = run_child(sys.argv[1:])
    if world > 1:
        # every rank's measuring process has to be gone before the N-GPU e2e job starts: the parents meet in a directory named after their launcher
        syncd = os.path.join(cache_dir(), "sync_%d" % os.getppid())
        os.makedirs(syncd, exist_ok=True)
        with open(os.path.join(syncd, "rank%d.tmp" % rank), "w") as f:
            f.write(str(rc))
        os.replace(os.path.join(syncd, "rank%d.tmp" % rank), os.path.join(syncd, "rank%d" % rank))
        if rank != 0:
            raise SystemExit(rc)
        if res is None:
            sys.stdout.write(err or "")
            raise SystemExit(rc or 1)
        deadline = time.time() + float(os.environ.get("BT2_BENCH_SYNC_WAIT_S", "1800"))
        rcs = {}
        while len(rcs) < world and time.time() < deadline:
            for r in range(world):
                fn = os.path.join(syncd, "rank%d" % r)
                if r not in rcs and os.path.exists(fn):
                    rcs[r] = open(fn).read().strip()
            if len(rcs) < world:
                time.sleep(0.1)
        import shutil
        shutil.rmtree(syncd, ignore_errors=True)
        pend = res.pop("e2e_pending", None)
        res.pop("_outer", None)
        if pend:
            if len(rcs) == world and all(v == "0" for v in rcs.values()):
                res["e2e"] = e2e_leg_ranks(pend, world, res["value"])
            else:
                res["e2e"] = {"error": "not run: the measuring processes of the ranks did not all exit cleanly (%s)" % rcs}
        print_line(res)
        raise SystemExit(rc)
    if res is None:
        sys.stdout.write(err or "")
        raise SystemExit(rc or 1)
    pend = res.pop("e2e_pending", None)
    todo = res.pop("_outer", None) or {}
    if pend:
        res["e2e"] = e2e_leg(pend["base"], pend["large"], pend["fq"], pend["preset"], pend["threads"], res["value"], pend["work"], pend["par"])
        res["config"]["workload"] += "; e2e leg: %d distinct reads, FASTQ file -> SAM file through the drop-in binary" % pend["fq"]["reads"]
    t_outer = time.time()
    budget = float(os.environ.get("BT2_BENCH_EXTRA_BUDGET_S", "700"))      # the legs below stop being started once this much time has gone into them
    if todo.get("pmc"):
        apply_live_traffic(res, live_pmc_traffic(todo["config"], todo["kernel"], ["--reads", str(todo["reads"])] + todo["forward"]))
    if todo.get("extra_configs"):
        res["configs"] = {}
        for name in todo["extra_configs"]:
            if time.time() - t_outer > budget:
                res["configs"][name] = {"error": "not run: the time budget of the extra legs (BT2_BENCH_EXTRA_BUDGET_S = %d s) was spent" % budget}
                continue
            argv = ["--config", name, "--steps", str(todo["extra_steps"]), "--warmup", str(min(todo["warmup"], 3)), "--parity-only", "--e2e-reads", "0", "--extra-configs", ""] + todo["forward"]
            if todo.get("genome_fasta") and CONFIGS[name].get("genome") != "ecoli":
                argv += ["--genome-fasta", todo["genome_fasta"]]
            d, err2, rc2 = run_child(argv, timeout_s=600)
            if d is None:
                res["configs"][name] = {"error": (err2 or "")[-300:], "returncode": rc2}
                continue
            o2 = d.pop("_outer", None) or {}
            d.pop("e2e_pending", None)
            if todo.get("pmc") and time.time() - t_outer < budget:
                apply_live_traffic(d, live_pmc_traffic(name, o2.get("kernel", "k_align_reads"), todo["forward"], timeout_s=240))
            res["configs"][name] = summarize_config(d)
        res["configs_note"] = ("BASELINE.json configs[1], [3], [4] measured in this same run, each in a process of its own after the headline: %d timed steps, one run of the unmodified "
                               "reference on the configuration's sample for cpu_baseline and the byte-for-byte SAM comparison (--parity-only protocol: not a median of 3)" % todo["extra_steps"])
    print_line(res)
    raise SystemExit(rc)


if __name__ == "__main__":
    outer()

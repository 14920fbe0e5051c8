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


def build_index_gpu(base, G, chrom_lens, large, device_index):
    """The GPU index builder on the in-memory genome -> <base>.{1,2,3,4,rev.1,rev.2}.bt2[l]; returns its stats."""
    import numpy as np
    import torch
    import bowtie2_amd as b
    lut = torch.tensor([ord(c) for c in "ACGTN"], dtype=torch.uint8, device=G.device)
    asc = lut[G.long()].cpu().numpy()
    names, seqs, o = [], [], 0
    for i, L in enumerate(chrom_lens):
        names.append("chr%d" % (i + 1))
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
    for j in range(names_t.shape[1]):
        acc ^= (names_t[:, j].long() << ((j & 3) << 3)) & 0xffffffff
    return (acc & 0xffffffff)


def write_fastq_fixed(path, seq, qual, names, append=False):
    """FASTQ of equal-length reads, assembled as one 2-D byte array."""
    import numpy as np
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[seq.cpu().numpy()]
    q = qual.cpu().numpy()
    n, L = s.shape
    w = names.shape[1]
    rec = np.empty((n, 1 + w + 1 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1:1 + w] = names; rec[:, 1 + w] = 10
    o = 2 + w
    rec[:, o:o + L] = s; rec[:, o + L] = 10; rec[:, o + L + 1] = ord("+"); rec[:, o + L + 2] = 10
    o2 = o + L + 3
    rec[:, o2:o2 + L] = q; rec[:, o2 + L] = 10
    with open(path, "ab" if append else "wb") as f:
        f.write(rec.tobytes())


# BASELINE.json configs[2..4] (+ the round-2 paired line).  `args` is the bowtie2-align command line of the configuration: the timed
# batches take their parameters from it through bt2g_cli_params (the drop-in binary's own option parser), the reference and the product
# binary of the parity check are run with it.
CONFIGS = {
    "se150":    {"args": ["--sensitive"], "paired": False, "readlen": 150, "reads": 2_000_000, "cpu_sample": 1_000_000,
                 "what": "--sensitive (-D 15 -R 2 -N 0 -L 22 -i S,1,1.15), end-to-end"},
    # "pipeline": steps in flight.  A batch of pairs or of long --local reads ends in a tail -- a few pathological reads keep a handful of the
    # 4096 waves busy for hundreds of ms after the rest are done -- which the next batch's waves fill when two batches are in flight (as in
    # the product driver, whose device-stage threads each issue their batch on their own stream).  The headline has no such tail: 1.
    "pe-sens":  {"args": ["--sensitive"], "paired": True, "readlen": 150, "reads": 400_000, "cpu_sample": 400_000, "pipeline": 3,
                 "what": "pairs, --sensitive, --fr -I 0 -X 500"},
    "pe-vsens": {"args": ["--very-sensitive", "-X", "500"], "paired": True, "readlen": 150, "reads": 400_000, "cpu_sample": 200_000, "pipeline": 3,
                 "what": "pairs, --very-sensitive (-D 20 -R 3 -N 0 -L 20 -i S,1,0.50), --fr -I 0 -X 500 (mate rescue)"},
    "local400": {"args": ["--local"], "paired": False, "readlen": 400, "reads": 200_000, "cpu_sample": 100_000, "pipeline": 2,
                 "what": "--local = --sensitive-local (-D 15 -R 2 -N 0 -L 20 -i S,1,0.75, --ma 2, --score-min G,20,8)"},
    # BASELINE.json configs[1]: a bacterial genome behind a small (.bt2: 64-byte sides, 32-bit offsets) index -- the uint32_t instantiations
    "ecoli100": {"args": ["--sensitive"], "paired": False, "readlen": 100, "reads": 1_000_000, "cpu_sample": 1_000_000, "genome": "ecoli",
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


def pmc_traffic(kernel, reads_per_launch, config="se150"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE + WRITE_SIZE, KiB),
    scaled from the reads-per-launch of that profile run to this run's.  PMC collection needs rocprofv3 around the
    process, so it cannot be taken inside the timed run; profiles/README.md has the recipe."""
    files = pmc_files(config)
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        k = d["kernels"][kernel]
        # MI355X_MICROARCH.md, HBM section: rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests as 64 bytes
        # (calibrated on wide coalesced reads) -> doubled; WRITE_SIZE is taken as reported (uncalibrated there)
        per_read = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0 / (d["reads_per_launch"] * d["launches"])
        return int(per_read * reads_per_launch), "profiles/%s ((2 x FETCH_SIZE + WRITE_SIZE) per read x %d reads; FETCH_SIZE doubled as the guide prescribes for gfx950)" % (os.path.basename(files[-1]), reads_per_launch)
    except Exception:
        return None, None


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
        return {"source": "profiles/" + os.path.basename(files[-1]), "valu_per_read": round(valu), "salu_per_read": round(salu), "lds_per_read": round(lds),
                "vmem_per_read": round(vmem), "simd_cycles_per_read": round(simd_cycles_per_read), "valu_busy_frac": round(valu * 4 / simd_cycles_per_read, 3),
                "note": "k_align_reads is one serial instruction stream per read (one wavefront each, 4 or 5 per SIMD): it is bound by the issue rate and the "
                        "dependent latencies of that stream, not by HBM; the hbm fraction above is reported because the contract asks for it"}
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
    ext = "bt2l" if large else "bt2"
    base = os.path.join(cache_dir(), "ecolilike_s3_%s" % ext if bacterial else "hg38like_%dmbp_s2_%s" % (args.genome_mbp, ext))
    if dry:
        base += "_dry"
    # ---- workload: genome (every rank, same seed) + index (rank 0 builds it on its GPU, the others wait) ----
    t0 = time.time()
    if bacterial:
        G, chrom_lens = synth_genome_bacterial(3, dev)
        args.genome_mbp = G.numel() / 1e6
    else:
        G, chrom_lens = synth_genome_gpu(args.genome_mbp, 2, dev)
    cuda.synchronize()
    log("[bench] genome: %.4g Mbp generated in %.1fs" % (args.genome_mbp, time.time() - t0))
    build_info = None
    if rank == 0 and not os.path.exists(base + ".rev.2." + ext):
        cuda.empty_cache()       # the builder allocates ~30 bytes per base with hipMalloc, next to torch's caching allocator
        build_info = _dry_build_index(base, ext) if dry else build_index_gpu(base, G, chrom_lens, large, local_rank)
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
    if args.paired:
        # mates interleaved: read 2i = forward mate at the fragment start, read 2i+1 = reverse-complemented mate at its end
        seq, qual = synth_pairs_gpu(G, n // 2, args.readlen, shard.shard_seed(2000, rank), dev)
        n = seq.shape[0]
    else:
        seq, qual = synth_reads_gpu(G, n, args.readlen, shard.shard_seed(1000, rank), dev)
    names = read_names(rank * n, n)
    # ---- the end-to-end leg's input (N = 1): the timed batch followed by further batches of DISTINCT reads of the same generator, as FASTQ ----
    if args.e2e_reads < 0:
        args.e2e_reads = 0 if (args.no_cpu_baseline or args.parity_only) else 12 * n
    e2e_fq = None
    if world == 1 and args.e2e_reads > 0:
        e2e_fq = write_e2e_fastq(G, seq, qual, names, n, args, dev, rank)
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
    depth = max(1, args.pipeline) if dist is None else 1      # (the N-GPU path keeps one step in flight: its RCCL gather sits inside the step)
    if args.warmup < depth:
        # every stream's working set is created (its arena allocated and zeroed) by its first batch: one untimed step per stream in flight
        log("[bench] --warmup raised from %d to %d: one untimed step per stream in flight" % (args.warmup, depth))
        args.warmup = depth
    streams = [cuda.Stream(device=dev) for _ in range(depth)] if depth > 1 else [cuda.current_stream()]
    for s_ in streams:
        s_.wait_stream(cuda.current_stream())
    issued = [0] * depth
    step_no = [0]

    def step(record):
        k = step_no[0] % depth
        step_no[0] += 1
        with cuda.stream(streams[k]):
            if depth > 1 and record and issued[k]:
                # the batch issued `depth` steps ago on this stream: its kernel times (blocks until it is done -- the pipeline's backpressure)
                kern_times.append(ctx.align_timing(on_current_stream=True))
            e0, e1 = ev(), ev()
            e0.record()
            res, stride = ctx.align_batch(batch, rp_t, P, args.readlen)
            if dist is not None:
                # N-GPU path: the per-GPU result records are cut to size and merged on rank 0 over RCCL (the only exchange of the job)
                packed, offs = ctx.results_pack(res, n, P.khits)
                last["gathered"] = shard.gather_packed(dist, packed, int(offs[n].item()), dev)
            e1.record()
            issued[k] = 1 if record else 0
            if record:
                stage_events.append((e0, e1))
                if depth == 1:
                    kern_times.append(ctx.align_timing())      # HIP events recorded by the library around each kernel
            last["res"], last["stride"] = res, stride

    def sync_all():
        cuda.synchronize()
        if dist is not None:
            dist.barrier()
            cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    sync_all()
    ctx.align_profile(reset=True)
    ctx.counters(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    sync_all()
    dt = time.perf_counter() - t0
    dt = shard.reduce_max(dist, dt, dev)      # the slowest rank defines the step time
    if depth > 1:
        for k in range(depth):                # the last batch of every stream
            if issued[k]:
                with cuda.stream(streams[k]):
                    kern_times.append(ctx.align_timing(on_current_stream=True))
                issued[k] = 0
    batch_ms = sum(a.elapsed_time(bb) for a, bb in stage_events) / len(stage_events)
    kavg = {k: sum(t[k] for t in kern_times) / len(kern_times) for k in kern_times[0]}
    kern_ms = kavg["k_align_reads"]
    cnt = ctx.counters()
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
        fm_ms = sum(v for k, v in kavg.items() if k != "k_align_reads")
        fm_bytes = (cnt.rank_queries * side + cnt.ftab_lookups * 2 * off_sz + cnt.sa_lookups * off_sz) / float(args.steps) + n * args.readlen * 2
        fm_achieved = fm_bytes / (fm_ms * 1e-3) / 1e9 if fm_ms > 0 else 0.0
        # the same requests priced at what this layout moves for them: a rank query reads ONE 64-byte block of the device rank index (occ words +
        # both bit planes, bt2g_device.hpp Blk) instead of a reference side, an offset lookup one 8-byte entry of the full suffix array
        fm_phys = (cnt.rank_queries * 64 + cnt.ftab_lookups * 2 * off_sz + cnt.sa_lookups * 8) / float(args.steps) + n * args.readlen * 2
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
        res = {
            "metric": ("aligned reads/sec (whole node), 2 x %d bp PE (mates counted as reads), hg38-like synthetic genome" % args.readlen if args.paired else
                       "aligned reads/sec (whole node), %d bp SE vs E. coli K-12-like synthetic genome, small index" % args.readlen if bacterial else
                       "aligned reads/sec (whole node), %d bp SE vs hg38-like synthetic genome (hg38 unavailable offline), large index" % args.readlen),
            "value": shard.throughput(world, n, steps, dt),
            "unit": "reads/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u32" if off_sz == 4 else "u8/u64", "data": "DRY RUN (--dry-ranks): no device, no alignment; only the N-rank plumbing is real" if dry else "synthetic",
            "config": {
                "workload": ("BASELINE.json configs[1]: E. coli K-12-like synthetic %.2f Mbp genome (one chromosome; 7 rRNA-operon-like and 40 IS-like repeat copies; the real sequence is not available offline), "
                             ".%s index (side %d B, %d-byte offsets) built in this run by the GPU index builder, %d x %d bp SE reads per GPU per step, %s"
                             % (args.genome_mbp, ext, side, off_sz, n, args.readlen, cfg["what"])) if bacterial else
                            "BASELINE.json %s: hg38-like synthetic %d Mbp genome%s (%d chromosomes; ~45 %% repeats: Alu-/L1-like and older diverged families, simple repeats, segmental duplications; N gaps), "
                            ".%s index (side %d B, %d-byte offsets) built in this run by the GPU index builder, %d x %d bp %s per GPU per step, %s"
                            % ({"se150": "configs[2]", "pe-vsens": "configs[3]", "local400": "configs[4]"}.get(args.config, "(extra) " + args.config), args.genome_mbp,
                               " = hg38 scale" if args.genome_mbp >= 3000 else "", N_CHROMS, ext, side, off_sz, n, args.readlen,
                               "reads as %d pairs" % (n // 2) if args.paired else "SE reads", cfg["what"]),
                "config_name": args.config, "command_line": " ".join(cfg["args"]),
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
                                           "sampled_rows": (prof[23] & 0xffffffff) / max(1, prof[9]), "sampled_rows_seen_list": (prof[23] >> 32) / max(1, prof[9])},
                "kernel_ms_per_step": {("k_align_pairs" if args.paired and k == "k_align_reads" else k): round(v, 3) for k, v in kavg.items()}, "batch_ms_events": round(batch_ms, 3),
                "fm_kernels_sides_per_read": cnt.rank_queries / float(n * args.steps),
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
            # the end-to-end leg runs once this process is gone (main()): what it needs travels in the line
            res["e2e_pending"] = {"base": base, "large": large, "fq": e2e_fq, "preset": cfg["args"], "threads": threads, "work": cache_dir(),
                                  "par": {"parity_identical": par.get("parity_identical")} if par else None}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def outer():
    """N = 1: the measurement runs in a child process and the end-to-end leg -- the drop-in binary, FASTQ file -> SAM file, the reference's own
    timed quantity (bt2_search.cpp:4863 "Multiseed full-index search") -- after the child has exited.  Measured (profiles/r05g_*): next to a
    second process that merely holds a context on the same GPU (this script with its torch runtime, idle) the binary runs at 0.55 x its rate
    on a GPU of its own (its persistent worker waves get time-sliced against the other process's queues); a user's run has the GPU to itself."""
    if os.environ.get("BT2_BENCH_CHILD") or int(os.environ.get("WORLD_SIZE", "1")) > 1 or "--gpus" in sys.argv and sys.argv[sys.argv.index("--gpus") + 1] != "1":
        return main()
    env = dict(os.environ, BT2_BENCH_CHILD="1")
    p = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], stdout=subprocess.PIPE, env=env, text=True)
    line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ""
    try:
        res = json.loads(line)
    except ValueError:
        sys.stdout.write(p.stdout)
        raise SystemExit(p.returncode or 1)
    pend = res.pop("e2e_pending", None)
    if pend:
        res["e2e"] = e2e_leg(pend["base"], pend["large"], pend["fq"], pend["preset"], pend["threads"], res["value"], pend["work"], pend["par"])
    print(json.dumps(res), flush=True)
    raise SystemExit(p.returncode)


if __name__ == "__main__":
    outer()

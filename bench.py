#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X multiseed hot path on synthetic 150 bp single-end reads.

Contract (see DESIGN.md "Measurement"):
  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)
A "step" is one pass of the implemented hot-path stages over one resident batch of reads
(inputs already in HBM).  Rank 0 prints ONE JSON line with metric/value plus `roofline`
(dominant kernel, HBM-bound, timed with events on the launch stream) and `cpu_baseline`
(the unmodified reference bowtie2-align-s from oracle/_ref, all host cores, bounded sample).

hg38 is not available offline, so the workload is a deterministic synthetic genome (uniform
random bases + planted repeats) indexed by the reference's own bowtie2-build from oracle/_ref;
`config.workload` names it.  Reads shard across ranks with no data-path collective ("weak").
"""
import argparse
import json
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


def genome_path(mbp, seed):
    d = os.environ.get("BT2_BENCH_CACHE", "/tmp/bt2_amd_bench")
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, "synth_%dmbp_s%d" % (mbp, seed))


def make_genome(mbp, seed):
    """Deterministic synthetic genome: 4 chromosomes of uniform random bases with planted repeats
    (so that multi-element SA ranges and repeat seeds occur) and a few N stretches."""
    import numpy as np
    base = genome_path(mbp, seed)
    fa = base + ".fa"
    npy = base + ".npy"
    if os.path.exists(fa) and os.path.exists(npy):
        return base, np.load(npy, allow_pickle=True)
    rng = np.random.default_rng(seed)
    total = mbp * 1_000_000
    nchr = 4
    chroms = []
    with open(fa + ".tmp", "wb") as f:
        for c in range(nchr):
            n = total // nchr
            a = rng.integers(0, 4, size=n, dtype=np.uint8)
            # planted repeats: copy segments elsewhere (2% of the chromosome, lengths 200-5000)
            nrep = max(1, n // 100_000)
            for _ in range(nrep):
                ln = int(rng.integers(200, 5000))
                src = int(rng.integers(0, n - ln))
                dst = int(rng.integers(0, n - ln))
                a[dst:dst + ln] = a[src:src + ln]
            # N stretches
            for _ in range(max(1, n // 2_000_000)):
                ln = int(rng.integers(10, 2000))
                p = int(rng.integers(0, n - ln))
                a[p:p + ln] = 4
            chroms.append(a)
            s = np.frombuffer(b"ACGTN", dtype=np.uint8)[a]
            f.write((">chr%d\n" % (c + 1)).encode())
            full = n - n % 80
            lines = s[:full].reshape(-1, 80)
            nl = np.full((lines.shape[0], 1), 10, dtype=np.uint8)
            f.write(np.hstack([lines, nl]).tobytes())
            if n % 80:
                f.write(s[full:].tobytes() + b"\n")
    os.replace(fa + ".tmp", fa)
    arr = np.empty(nchr, dtype=object)
    for i, a in enumerate(chroms):
        arr[i] = a
    np.save(npy, arr, allow_pickle=True)
    return base, arr


def build_index(base, threads):
    if os.path.exists(base + ".rev.2.bt2"):
        return
    exe = os.path.join(ROOT, "oracle", "_ref", "bowtie2-build-s")
    if not os.path.exists(exe):
        raise RuntimeError("oracle/_ref/bowtie2-build-s missing: run __graft_entry__.build() where /root/reference exists")
    t0 = time.time()
    # bowtie2-build's blockwise suffix sorter scales poorly past a few dozen threads
    subprocess.check_call([exe, "--threads", str(min(threads, 16)), "-q", base + ".fa", base], stdout=subprocess.DEVNULL)
    log("[bench] index built in %.1fs" % (time.time() - t0))


def synth_reads_gpu(chroms, n, length, seed, device):
    """SURVEY.md 8d generator on the GPU: uniform position, 50/50 strand, 1% substitutions,
    0.1% insertions + 0.1% deletions (at most one indel per read here), Phred from {38,38,38,30,20,12}."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lens = torch.tensor([len(c) for c in chroms], dtype=torch.int64)
    starts = torch.cumsum(lens, 0) - lens
    genome = torch.cat([torch.from_numpy(c) for c in chroms]).to(device)
    ci = torch.randint(0, len(chroms), (n,), generator=g, device=device)
    clen = lens.to(device)[ci]
    pos = (torch.rand(n, generator=g, device=device, dtype=torch.float64) * (clen - length - 2).double()).long()
    gpos = starts.to(device)[ci] + pos
    idx = torch.arange(length, device=device).unsqueeze(0).expand(n, length)
    # one indel per read with probability ~ length * 0.002
    has_indel = torch.rand(n, generator=g, device=device) < length * 0.002
    is_ins = torch.rand(n, generator=g, device=device) < 0.5
    k = torch.randint(5, length - 5, (n,), generator=g, device=device).unsqueeze(1)
    shift = torch.zeros(n, length, dtype=torch.int64, device=device)
    dele = (has_indel & ~is_ins).unsqueeze(1)
    ins = (has_indel & is_ins).unsqueeze(1)
    shift = torch.where(dele & (idx >= k), torch.ones_like(shift), shift)
    shift = torch.where(ins & (idx > k), -torch.ones_like(shift), shift)
    seq = genome[gpos.unsqueeze(1) + idx + shift]
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


def write_fastq(path, seq, qual, n, repeat=1):
    """FASTQ of the first n reads; the same block is written `repeat` times (read names repeat too)."""
    import numpy as np
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[seq[:n].cpu().numpy()]
    q = qual[:n].cpu().numpy()
    parts = []
    for i in range(n):
        parts.append(b"@r%d\n" % i)
        parts.append(s[i].tobytes())
        parts.append(b"\n+\n")
        parts.append(q[i].tobytes())
        parts.append(b"\n")
    block = b"".join(parts)
    with open(path, "wb") as f:
        for _ in range(repeat):
            f.write(block)


def cpu_baseline(base, seq, qual, sample, threads, repeat=1):
    """Reference bowtie2-align-s (oracle/_ref, unmodified v2.5.5, SSE2 build) on a bounded sample."""
    exe = os.path.join(ROOT, "oracle", "_ref", "bowtie2-align-s")
    if not os.path.exists(exe):
        return None
    fq = base + ".bench_sample.fq"
    write_fastq(fq, seq, qual, sample, repeat)
    cmd = [exe, "--sensitive", "-p", str(threads), "--reorder", "-t", "-x", base, "-U", fq, "-S", "/dev/null"]
    t0 = time.time()
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    wall = time.time() - t0
    if p.returncode != 0:
        log("[bench] cpu baseline failed:", p.stderr[-500:])
        return None
    m = re.search(r"Multiseed full-index search: (\d+):(\d+):(\d+)", p.stderr)
    search = wall
    if m:
        s = int(m.group(1)) * 3600 + int(m.group(2)) * 60 + int(m.group(3))
        if s >= 5:          # the -t line has 1 s resolution; only trust it when it is long enough
            search = float(s)
    al = re.search(r"([\d.]+)% overall alignment rate", p.stderr)
    return {"value": sample * repeat / search, "unit": "reads/s", "cores": threads, "kind": "reference",
            "sample": "%d of the same synthetic 150 bp reads x %d passes, bowtie2-align-s v2.5.5 (oracle/_ref, -O3 -msse2) --sensitive -p %d -S /dev/null; "
                      "time = %s; overall alignment rate %s%%" % (sample, repeat, threads, "'-t' search time" if search != wall else "wall incl. index load",
                                                                  al.group(1) if al else "?")}


def pmc_traffic(kernel, reads_per_launch):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE + WRITE_SIZE, KiB),
    scaled from the reads-per-launch of that profile run to this run's.  PMC collection needs rocprofv3 around the
    process, so it cannot be taken inside the timed run; profiles/README.md has the recipe."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        k = d["kernels"][kernel]
        per_read = (k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0 / (d["reads_per_launch"] * d["launches"])
        return int(per_read * reads_per_launch), "profiles/%s (FETCH_SIZE+WRITE_SIZE per read x %d reads)" % (os.path.basename(files[-1]), reads_per_launch)
    except Exception:
        return None, None


def synth_pairs_gpu(chroms, npairs, length, seed, device):
    """Pairs for --paired: fragment length N(300,30) clipped to [length+1, 450], mate 1 = fragment start (forward), mate 2 = reverse
    complement of the fragment end, 1 % substitutions, half of the pairs with the roles of the mates swapped.  Returns [2*npairs, length]."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lens = torch.tensor([len(c) for c in chroms], dtype=torch.int64)
    starts = torch.cumsum(lens, 0) - lens
    genome = torch.cat([torch.from_numpy(c) for c in chroms]).to(device)
    ci = torch.randint(0, len(chroms), (npairs,), generator=g, device=device)
    clen = lens.to(device)[ci]
    frag = (300.0 + 30.0 * torch.randn(npairs, generator=g, device=device)).long().clamp(length + 1, 450)
    pos = (torch.rand(npairs, generator=g, device=device, dtype=torch.float64) * (clen - 460).double()).long()
    gpos = starts.to(device)[ci] + pos
    idx = torch.arange(length, device=device).unsqueeze(0).expand(npairs, length)
    m1 = genome[gpos.unsqueeze(1) + idx]
    m2 = (3 - genome[(gpos + frag - length).unsqueeze(1) + idx].clamp(max=3)).flip(1)
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
    ap.add_argument("--genome-mbp", type=int, default=int(os.environ.get("BT2_BENCH_MBP", "32")))
    ap.add_argument("--reads", type=int, default=int(os.environ.get("BT2_BENCH_READS", "200000")), help="reads per GPU per step")
    ap.add_argument("--readlen", type=int, default=150)
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("BT2_BENCH_CPU_SAMPLE", "200000")))
    ap.add_argument("--cpu-repeat", type=int, default=int(os.environ.get("BT2_BENCH_CPU_REPEAT", "12")),
                    help="passes over the CPU sample, so that the reference runs for ~10 s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--paired", action="store_true",
                    help="measure the paired-end kernel instead: --reads/2 pairs of 2 x --readlen, fragments N(300,30), --fr (not the headline metric)")
    args = ap.parse_args()

    import torch
    import bowtie2_amd as b

    from bowtie2_amd import shard
    rank, local_rank, world = shard.env_rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = shard.init("nccl", dev)      # RCCL; None when WORLD_SIZE == 1

    threads = nproc()
    # ---- workload: index (rank 0 builds, others wait) ----
    if rank == 0:
        base, chroms = make_genome(args.genome_mbp, 2)
        build_index(base, threads)
    if dist is not None:
        dist.barrier()
    if rank != 0:
        base, chroms = make_genome(args.genome_mbp, 2)

    ctx = b.Context(local_rank)
    info = ctx.load_index(base)
    # per-rank shard of reads (weak scaling: fixed reads per GPU)
    seq, qual = synth_reads_gpu(chroms, args.reads, args.readlen, shard.shard_seed(1000, rank), dev)
    n = args.reads
    if args.paired:
        # mates interleaved: read 2i = forward mate at the fragment start, read 2i+1 = reverse-complemented mate at its end
        seq, qual = synth_pairs_gpu(chroms, n // 2, args.readlen, shard.shard_seed(2000, rank), dev)
        n = seq.shape[0]
    off = (torch.arange(n + 1, device=dev, dtype=torch.int64) * args.readlen)
    batch = b.ReadBatch(seq.view(-1), qual.view(-1), off, n)

    # --sensitive, 150 bp: -L 22, -i S,1,1.15 -> interval 1+1.15*sqrt(150) = 15 (bt2_search.cpp:3443-3450)
    import ctypes as C
    import math
    import numpy as np
    L = 22
    interval = max(1, int(1 + 1.15 * math.sqrt(args.readlen)))
    P = b.AlignParams(mm_type=3, mm_max=6, mm_min=2, n_pen=1, rdgapo=8, rdgape=3, rfgapo=8, rfgape=3, gapbar=4, match_bonus=0,
                      khits=1, mhits=50, max_dp_streak=15, max_ug=300, max_dp=300, max_iters=400, n_seed_rounds=2,
                      seed_boost_thresh=300, tighten=3, maxhalf=15, nofw=0, norc=0, do_exact_upfront=1, do_1mm_upfront=1,
                      do_ungapped=1, do_extend=1, large_index=1 if info.off_size == 8 else 0)
    if args.paired:
        # default pair policy: --fr, -I 0 -X 500, mixed + discordant reporting, containment and overlap allowed (bt2_search.cpp:303-502)
        P.paired, P.pe_policy, P.pe_maxfrag, P.pe_minfrag, P.pe_flags, P.max_mate_streak = 1, 3, 500, 0, 2 | 4 | 8 | 32 | 64 | 128, 10
        interval = max(1, int(interval * 1.2 + 0.5))       # both mates pass their filters (bt2_search.cpp:3427-3434)
    # per-read parameters as the host derives them (minsc = (long)(-0.6 + -0.6*len), nceil = 0.15*len; seeds from read content)
    minsc = int(-0.6 + -0.6 * args.readlen)
    rp = np.zeros(n, dtype=[("minsc", "<i4"), ("interval", "<i4"), ("nceil", "<i4"), ("seedlen", "<i4"), ("seed", "<u4"), ("filt", "<u4")])
    rp["minsc"] = minsc; rp["interval"] = interval; rp["nceil"] = int(0.15 * args.readlen); rp["seedlen"] = L; rp["filt"] = 15
    rp["seed"] = np.random.default_rng(7 + rank).integers(0, 2**32, size=n, dtype=np.uint32)   # stands in for genRandSeed(name,seq,qual)
    rp_t = torch.from_numpy(rp.view(np.uint8).copy()).to(dev)

    ev = lambda: torch.cuda.Event(enable_timing=True)
    stage_events = []
    last = {}

    kern_times = []

    def step(record):
        e0, e1 = ev(), ev()
        e0.record()
        res, stride = ctx.align_batch(batch, rp_t, P, args.readlen)
        e1.record()
        if record:
            stage_events.append((e0, e1))
            kern_times.append(ctx.align_timing())      # HIP events recorded by the library around each kernel
        last["res"], last["stride"] = res, stride

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

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
    batch_ms = sum(a.elapsed_time(bb) for a, bb in stage_events) / len(stage_events)
    kavg = {k: sum(t[k] for t in kern_times) / len(kern_times) for k in kern_times[0]}
    kern_ms = kavg["k_align_reads"]
    prof = ctx.align_profile()
    cnt = ctx.counters()

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
        sides_per_launch = prof[8] / float(args.steps)
        dp_cells = float(h["n_ex_dps"].sum()) * args.readlen * (args.readlen + 61)
        # Algorithmic bytes.  k_align_reads, the dominant kernel: the rank queries it still issues itself
        # (offset resolution, re-seeding rounds) * side_sz + DP reference windows + reads in + result records out.
        alg_bytes = sides_per_launch * side + h["n_ex_dps"].sum() * ((args.readlen + 61 + 3) // 4) + n * args.readlen * 2 + n * stride
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        # The four lane-per-task FM kernels in front of it carry most of the rank queries of the path.
        fm_ms = sum(v for k, v in kavg.items() if k != "k_align_reads")
        fm_bytes = (cnt.rank_queries * side + cnt.ftab_lookups * 2 * off_sz + cnt.sa_lookups * off_sz) / float(args.steps) + n * args.readlen * 2
        fm_achieved = fm_bytes / (fm_ms * 1e-3) / 1e9 if fm_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic("k_align_reads", n)
        res = {
            "metric": ("aligned reads/sec (whole node), 2 x 150 bp PE (mates counted as reads), synthetic genome" if args.paired else
                       "aligned reads/sec (whole node), 150 bp SE, synthetic genome (hg38 unavailable offline)"),
            "value": shard.throughput(world, n, steps, dt),
            "unit": "reads/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u32" if off_sz == 4 else "u8/u64", "data": "synthetic",
            "config": {
                "workload": "synthetic %d Mbp genome (.bt2, built by reference bowtie2-build), %d x %d bp SE reads per GPU per step, --sensitive (-D 15 -R 2 -N 0 -L 22 -i S,1,1.15), end-to-end"
                            % (args.genome_mbp, n, args.readlen),
                "stages_timed": "one bt2g_align_batch per step = the whole per-read worker: k_exact_sweep, k_one_mm, k_seed_search_exact, k_extend_hits (lane-per-task FM kernels) then k_align_reads "
                                "(rank+prioritise, offset resolution, re-seeding, SW fill + backtrace, -M reporting)",
                "not_in_timed_region": "FASTQ parse and SAM text formatting (host side, SURVEY.md 8f)",
                "fraction_aligned": all_aligned / float(world * n),
                "index_bytes_hbm": int(info.hbm_bytes), "side_sz": int(side),
                "bw_ops_per_read": rankq / n, "dp_fills_per_read": float(h["n_ex_dps"].sum()) / n,
                "backtraces_per_read": float(h["n_bt_attempts"].sum()) / n,
                "reads_overflowed": int((h["status"] != 0).sum()),
                "worker_phase_us_per_read": dict(zip(["sweep", "mm1", "seeds", "rank_prioritise", "resolve", "dp_fill", "backtrace", "whole_read", "gather_cells", "report", "ungapped",
                                                      "bt_tile_fetch", "gather_lastrow", "gather_zero_masks", "red_overlap", "red_add", "sink_report"],
                                                     [round(prof[i] / 100.0 / max(1, prof[9]), 1) for i in (0, 1, 2, 3, 4, 5, 6, 7, 10, 11, 12, 16, 17, 18, 19, 20, 21)])),
                "worker_counts_per_read": {"bt_steps": prof[13] / max(1, prof[9]), "bt_tiles": prof[14] / max(1, prof[9]), "cand_cells": prof[15] / max(1, prof[9])},
                "kernel_ms_per_step": {k: round(v, 3) for k, v in kavg.items()}, "batch_ms_events": round(batch_ms, 3),
                "fm_kernels_sides_per_read": cnt.rank_queries / float(n * args.steps),
                "sides_per_read": prof[8] / max(1, prof[9]),
            },
            "roofline": {"bound": "hbm", "kernel": "k_align_reads", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": kern_ms,
                         "dp_gcups": dp_cells / (kern_ms * 1e-3) / 1e9,
                         "note": "k_align_reads is bound by scalar-instruction issue of its wave-uniform control code, not by HBM (DESIGN.md 6); "
                                 "the FM kernels below are the HBM-shaped part of the path",
                         "fm_kernels": {"kernels": ["k_exact_sweep", "k_one_mm", "k_seed_search_exact", "k_extend_hits"], "bound": "hbm",
                                        "ms_per_launch_sum": fm_ms, "algorithmic_bytes_per_launch": int(fm_bytes),
                                        "achieved": fm_achieved, "unit": "GB/s", "frac": fm_achieved / HBM_PEAK_GBS}},
        }
        cb = None
        if not args.no_cpu_baseline and world == 1 and not args.paired:      # reported at N=1 only (the paired mode is a kernel study, no CPU leg)
            cb = cpu_baseline(base, seq, qual, min(args.cpu_sample, n), threads, args.cpu_repeat)
        res["cpu_baseline"] = cb
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

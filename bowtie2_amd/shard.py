"""Read sharding across GPUs: the unit is one read, there is no cross-read state
(`msNoCache`, bt2_search.cpp:430; the RNG is seeded per read, pat.cpp:45-84), so ranks take disjoint
blocks of reads, every rank holds the whole index, and the data path has no collective.  The only
exchanges are (i) a max-reduction of the timed region, (ii) sum-reductions of the summary counters
(the reference's ReportingMetrics merge, aln_sink.cpp:33-101) and (iii) gathering result records to
rank 0 in read order.  One process per GPU; backend "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests.
"""
import os


def env_rank():
    """(rank, local_rank, world) as torch.distributed.run exports them."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend, device=None):
    """Join the process group if WORLD_SIZE > 1; returns the torch.distributed module or None."""
    rank, _, world = env_rank()
    if world <= 1 and not (os.environ.get("BT2G_DIST_WORLD1") == "1" and "MASTER_ADDR" in os.environ):
        return None      # (BT2G_DIST_WORLD1=1 under torch.distributed.run: a process group of one rank -- lets a single-GPU box run the N-GPU code path, RCCL included)
    import torch.distributed as dist
    if not dist.is_initialized():
        kw = {}
        if device is not None and backend == "nccl":
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


BLOCK = 1 << 16   # reads per dealt block: big enough to fill the persistent waves, small enough to balance tails


def blocks_of(n_reads, rank, world, block=BLOCK):
    """Strong-scaling partition of reads [0, n_reads): contiguous blocks dealt round-robin to the ranks
    (equivalent to the reference's -s/-u windows).  Returns [(begin, end), ...] for this rank.
    Paired batches hold the mates interleaved (read 2i, 2i+1): with an even `block` (the default is) a pair is never split."""
    out = []
    nblocks = (n_reads + block - 1) // block
    for b in range(rank, nblocks, world):
        out.append((b * block, min(n_reads, (b + 1) * block)))
    return out


def shard_seed(base_seed, rank):
    """Weak-scaling synthetic shards: every rank generates its own reads from a distinct seed."""
    return base_seed + rank


def reduce_max(dist, value, device):
    import torch
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(dist, values, device):
    """Element-wise sum of a list of counters over the ranks (the stderr summary / metrics merge)."""
    import torch
    if dist is None:
        return [int(v) for v in values]
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


def throughput(world, reads_per_rank_per_step, steps, max_seconds):
    """Whole-job reads/s: what all ranks processed over the slowest rank's timed region."""
    return world * reads_per_rank_per_step * steps / max_seconds


def gather_in_read_order(dist, my_blocks, my_records, n_reads, record_bytes, device):
    """Gather fixed-size result records (uint8 tensor [n_mine, record_bytes], rows in the order of
    `my_blocks`) to rank 0 and place them at their read index.  Returns the [n_reads, record_bytes]
    tensor on rank 0, None elsewhere.  Host-side merge; xGMI bandwidth is irrelevant at ~1.3 KB/read."""
    import torch
    rank, _, world = env_rank()
    if dist is None:
        return my_records
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((my_blocks, my_records.cpu()), gathered, dst=0)
    if rank != 0:
        return None
    out = torch.zeros((n_reads, record_bytes), dtype=torch.uint8)
    for blocks, recs in gathered:
        pos = 0
        for b, e in blocks:
            out[b:e] = recs[pos:pos + (e - b)]
            pos += e - b
    return out


def gather_packed(dist, packed, nbytes, device):
    """The per-step merge of the N-GPU path: every rank's packed result records (bt2g_results_pack: a uint8 tensor in HBM, the
    first `nbytes` bytes valid) gathered to rank 0 over the process group (RCCL over xGMI on GPUs).  Returns the list of
    per-rank tensors on rank 0, None elsewhere.  dist.gather wants equal shapes: pieces are padded to the largest."""
    import torch
    rank, _, world = env_rank()
    if dist is None:
        return [packed[:nbytes]]
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([nbytes], dtype=torch.int64, device=device))
    sizes = [int(s.item()) for s in sizes]
    width = max(sizes)
    if packed.numel() < width:
        packed = torch.cat([packed, torch.zeros(width - packed.numel(), dtype=torch.uint8, device=device)])
    got = [torch.empty(width, dtype=torch.uint8, device=device) for _ in range(world)] if rank == 0 else None
    dist.gather(packed[:width], got, dst=0)
    if rank != 0:
        return None
    return [g[:sz] for g, sz in zip(got, sizes)]

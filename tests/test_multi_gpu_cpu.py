"""The N>1 path of bench.py / the sharded driver, on CPU with gloo and world_size 2: rendezvous on
127.0.0.1, disjoint read blocks that cover the input, per-rank seeds, max-over-ranks timing, summed
counters, and result records gathered to rank 0 in read order.  No kernels run here."""
import json
import os
import socket
import subprocess
import sys

from bt2test import build_hostsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["BT2_ROOT"])
import torch
from bowtie2_amd import shard
rank, local_rank, world = shard.env_rank()
dist = shard.init("gloo")
dev = torch.device("cpu")
n_reads, rec = 300000, 16
blocks = shard.blocks_of(n_reads, rank, world, block=4096)
mine = sum(e - b for b, e in blocks)
# fake "alignment": the record of read i is i written into 8 bytes, plus the rank that produced it
recs = torch.zeros((mine, rec), dtype=torch.uint8)
pos = 0
for b, e in blocks:
    idx = torch.arange(b, e, dtype=torch.int64)
    recs[pos:pos + (e - b), :8] = idx.view(-1, 1).view(torch.uint8).view(-1, 8)
    recs[pos:pos + (e - b), 8] = rank
    pos += e - b
dt = shard.reduce_max(dist, 1.0 + rank, dev)               # rank 1 is the slow one
aligned, total = shard.reduce_sum(dist, [mine - rank, mine], dev)
allrec = shard.gather_in_read_order(dist, blocks, recs, n_reads, rec, dev)
# per-step merge of packed records (ragged sizes): rank r sends 1000 + 37 * r bytes of value r + 1
pk = shard.gather_packed(dist, torch.full((5000,), rank + 1, dtype=torch.uint8), 1000 + 37 * rank, dev)
out = {"rank": rank, "world": world, "dt": dt, "aligned": aligned, "total": total, "mine": mine,
       "seed": shard.shard_seed(1000, rank), "value": shard.throughput(world, 1000, 5, dt)}
if rank == 0:
    ids = allrec[:, :8].contiguous().view(torch.int64).view(-1)
    out["in_order"] = bool((ids == torch.arange(n_reads)).all())
    out["from_both"] = sorted(set(allrec[:, 8].tolist()))
    out["packed"] = [[int(t.numel()), int(t.min()), int(t.max())] for t in pk]
dist.barrier()
print("RESULT " + json.dumps(out), flush=True)
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_blocks_partition():
    sys.path.insert(0, ROOT)
    from bowtie2_amd import shard
    for n, world, block in ((0, 2, 8), (1, 2, 8), (17, 2, 8), (64, 4, 8), (1000, 8, 7), (300000, 8, shard.BLOCK)):
        seen = []
        for r in range(world):
            for b, e in shard.blocks_of(n, r, world, block):
                assert 0 <= b < e <= n
                seen.extend(range(b, e))
        assert sorted(seen) == list(range(n))           # disjoint and complete
    sizes = [sum(e - b for b, e in shard.blocks_of(1 << 20, r, 8)) for r in range(8)]
    assert max(sizes) - min(sizes) <= shard.BLOCK       # balanced to one block


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), BT2_ROOT=ROOT, GLOO_SOCKET_IFNAME="lo")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, se[-2000:]
        outs.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][0][7:]))
    outs.sort(key=lambda o: o["rank"])
    assert [o["world"] for o in outs] == [2, 2]
    assert outs[0]["dt"] == outs[1]["dt"] == 2.0                      # max over ranks
    assert outs[0]["total"] == outs[1]["total"] == 300000             # shards cover the input
    assert outs[0]["aligned"] == 300000 - 1                           # summed counters
    assert outs[0]["mine"] + outs[1]["mine"] == 300000
    assert outs[0]["seed"] != outs[1]["seed"]
    assert outs[0]["value"] == 2 * 1000 * 5 / 2.0                     # whole-job throughput over the slowest rank
    assert outs[0]["in_order"] and outs[0]["from_both"] == [0, 1]
    assert outs[0]["packed"] == [[1000, 1, 1], [1037, 2, 2]]


def test_blocks_keep_mates_together():
    """Paired batches are interleaved (mate 1 at even, mate 2 at odd read indexes): every dealt block starts and ends on a pair."""
    from bowtie2_amd import shard
    n = 2 * 123457
    seen = 0
    for world in (1, 2, 3, 8):
        cover = []
        for rank in range(world):
            for b, e in shard.blocks_of(n, rank, world):
                assert b % 2 == 0 and e % 2 == 0
                cover.append((b, e))
        cover.sort()
        assert cover[0][0] == 0 and cover[-1][1] == n and all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1))
        seen += 1
    assert seen == 4


def test_sharded_driver_two_ranks_sam_identical(tmp_path):
    """bowtie2_amd.mgpu (the N-GPU driver: the product executable per rank with --shard r/N, SAM pieces gathered and
    counters all-reduced over the process group) with world size 2 on gloo.  The engine here is the CPU twin of the
    worker (tests/hostsim, test-only, same command line and reader); the merged SAM and the merged summary must be
    byte-identical to the reference's golden SAM / the 1-rank run."""
    gold = os.path.join(ROOT, "tests", "golden")
    hs = os.path.join(ROOT, "tests", "hostsim", "hostsim")
    build_hostsim(hs)
    common = ["--sensitive", "--batch", "64", "-x", os.path.join(gold, "tiny_s"), "-U", os.path.join(gold, "align_reads.fq")]
    one = subprocess.run([hs] + common, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True)
    port = free_port()
    out = tmp_path / "merged.sam"
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo", PYTHONPATH=ROOT)
        procs.append(subprocess.Popen([sys.executable, "-m", "bowtie2_amd.mgpu", "--engine", hs, "--backend", "gloo", "--sharding", "blocks", "--"] + common + ["-S", str(out)],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    errs = []
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-2000:]
        errs.append(se)
    got = [l for l in open(out).read().splitlines() if not l.startswith("@PG")]
    want = open(os.path.join(gold, "align_golden_s_sens.sam")).read().splitlines()
    assert got == want
    assert got == [l for l in one.stdout.splitlines() if not l.startswith("@PG")]
    # the merged summary (rank 0) equals the single-process summary
    assert errs[0].strip().splitlines()[-6:] == one.stderr.strip().splitlines()[-6:]

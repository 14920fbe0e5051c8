"""bench.py as the driver's scaling run launches it -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1
--master-port P bench.py --gpus 8 --steps K --warmup W` -- with --dry-ranks: eight ranks, gloo instead of RCCL, a stand-in for the device
(VERDICT r4 item 7).  What runs is the script's own N-GPU plumbing: the rendezvous, rank 0 filling the index cache while the others wait at the
barrier (and every rank checking that it can see the files afterwards), per-rank read shards with per-rank seeds, the step loop with its
per-step gather of packed result records to rank 0, barrier + max-over-ranks timing, the summed counters, and rank 0's single JSON line.
Nothing in it is a measurement (the line's `data` says so).  Also: the host threads bowtie2_amd.mgpu gives each rank's executable."""
import json
import os
import shutil
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_dry(tmp_path, world, extra=()):
    cache = str(tmp_path / "cache")
    env = dict(os.environ, BT2_BENCH_CACHE=cache, GLOO_SOCKET_IFNAME="lo", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--dry-ranks", "--reads", "3000"] + list(extra)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0: %r" % p.stdout[-500:]
    return json.loads(lines[0]), p.stderr, cache


def test_eight_ranks_through_bench_py(tmp_path):
    d, err, cache = run_dry(tmp_path, 8)
    # (round 6: two steps in flight on the N-GPU path too -- a batch's records are packed and gathered after the next batch has been issued -- and
    # one untimed step per stream: --warmup 1 is raised to 2)
    assert d["n_gpus"] == 8 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak" and d["config"]["steps_in_flight"] == 2
    assert d["data"].startswith("DRY RUN")
    # whole-job aggregate over the slowest rank's timed region
    assert abs(d["value"] - 8 * 3000 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
    # the per-step merge: eight ranks' packed records (152 bytes per read in the stand-in) arrived on rank 0
    assert "%d bytes arrived on rank 0" % (8 * 3000 * 152) in d["config"]["n_gpu_merge"]
    assert d["config"]["fraction_aligned"] == 1.0          # summed over the ranks, divided by world * reads
    assert d["cpu_baseline"] is None and "e2e" not in d     # N = 1 only
    # rank 0 built the index once; everybody loaded it after the barrier
    assert err.count("index loaded into HBM") == 8
    assert os.path.exists(os.path.join(cache, "hg38like_1mbp_s2_bt2l_dry.rev.2.bt2l"))
    # a second job finds the cache filled and builds nothing
    d2, err2, _ = run_dry(tmp_path, 8)
    assert d2["config"]["index_build"] is None and d["config"]["index_build"] is not None
    shutil.rmtree(cache, ignore_errors=True)


def test_n_gpu_e2e_leg_through_mgpu(tmp_path):
    """The N-GPU end-to-end leg: every rank writes its distinct reads to its place in ONE FASTQ file, the measuring processes exit, rank 0's
    parent launches the product's N-GPU driver (python -m torch.distributed.run -m bowtie2_amd.mgpu: byte-range sharding, pieces merged in rank order) -- here
    around bench.py's stand-in engine over gloo -- and the line carries `e2e` with the ranks' reads, the slowest rank's search time and the merged file's checks."""
    d, err, cache = run_dry(tmp_path, 3, ["--e2e-reads", "6000"])
    e = d["e2e"]
    assert "error" not in e, e
    assert e["n_gpus"] == 3 and e["reads"] == 3 * 6000 and e["reads_aligned_by_the_ranks"] == 3 * 6000
    assert len(e["search_s_per_rank"]) == 3 and e["reads_per_s_after_load"] > 0 and e["frac_of_resident"] > 0
    assert e["sam_records"] == 3 * 6000 and e["sam_complete_and_in_input_order"] is True
    assert e["host_threads_per_rank"] >= 1 and "bowtie2_amd.mgpu" in e["command"]
    assert "3 ranks x 6000 distinct reads written into one FASTQ file" in err
    assert not [f for f in os.listdir(cache) if f.startswith("sync_")]       # the parents' meeting place is gone
    shutil.rmtree(cache, ignore_errors=True)


def test_real_reference_hook(tmp_path):
    """--genome-fasta / BT2_BENCH_HG38: a real FASTA on the box replaces the synthetic genome -- parsed, indexed (here: the dry stand-in), reads
    sampled from it by the same generator -- and the metric says which genome ran.  On the reference's own example genome."""
    fa = os.path.join(ROOT, "tests", "golden", "example", "lambda_virus.fa")
    d, err, cache = run_dry(tmp_path, 1, ["--genome-fasta", fa])
    assert "REAL reference lambda_virus.fa (1 sequences, 0.0 Mbp incl. N)" in d["metric"] or "REAL reference lambda_virus.fa" in d["metric"]
    assert d["config"]["genome"].startswith("lambda_virus.fa") and "REAL reference lambda_virus.fa" in d["config"]["workload"]
    assert "read from lambda_virus.fa" in err
    assert any(f.startswith("real_lambda_virus_fa_") for f in os.listdir(cache))
    # the environment hook, and the synthetic wording without it
    env_cache = str(tmp_path / "c2")
    env = dict(os.environ, BT2_BENCH_CACHE=env_cache, BT2_BENCH_HG38=fa, OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-ranks", "--steps", "2", "--warmup", "1", "--reads", "2000"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, env=env, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d2 = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert "REAL reference lambda_virus.fa" in d2["metric"]
    d3, _, _ = run_dry(tmp_path, 1)
    assert "hg38-like synthetic genome" in d3["metric"] and d3["config"]["genome"] == "synthetic"
    assert "timed region: 3 x the same resident batch of 3000 reads per GPU" in d3["config"]["workload"]
    shutil.rmtree(cache, ignore_errors=True); shutil.rmtree(env_cache, ignore_errors=True)


def test_pairs_config_two_ranks(tmp_path):
    d, _, cache = run_dry(tmp_path, 2, ["--config", "pe-vsens"])
    assert d["n_gpus"] == 2 and d["config"]["config_name"] == "pe-vsens" and d["config"]["steps_in_flight"] == 3      # the N-GPU path pipelines like the 1-GPU one since round 6 (records gathered one batch behind)
    shutil.rmtree(cache, ignore_errors=True)


def test_host_threads_per_rank():
    """bowtie2_amd.mgpu caps the host threads of each rank's executable at its share of the node's usable cores (not the node's core count)."""
    sys.path.insert(0, ROOT)
    from bowtie2_amd import mgpu
    cores = mgpu.usable_cores()
    assert 1 <= cores <= (os.cpu_count() or 1)
    src = open(os.path.join(ROOT, "bowtie2_amd", "mgpu.py")).read()
    assert "LOCAL_WORLD_SIZE" in src and 'usable_cores() // max(1, local_world)' in src
    for lw in (1, 2, 8, 64):
        assert max(1, cores // lw) * min(lw, cores) <= max(cores, lw)       # N ranks never ask for more threads than the node has cores (beyond one each)

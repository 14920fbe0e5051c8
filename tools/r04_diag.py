"""Diagnostic: the slowest reads / pairs of a batch with their work counters (needs build/variants/libbt2g_diag.so: -DBT2G_DIAG_TICKS puts per-read
device ticks into the record's n_ext_left / n_ext_right / n_resolve_steps / n_sides fields).  usage: BT2G_LIB=... python tools/r04_diag.py <config> [reads]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
import bowtie2_amd as b
cfgname = sys.argv[1]; cfg = B.CONFIGS[cfgname]
n = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["reads"]
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
base = os.path.join(B.cache_dir(), "hg38like_3100mbp_s2_bt2l")
G, cl = B.synth_genome_gpu(3100, 2, dev)
if not os.path.exists(base + ".rev.2.bt2l"):
    torch.cuda.empty_cache(); B.build_index_gpu(base, G, cl, True, 0)
ctx = b.Context(0); info = ctx.load_index(base)
L = cfg["readlen"]
if cfg["paired"]:
    seq, qual = B.synth_pairs_gpu(G, n // 2, L, 2000, dev)
else:
    seq, qual = B.synth_reads_gpu(G, n, L, 1000, dev)
n = seq.shape[0]; del G
names = B.read_names(0, n); names_t = torch.from_numpy(names).to(dev)
off = torch.arange(n + 1, device=dev, dtype=torch.int64) * L
batch = b.ReadBatch(seq.view(-1), qual.view(-1), off, n)
cli = list(cfg["args"]) + (["-1", "a", "-2", "b"] if cfg["paired"] else ["-U", "a"])
P, rp1 = b.cli_params(cli, L, large_index=True, both_mates_pass=cfg["paired"])
P.max_seeds = 1 + max(0, L - rp1.seedlen) // rp1.interval
P.profile = 1
rp = np.zeros(n, dtype=[("minsc", "<i4"), ("interval", "<i4"), ("nceil", "<i4"), ("seedlen", "<i4"), ("seed", "<u4"), ("filt", "<u4")])
rp["minsc"] = rp1.minsc; rp["interval"] = rp1.interval; rp["nceil"] = rp1.nceil; rp["seedlen"] = rp1.seedlen; rp["filt"] = rp1.filt
rp["seed"] = B.gen_rand_seeds(seq, qual, names_t).cpu().numpy().astype(np.uint32)
rp_t = torch.from_numpy(rp.view(np.uint8).copy()).to(dev)
res, stride = ctx.align_batch(batch, rp_t, P, L); torch.cuda.synchronize()
rec = res.view(n, stride)[:, :C.sizeof(b.ReadResult) - C.sizeof(b.Aln)].cpu().numpy()
hdr = np.dtype([("status", "u1"), ("aligned", "u1"), ("maxed", "u1"), ("filt", "u1"), ("exhausted", "u1"), ("has_secbest", "u1"), ("pad", "u1", 2), ("secbest", "<i4"), ("best", "<i4"), ("nalns", "<u4"), ("nreport", "<u4"),
                ("n_ex_iters", "<u4"), ("n_ex_dps", "<u4"), ("n_ex_ugs", "<u4"), ("n_dp_fail_streak_max", "<u4"), ("n_bwops_seed", "<u4"), ("n_bwops_ext", "<u4"), ("n_redundants", "<u4"), ("n_bt_attempts", "<u4"),
                ("t_whole", "<u4"), ("t_dp", "<u4"), ("t_bt", "<u4"), ("t_prio", "<u4"), ("pair_best", "<i4"), ("pair_secbest", "<i4"), ("n_mate_dps", "<u4"), ("pad2", "<u4")])
h = np.frombuffer(rec.tobytes(), dtype=hdr)
step = 2 if cfg["paired"] else 1
t = h["t_whole"][::step].astype(np.float64) / 100.0
print(cfgname, "units", len(t), "mean us", t.mean(), "median", np.median(t), "p99", np.percentile(t, 99), "p99.9", np.percentile(t, 99.9), "max", t.max(), "sum of top 0.1% / total", np.sort(t)[-len(t) // 1000:].sum() / t.sum())
srt = np.sort(t)[::-1]; cs = np.cumsum(srt) / t.sum()
for f in (0.0001, 0.001, 0.01, 0.05, 0.1, 0.25, 0.5):
    print("  slowest %.2f%% of units: %.1f%% of the time (threshold %.0f us)" % (100 * f, 100 * cs[max(0, int(len(t) * f) - 1)], srt[max(0, int(len(t) * f) - 1)]))
for lo, hi in ((0, 200), (200, 400), (400, 800), (800, 1600), (1600, 3200), (3200, 1e9)):
    m = (t >= lo) & (t < hi); hh = h[::step][m]
    print("  %5.0f-%-6.0f us: %5.1f%% of units, %5.1f%% of time | mean dps %.1f bt %.1f iters %.1f dp_us %.0f bt_us %.0f prio_us %.0f" % (lo, hi, 100.0 * m.mean(), 100.0 * t[m].sum() / t.sum(), hh["n_ex_dps"].mean() if m.any() else 0, hh["n_bt_attempts"].mean() if m.any() else 0, hh["n_ex_iters"].mean() if m.any() else 0, hh["t_dp"].mean() / 100.0 if m.any() else 0, hh["t_bt"].mean() / 100.0 if m.any() else 0, hh["t_prio"].mean() / 100.0 if m.any() else 0))
order = np.argsort(-t)[:12]
for i in order:
    r = h[i * step]
    print("unit", i, "us", t[i], "dp_us", r["t_dp"] / 100.0, "bt_us", r["t_bt"] / 100.0, "prio_us", r["t_prio"] / 100.0, "iters", r["n_ex_iters"], "dps", r["n_ex_dps"], "ugs", r["n_ex_ugs"], "bt", r["n_bt_attempts"], "red", r["n_redundants"], "nalns", r["nalns"], "mate_dps", r["n_mate_dps"],
          ("| mate2: oppmate_us %.0f sweepphase %.0f mm1phase %.0f seeds %.0f" % (h[i * step + 1]["t_whole"] / 100.0, h[i * step + 1]["t_dp"] / 100.0, h[i * step + 1]["t_bt"] / 100.0, h[i * step + 1]["t_prio"] / 100.0)) if step == 2 else "")

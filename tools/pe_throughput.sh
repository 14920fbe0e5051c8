#!/bin/bash
# end-to-end paired-end throughput of the drop-in binary vs the reference on the bench genome (default 1 M pairs, 2 x 150 bp)
NP=${1:-1000000}
mkdir -p gpurun_out/pe
python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pe/bench.json 2> gpurun_out/pe/bench.err
B=/tmp/bt2_amd_bench/synth_32mbp_s2
python3 - "$B" "$NP" <<'PY'
import sys, numpy as np
base, n = sys.argv[1], int(sys.argv[2])
seqs = []
cur = []
for line in open(base + ".fa"):
    if line.startswith(">"):
        if cur: seqs.append("".join(cur)); cur = []
    else: cur.append(line.strip())
if cur: seqs.append("".join(cur))
g = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
bounds = np.cumsum([0] + [len(s) for s in seqs])
rng = np.random.default_rng(3)
comp = np.zeros(256, dtype=np.uint8); comp[:] = ord('N')
for a, b in zip(b"ACGT", b"TGCA"): comp[a] = b
L = 150
frag = np.clip(rng.normal(300, 30, n).astype(np.int64), L + 1, 450)
chrom = rng.integers(0, len(seqs), n)
start = np.array([rng.integers(bounds[c], bounds[c + 1] - 460) for c in chrom]) if len(seqs) < 64 else None
qual = np.frombuffer(b"GGG?5-", dtype=np.uint8)
with open("/tmp/pe_1.fq", "wb") as f1, open("/tmp/pe_2.fq", "wb") as f2:
    for i in range(n):
        s = int(start[i]); fr = int(frag[i])
        m1 = g[s:s + L].copy(); m2 = comp[g[s + fr - L:s + fr]][::-1].copy()
        for m in (m1, m2):
            k = rng.random(L) < 0.01
            m[k] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(k.sum()))
        if rng.random() < 0.5: m1, m2 = m2, m1
        q = qual[rng.integers(0, 6, L)].tobytes()
        f1.write(b"@p%d/1\n" % i + m1.tobytes() + b"\n+\n" + q + b"\n")
        f2.write(b"@p%d/2\n" % i + m2.tobytes() + b"\n+\n" + q + b"\n")
PY
s=$(date +%s.%N)
bowtie2_amd/bin/bowtie2-align-s --sensitive -p 16 -t -x $B -1 /tmp/pe_1.fq -2 /tmp/pe_2.fq -S /tmp/pe_ours.sam 2> gpurun_out/pe/ours.err
e=$(date +%s.%N)
awk -v n=$NP -v s=$s -v e=$e 'BEGIN{printf "ours -p 16: %d pairs in %.2f s wall -> %.0f pairs/s (%.0f reads/s) end to end\n", n, e-s, n/(e-s), 2*n/(e-s)}'
grep "bt2g\|overall\|concordantly" gpurun_out/pe/ours.err | head -8
if [ -x oracle/_ref/bowtie2-align-s ]; then
  head -n $(( 4 * 200000 )) /tmp/pe_1.fq > /tmp/pe_1s.fq; head -n $(( 4 * 200000 )) /tmp/pe_2.fq > /tmp/pe_2s.fq
  s=$(date +%s.%N)
  oracle/_ref/bowtie2-align-s --sensitive -p 16 --reorder -x $B -1 /tmp/pe_1s.fq -2 /tmp/pe_2s.fq -S /tmp/pe_ref.sam 2> gpurun_out/pe/ref.err
  e=$(date +%s.%N)
  awk -v s=$s -v e=$e 'BEGIN{printf "reference -p 16 --reorder: 200000 pairs in %.2f s wall -> %.0f pairs/s\n", e-s, 200000/(e-s)}'
  grep -v "^@" /tmp/pe_ours.sam | head -n 400000 > /tmp/a.sam; grep -v "^@" /tmp/pe_ref.sam > /tmp/b.sam
  if cmp -s /tmp/a.sam /tmp/b.sam; then echo "SAM identical on the first 200000 pairs"; else echo "SAM DIFFERS"; diff /tmp/a.sam /tmp/b.sam | head -4 | cut -c1-200; fi
fi

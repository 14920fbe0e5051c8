#!/bin/bash
# steady-state end-to-end throughput of the drop-in binary on a larger input (default: 9.6 M reads, 3 GB FASTQ)
mkdir -p gpurun_out/cli
python bench.py --steps 1 --warmup 1 --cpu-repeat 12 > gpurun_out/cli/bench.json 2> gpurun_out/cli/bench.err
B=/tmp/bt2_amd_bench/synth_32mbp_s2
FQ=/tmp/big.fq
cat $B.bench_sample.fq $B.bench_sample.fq $B.bench_sample.fq $B.bench_sample.fq > $FQ
N=$(( $(wc -l < $FQ) / 4 ))
for p in 8 16; do
  s=$(date +%s.%N)
  bowtie2_amd/bin/bowtie2-align-s --sensitive -p $p -t -x $B -U $FQ -S /tmp/ours_big.sam 2> gpurun_out/cli/ours_big_p$p.err
  e=$(date +%s.%N)
  awk -v n=$N -v s=$s -v e=$e -v p=$p 'BEGIN{printf "ours -p %d: %d reads in %.2f s wall -> %.0f reads/s end to end\n", p, n, e-s, n/(e-s)}'
  grep "bt2g" gpurun_out/cli/ours_big_p$p.err
done
s=$(date +%s.%N)
bowtie2_amd/bin/bowtie2-align-s --sensitive -p 16 -t -x $B -U $FQ -S /dev/null 2> gpurun_out/cli/ours_big_null.err
e=$(date +%s.%N)
awk -v n=$N -v s=$s -v e=$e 'BEGIN{printf "ours -p 16 -S /dev/null: %d reads in %.2f s wall -> %.0f reads/s\n", n, e-s, n/(e-s)}'

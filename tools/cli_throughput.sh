#!/bin/bash
# end-to-end throughput of the drop-in binary (FASTQ in -> SAM out) next to the reference binary
mkdir -p gpurun_out/cli
python bench.py --steps 1 --warmup 1 --cpu-repeat 12 > gpurun_out/cli/bench.json 2> gpurun_out/cli/bench.err   # also leaves the FASTQ sample + index in the cache
B=/tmp/bt2_amd_bench/synth_32mbp_s2
FQ=$B.bench_sample.fq
ls -la $FQ | awk '{print $5, $9}'
N=$(( $(wc -l < $FQ) / 4 ))
for p in 1 4 16; do
  s=$(date +%s.%N)
  bowtie2_amd/bin/bowtie2-align-s --sensitive -p $p -t -x $B -U $FQ -S /tmp/ours_p$p.sam 2> gpurun_out/cli/ours_p$p.err
  e=$(date +%s.%N)
  awk -v n=$N -v s=$s -v e=$e -v p=$p 'BEGIN{printf "ours -p %d: %d reads in %.2f s wall -> %.0f reads/s end to end\n", p, n, e-s, n/(e-s)}'
  grep 'device search' gpurun_out/cli/ours_p$p.err
done
s=$(date +%s.%N)
oracle/_ref/bowtie2-align-s --sensitive -p 16 --reorder -t -x $B -U $FQ -S /tmp/ref.sam 2> gpurun_out/cli/ref.err
e=$(date +%s.%N)
awk -v n=$N -v s=$s -v e=$e 'BEGIN{printf "reference -p 16: %d reads in %.2f s wall -> %.0f reads/s end to end\n", n, e-s, n/(e-s)}'
cmp <(grep -v '^@PG' /tmp/ours_p16.sam) <(grep -v '^@PG' /tmp/ref.sam) && echo "SAM identical (2.4M reads)"
cmp <(grep -v '^@PG' /tmp/ours_p1.sam) <(grep -v '^@PG' /tmp/ours_p16.sam) && echo "-p 1 == -p 16"

#!/bin/bash
# bench.py on a genome large enough that the FM index no longer fits the 256 MB Infinity Cache
MBP=${1:-256}
mkdir -p gpurun_out/big
( time timeout 1500 python bench.py --genome-mbp $MBP --steps 3 --warmup 1 --cpu-repeat 4 ) > gpurun_out/big/bench_${MBP}mbp.json 2> gpurun_out/big/bench_${MBP}mbp.err
tail -5 gpurun_out/big/bench_${MBP}mbp.err
cut -c1-400 gpurun_out/big/bench_${MBP}mbp.json

#!/bin/bash
# End to end through the drop-in binary (process start, index load + transcoding, FASTQ parse, device, SAM text out) on the headline
# workload: 4 M of the bench's reads, file to file.  `-t` prints the wall seconds of each host stage's thread (stages overlap).
#   gpurun --timeout 900 -- 'bash tools/r03_cli_e2e.sh TAG'
T=${1:-r03s}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(timeout 300 python bench.py --steps 1 --warmup 0 --parity-only 2>$O/bench.err | tail -1) > $O/bench.json
C=/tmp/bt2_amd_bench; B=$C/hg38like_3100mbp_s2_bt2l; FQ=/tmp/e2e_4m.fq
cat $C/sample.fq $C/sample.fq $C/sample.fq $C/sample.fq > $FQ
N=$(( $(wc -l < $FQ) / 4 ))
runs=("-p 16 -S /tmp/e2e.sam" "-p 16 -S /dev/null" "-p 16 --batch 1048576 -S /dev/null" "-p 16 --batch 2000000 -S /dev/null")
for run in "${runs[@]}"; do
  s=$(date +%s.%N)
  timeout 200 bowtie2_amd/bin/bowtie2-align-l --sensitive -t -x $B -U $FQ $run 2> $O/e2e.err
  e=$(date +%s.%N)
  awk -v n=$N -v s=$s -v e=$e -v r="$run" 'BEGIN{printf "%s: %d reads in %.2f s wall -> %.0f reads/s end to end (index load included)\n", r, n, e-s, n/(e-s)}' | tee -a $O/e2e.txt
  grep "bt2g\|Time loading" $O/e2e.err | tee -a $O/e2e.txt
done

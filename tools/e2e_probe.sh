#!/bin/bash
# What bounds the drop-in binary end to end: the bench's e2e input (24 M distinct reads) is kept and aligned again under variations.
#   gpurun --timeout 1500 -- 'bash tools/e2e_probe.sh TAG'
T=${1:-r05e2e}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
(BT2_BENCH_KEEP_E2E=1 timeout 900 python bench.py --steps 4 --warmup 2 --parity-only --e2e-reads 24000000 2>$O/bench.err | tail -1) > $O/bench.json
python3 -c "
import json; d=json.load(open('$O/bench.json')); print('resident', round(d['value']), 'in-bench e2e', d['e2e'].get('reads_per_s_after_load'), d['e2e'].get('stages'))" | tee $O/probe.txt
C=/tmp/bt2_amd_bench; B=$C/hg38like_3100mbp_s2_bt2l
run() { name=$1; shift; echo "== $name" | tee -a $O/probe.txt; ( "$@" ) 2>&1 | grep "bt2g" | tee -a $O/probe.txt; }
free -g | head -2 | tee -a $O/probe.txt; df -h /tmp | tail -1 | tee -a $O/probe.txt; nproc | tee -a $O/probe.txt
run "file out" bowtie2_amd/bin/bowtie2-align-l --sensitive -t -p 16 -x $B -U $C/e2e.fq -S $C/e2e.sam
run "file out again" bowtie2_amd/bin/bowtie2-align-l --sensitive -t -p 16 -x $B -U $C/e2e.fq -S $C/e2e.sam
rm -f $C/e2e.sam; sync
run "devnull" bowtie2_amd/bin/bowtie2-align-l --sensitive -t -p 16 -x $B -U $C/e2e.fq -S /dev/null
run "devnull again" bowtie2_amd/bin/bowtie2-align-l --sensitive -t -p 16 -x $B -U $C/e2e.fq -S /dev/null
run "devnull sdma off" env HSA_ENABLE_SDMA=0 bowtie2_amd/bin/bowtie2-align-l --sensitive -t -p 16 -x $B -U $C/e2e.fq -S /dev/null
run "devnull batch 524288" bowtie2_amd/bin/bowtie2-align-l --sensitive -t -p 16 --batch 524288 -x $B -U $C/e2e.fq -S /dev/null
run "devnull batch 1048576" bowtie2_amd/bin/bowtie2-align-l --sensitive -t -p 16 --batch 1048576 -x $B -U $C/e2e.fq -S /dev/null
run "devnull batch 131072" bowtie2_amd/bin/bowtie2-align-l --sensitive -t -p 16 --batch 131072 -x $B -U $C/e2e.fq -S /dev/null
run "devnull p 8" bowtie2_amd/bin/bowtie2-align-l --sensitive -t -p 8 -x $B -U $C/e2e.fq -S /dev/null
run "devnull p 4" bowtie2_amd/bin/bowtie2-align-l --sensitive -t -p 4 -x $B -U $C/e2e.fq -S /dev/null
rm -f $C/e2e.fq $C/e2e.sam

#!/bin/bash
# After a bench run that found SAM differences (gpurun_out/parity_diff): align just those reads again -- product with and
# without the batch pre-computation kernels, and the reference -- so that the culprit stage can be told apart.
R=$GRAFT_REPO_ROOT; D=$R/gpurun_out/parity_diff; mkdir -p $D; [ -s $D/reads.fq ] || cp $R/tools/probe/reads.fq $D/reads.fq
IDX=${1:-/tmp/bt2_amd_bench/hg38like_1024mbp_s2_bt2l}
[ -s $D/reads.fq ] || { echo "no differing reads"; exit 0; }
cd $R
bowtie2_amd/bin/bowtie2-align-l --sensitive --met -x $IDX -U $D/reads.fq > $D/gpu_default.sam 2> $D/gpu_default.err
BT2G_NO_PRECOMP=1 bowtie2_amd/bin/bowtie2-align-l --sensitive --met -x $IDX -U $D/reads.fq > $D/gpu_noprecomp.sam 2> $D/gpu_noprecomp.err
oracle/_ref/bowtie2-align-l-v256 --sensitive -x $IDX -U $D/reads.fq > $D/ref.sam 2> $D/ref.err
oracle/_ref/bowtie2-align-l --sensitive -x $IDX -U $D/reads.fq > $D/ref_sse.sam 2> /dev/null
for f in gpu_default gpu_noprecomp ref ref_sse; do grep -v '^@' $D/$f.sam | cut -f1-9,12- > $D/$f.short; done
echo "default vs ref:"; diff $D/gpu_default.short $D/ref.short | head -20
echo "noprecomp vs ref:"; diff $D/gpu_noprecomp.short $D/ref.short | head -20
echo "ref avx2 vs sse2:"; diff $D/ref.short $D/ref_sse.short | head
grep MET $D/gpu_default.err | head -5; grep MET $D/gpu_noprecomp.err | head -5

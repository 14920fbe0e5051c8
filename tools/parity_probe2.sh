#!/bin/bash
# second-level probe: the CPU twin of the worker (tests/hostsim, test-only) on the differing reads against the same index
R=$GRAFT_REPO_ROOT; D=$R/gpurun_out/parity_diff; mkdir -p $D; [ -s $D/reads.fq ] || cp $R/tools/probe/reads.fq $D/reads.fq
IDX=${1:-/tmp/bt2_amd_bench/hg38like_1024mbp_s2_bt2l}
cd $R/tests/hostsim && g++ -O2 -std=c++17 -w -I../../include -o hostsim hostsim.cpp ../../bowtie2_amd/csrc/bt2g_index.cpp -lz -lpthread || exit 1
cd $R
tests/hostsim/hostsim --sensitive --met -x $IDX -U $D/reads.fq > $D/host.sam 2> $D/host.err
grep -v '^@' $D/host.sam | cut -f1-9,12- > $D/host.short
echo "hostsim vs ref:"; diff $D/host.short $D/ref.short | head; grep MET $D/host.err | head -3

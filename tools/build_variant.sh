#!/bin/bash
# build/variants/libbt2g_<tag>.so = the current tree with bt2g_align_kernel.hip compiled under other flags (the other objects are the
# tree's own, including the 96-register class: run the variant with BT2G_NO_W5=1 to see its own kernel on unpaired end-to-end batches):
#   tools/build_variant.sh TAG WPE [extra flags...]   -- for tools/bench_variants.sh (BT2G_LIB)
set -e
TAG=$1; WPE=$2; shift 2
cd "$(dirname "$0")/../bowtie2_amd/csrc"
make -s
O=../../build/variants; mkdir -p $O
/opt/rocm/bin/hipcc -DBT2G_WAVES_PER_EU=$WPE "$@" -w -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -c bt2g_align_kernel.hip -o $O/align_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libbt2g_$TAG.so bt2g_capi.o bt2g_kernels.o bt2g_rankidx.o $O/align_$TAG.o bt2g_align_kernel_w5.o bt2g_index.o bt2g_build.o bt2g_search.o -lz -lpthread
rm -f $O/align_$TAG.o
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $O/libbt2g_$TAG.so 2>/dev/null | grep -A12 "k_align_readsIm" | grep "private_segment_fixed_size\|vgpr_count\|vgpr_spill\|sgpr_spill\|agpr_count" | tr -s ' ' | tr '\n' ' '; echo

#!/bin/bash
# How fast is the drop-in binary's HOST side (FASTQ split / parse / pack, SAM format / write) on its own?  Runs the product's driver on the CPU
# (tests/hostsim/hostsim_driver_twin, built by `pytest tests/test_driver_twin.py`) with BT2G_TWIN_NULL_ALIGNER=1: no alignment at all, every read
# "aligns" without edits at a made-up position, so the stage times of `-t` are those of the host pipeline alone.  No GPU needed.
#   usage: tools/host_pipeline_rate.sh <reads.fq> [threads ...]        e.g. tools/host_pipeline_rate.sh /tmp/reads.fq 1 4 16
FQ=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for p in "${@:-1 8}"; do
  s=$(date +%s%N)
  BT2G_TWIN_NULL_ALIGNER=1 $ROOT/tests/hostsim/hostsim_driver_twin -x $ROOT/tests/golden/tiny_s -U $FQ -p $p -t -S /dev/null 2>&1 | grep "host stages"
  echo "-p $p: $(( ($(date +%s%N) - s) / 1000000 )) ms wall"
done

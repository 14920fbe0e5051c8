#!/bin/bash
# usage: det.sh <libdir> <runs>
export LD_LIBRARY_PATH=$1:$LD_LIBRARY_PATH
C=/tmp/bt2_amd_test_cache
for w in s l; do
  d=$(ls -d $C/rep_$w 2>/dev/null)
  rm -f /tmp/det_$w.txt
  for i in $(seq 1 $2); do
    bowtie2_amd/bin/bowtie2-align-$w --met -x $d/rep -U $d/rep.fq -S /dev/null 2>&1 | grep ^MET | md5sum >> /tmp/det_$w.txt
  done
  echo "lib=$1 width=$w distinct=$(sort -u /tmp/det_$w.txt | wc -l) of $2"
done

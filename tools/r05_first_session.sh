#!/bin/bash
# First GPU session of the next round: the occupancy points VERDICT r3 asked for and round 4 did not measure, in one call (~4 GPU-minutes).
#   on the build host first:   tools/r05_first_session.sh prepare        (builds build/variants/libbt2g_{w3,w4}.so: they travel with the tree)
#   then:                      gpurun --timeout 900 -- 'bash tools/r05_first_session.sh run r05a'
# w3 = the worker with 168 registers, 3 waves per SIMD (12 waves per CU, the point of "zero spills"); w4 = the 128-register class alone;
# cur = the shipped library (96 registers / 5 waves per SIMD for unpaired end-to-end batches).  Same box, same reads, SAM compared each time.
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "prepare" ]; then
  tools/build_variant.sh w3 3 -DBT2G_NUM_VGPR=168
  tools/build_variant.sh w4 4 -DBT2G_NUM_VGPR=128
  exit 0
fi
T=${2:-r05a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp
for v in cur w4 w3; do
  unset BT2G_LIB BT2G_NO_W5
  if [ $v != cur ]; then export BT2G_LIB=$R/build/variants/libbt2g_$v.so BT2G_NO_W5=1; fi
  (timeout 300 python bench.py --steps 6 --warmup 2 --parity-only 2>$O/bench_$v.err | tail -1) > $O/bench_$v.json || true
  python3 -c "
import json
d=json.load(open('$O/bench_$v.json')); c=d['config']; print('$v', round(d['value']), 'reads/s', c['kernel_ms_per_step'], 'parity', c.get('parity_identical'), 'flagged', c.get('reads_overflowed'))" | tee -a $O/summary.txt || true
done

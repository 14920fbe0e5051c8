#!/bin/bash
# bench.py against several builds of libbt2g.so (build/variants/libbt2g_<tag>.so) -- used to bisect kernel-time changes
mkdir -p gpurun_out/variants
for v in "$@"; do
  if [ "$v" = "cur" ]; then unset BT2G_LIB; else export BT2G_LIB=$PWD/build/variants/libbt2g_$v.so; fi
  timeout 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/variants/$v.err | tail -1 > gpurun_out/variants/$v.json
  python3 -c "
import json,sys
j=json.load(open('gpurun_out/variants/$v.json'))
print('$v', 'reads/s %.0f' % j['value'], 'ms/step %.2f' % j['ms_per_step'], 'k_align_reads ms', j.get('roofline',{}).get('kernel_ms_per_step'))
"
done

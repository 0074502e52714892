#!/bin/bash
# bench lines for the base library and several variants.  usage: tools/ab_many.sh <out tag> "<variants...>" workloads...
TAG=$1; VARS=$2; shift 2
for v in "" $VARS; do for wl in "$@"; do
  lib=""; [ -n "$v" ] && lib=$PWD/rustraytracer_amd/csrc/build/variants/$v.so
  RT_AMD_LIB=$lib timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 3 --warmup 1 > gpurun_out/tmp.json 2>gpurun_out/tmp.err
  python - $wl "${v:-base}" <<'PY'
import json,sys
d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; k=r['kernels']['k_shade']
print('%s %-12s Mrays/s %.0f ms %.2f trace %.2f shade %.2f | nodes/ray %.3f prims/ray %.3f'%(sys.argv[1],sys.argv[2],d['value'],d['ms_per_step'],r['avg_launch_ms']*r['launches_per_step'],k['avg_launch_ms']*k['launches_per_step'],r['nodes_per_ray'],r['tris_per_ray']))
PY
done; done 2>&1 | tee gpurun_out/$TAG.txt

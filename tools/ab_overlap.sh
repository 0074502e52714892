#!/bin/bash
# On the GPU box: the light kernel on the side stream (RT_LIGHT_OVERLAP) against the serial schedule.  usage: tools/ab_overlap.sh [workloads...]
run() {
  for wl in "$@"; do
    timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 3 --warmup 1 > gpurun_out/tmp.json 2>gpurun_out/tmp.err || { echo "FAILED"; tail -3 gpurun_out/tmp.err; continue; }
    python - $wl <<'PY'
import json,sys
d=json.load(open('gpurun_out/tmp.json'))
r=d['roofline']; K=r['kernels']
print('   %s Mrays/s %.0f ms %.2f trace %.2f classify %.2f shade %.2f'%(sys.argv[1],d['value'],d['ms_per_step'],K['k_trace']['ms_per_step'],K['k_classify']['ms_per_step'],K['k_shade']['ms_per_step']))
PY
  done
}
WLS="${@:-c4 c3 c2}"
echo "serial";                                   run $WLS
echo "serial, k_trace 4 blocks/CU";              RT_TRACE_BLOCKS_PER_CU=4 run $WLS
echo "overlap, k_trace 5 blocks/CU";             RT_LIGHT_OVERLAP=1 run $WLS
echo "overlap, k_trace 4 blocks/CU";             RT_LIGHT_OVERLAP=1 RT_TRACE_BLOCKS_PER_CU=4 run $WLS
echo "overlap, k_trace 4, light 1 block/CU";     RT_LIGHT_OVERLAP=1 RT_TRACE_BLOCKS_PER_CU=4 RT_LIGHT_BLOCKS_PER_CU=1 run $WLS
echo "overlap, k_trace 4, light 2 blocks/CU";    RT_LIGHT_OVERLAP=1 RT_TRACE_BLOCKS_PER_CU=4 RT_LIGHT_BLOCKS_PER_CU=2 run $WLS
echo "overlap, k_trace 5, light 1 block/CU";     RT_LIGHT_OVERLAP=1 RT_LIGHT_BLOCKS_PER_CU=1 run $WLS

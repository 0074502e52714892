#!/bin/bash
source_run() { :; }
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
export RT_LIGHT_OVERLAP=1 RT_LIGHT_BLOCKS_PER_CU=1
echo "overlap, light 1 block/CU (first)";            run $WLS
echo "  + joined before k_trace";                    RT_LIGHT_JOIN_TRACE=1 run $WLS
echo "  + launched last";                            RT_LIGHT_LAST=1 run $WLS
echo "  + launched last, joined before k_trace";     RT_LIGHT_LAST=1 RT_LIGHT_JOIN_TRACE=1 run $WLS
echo "  + RT_LANES=2";                               RT_LANES=2 run $WLS
unset RT_LIGHT_OVERLAP RT_LIGHT_BLOCKS_PER_CU
echo "serial RT_LANES=2";                            RT_LANES=2 run $WLS

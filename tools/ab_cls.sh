#!/bin/bash
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
echo "default";              run $WLS
echo "RT_CLS_STREAMS=2";     RT_CLS_STREAMS=2 run $WLS
echo "RT_CLS_STREAMS=3";     RT_CLS_STREAMS=3 run $WLS
echo "default again";        run $WLS

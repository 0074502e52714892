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
export RT_CLS_STREAMS=2
for m in 4 8 12 2 6 10; do echo "side mask $m"; RT_CLS_SIDE_MASK=$m run c4; done
for m in 4 2; do echo "side mask $m"; RT_CLS_SIDE_MASK=$m run c3 c2 hdr teapot; done

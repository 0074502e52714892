#!/bin/bash
# usage: tools/gpu_quick.sh [tests] [workloads...]   -- parity tests (optional) then one bench line per workload
mkdir -p gpurun_out
if [ "$1" = "tests" ]; then shift; timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -4; fi
for wl in "$@"; do
  timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 3 --warmup 1 > gpurun_out/tmp.json 2>gpurun_out/tmp.err
  python - $wl <<'PY'
import json,sys
d=json.load(open('gpurun_out/tmp.json'))
r=d['roofline']; K=r['kernels']
print('%s Mrays/s %.0f ms %.2f trace %.2f classify %.2f shade %.2f'%(sys.argv[1],d['value'],d['ms_per_step'],K['k_trace']['ms_per_step'],K['k_classify']['ms_per_step'],K['k_shade']['ms_per_step']))
PY
done

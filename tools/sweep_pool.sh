#!/bin/bash
# pool size (paths in flight) against the 256-MiB Infinity Cache -- 2 x 248 B of ping-pong state per
# path: 2^19 paths = 260 MB.  One line per workload and pool size.  usage: tools/sweep_pool.sh [out tag]
TAG=${1:-r03}
for wl in c2 c3 c4; do for lg in 19 20 22 24 26; do
  timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 2 --warmup 1 --paths-in-flight $((1 << lg)) > gpurun_out/tmp.json 2>gpurun_out/tmp.err
  python - $wl $lg <<'PY'
import json,sys
d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; k=r['kernels']['k_shade']
print('%s pool 2^%s Mrays/s %.0f ms %.2f trace %.2f shade %.2f launches %d'%(sys.argv[1],sys.argv[2],d['value'],d['ms_per_step'],r['avg_launch_ms']*r['launches_per_step'],k['avg_launch_ms']*k['launches_per_step'],r['launches_per_step']))
PY
done; done 2>&1 | tee gpurun_out/${TAG}_sweep_pool.txt

#!/bin/bash
# pools beyond the 64 Mi default (needs the cap in abi.hip at 2^28): 2^27 = 70 GB, 2^28 = 140 GB of path state.  usage: tools/sweep_pool_big.sh [tag]
TAG=${1:-r03}
for wl in c4 c3; do for lg in 26 27 28; do
  timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 2 --warmup 1 --paths-in-flight $((1 << lg)) > gpurun_out/tmp.json 2>gpurun_out/tmp.err
  python - $wl $lg <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; k=r['kernels']['k_shade']
    print('%s pool 2^%s Mrays/s %.0f ms %.2f trace %.2f shade %.2f launches %d'%(sys.argv[1],sys.argv[2],d['value'],d['ms_per_step'],r['avg_launch_ms']*r['launches_per_step'],k['avg_launch_ms']*k['launches_per_step'],r['launches_per_step']))
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'failed', e, open('gpurun_out/tmp.err').read()[-300:])
PY
done; done 2>&1 | tee gpurun_out/${TAG}_sweep_pool_big.txt

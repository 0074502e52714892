#!/bin/bash
mkdir -p gpurun_out
for ns in 1 4 16 64; do for blk in 256 64; do
  for wl in c2 c3; do
    RT_SHARDS=$ns RT_SHADE_BLOCK=$blk timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 4 --warmup 1 > gpurun_out/tmp.json 2>gpurun_out/tmp.err
    python - "$ns" "$blk" "$wl" <<'PY'
import json,sys
d=json.load(open('gpurun_out/tmp.json'))
r=d['roofline']; k=r['kernels']['k_shade']
print('ns',sys.argv[1],'blk',sys.argv[2],sys.argv[3],'Mrays/s %.0f ms %.2f trace %.2f shade %.2f'%(d['value'],d['ms_per_step'],r['avg_launch_ms']*r['launches_per_step'],k['avg_launch_ms']*k['launches_per_step']))
PY
  done
done; done 2>&1 | tee gpurun_out/r02c_shard_sweep.txt

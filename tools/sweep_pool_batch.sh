#!/bin/bash
# pool size x batch size on one workload (both log2).  usage: tools/sweep_pool_batch.sh <workload> "<pool:batch> ..." [bench args]
WL=$1; CFGS=$2; shift 2
for pb in $CFGS; do
  p=${pb%%:*}; b=${pb##*:}
  RT_BATCH_LOG2=$b timeout 900 python bench.py --workload $WL --no-cpu-baseline --no-extra --paths-in-flight $((1 << p)) "$@" > gpurun_out/tmp.json 2>gpurun_out/tmp.err
  python - $WL $p $b <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; k=r['kernels']['k_shade']
    print('%s pool 2^%s batch 2^%s Mrays/s %.0f ms %.2f trace %.2f shade %.2f launches %d'%(sys.argv[1],sys.argv[2],sys.argv[3],d['value'],d['ms_per_step'],r['avg_launch_ms']*r['launches_per_step'],k['avg_launch_ms']*k['launches_per_step'],r['launches_per_step']))
except Exception as e:
    print(sys.argv[1:], 'failed', e, open('gpurun_out/tmp.err').read()[-300:])
PY
done

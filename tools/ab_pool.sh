#!/bin/bash
run() { # batchlog2 pool wl
  RT_BATCH_LOG2=$1 timeout 900 python bench.py --workload $3 --paths-in-flight $2 --no-cpu-baseline --no-extra --steps 3 --warmup 1 > gpurun_out/tmp.json 2>gpurun_out/tmp.err || { echo FAILED; tail -2 gpurun_out/tmp.err; return; }
  python - "$@" <<'PY'
import json,sys
d=json.load(open('gpurun_out/tmp.json'))
K=d['roofline']['kernels']
print('batch 2^%s pool %s %s: Mrays/s %.0f ms %.2f trace %.2f classify %.2f shade %.2f launches %.0f'%(sys.argv[1],sys.argv[2],sys.argv[3],d['value'],d['ms_per_step'],K['k_trace']['ms_per_step'],K['k_classify']['ms_per_step'],K['k_shade']['ms_per_step'],d['roofline']['k_trace_detail']['launches_per_step']))
PY
}
for wl in c4 c5; do
for p in 67108864 134217728 268435456; do run 31 $p $wl; done
run 30 134217728 $wl
done

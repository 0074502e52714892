#!/bin/bash
run() { label=$1; wl=$2; shift 2
  env "$@" timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 3 --warmup 1 > gpurun_out/tmp.json 2>gpurun_out/tmp.err
  python - "$label $wl" <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; k=r['kernels']['k_shade']
    print('%-22s Mrays/s %.0f ms %.2f trace %.2f shade %.2f other %.2f'%(sys.argv[1],d['value'],d['ms_per_step'],r['avg_launch_ms']*r['launches_per_step'],k['avg_launch_ms']*k['launches_per_step'], d['device_ms_per_step']-r['avg_launch_ms']*r['launches_per_step']-k['avg_launch_ms']*k['launches_per_step']))
except Exception as e: print(sys.argv[1],'FAILED',e, open('gpurun_out/tmp.err').read()[-300:])
PY
}
for wl in c4 c3; do
run "pad 0" $wl RT_STATE_PAD=0
run "pad 256" $wl RT_STATE_PAD=256
run "pad 4352" $wl RT_STATE_PAD=4352
run "pad 69888" $wl RT_STATE_PAD=69888
run "pad 1052928" $wl RT_STATE_PAD=1052928
done

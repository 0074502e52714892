#!/bin/bash
run() { label=$1; wl=$2; shift 2
  env "$@" timeout 900 python bench.py --workload $wl --precision f32 --no-cpu-baseline --no-extra --steps 2 --warmup 1 > gpurun_out/tmp.json 2>gpurun_out/tmp.err
  python - "$label $wl" <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; k=r['kernels']['k_shade']
    print('%-22s Mrays/s %.0f ms %.1f trace %.1f shade %.1f launches %d'%(sys.argv[1],d['value'],d['ms_per_step'],r['avg_launch_ms']*r['launches_per_step'],k['avg_launch_ms']*k['launches_per_step'],r['launches_per_step']))
except Exception as e: print(sys.argv[1],'FAILED',e, open('gpurun_out/tmp.err').read()[-300:])
PY
}
V=$PWD/rustraytracer_amd/csrc/build/variants
for wl in c4 c3 c2; do
run "f32 base" $wl X=1
run "f32 shade 4 waves" $wl RT_AMD_LIB=$V/f32w4.so
run "f32 shade 5 waves" $wl RT_AMD_LIB=$V/f32w5.so
done

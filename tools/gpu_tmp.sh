#!/bin/bash
V=$PWD/rustraytracer_amd/csrc/build/variants
RT_AMD_LIB=$V/ol.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_arms.py -x -q -m gpu 2>&1 | tail -1
run() { label=$1; wl=$2; shift 2
  env "$@" timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 3 --warmup 1 > gpurun_out/tmp.json 2>gpurun_out/tmp.err
  python - "$label $wl" <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; k=r['kernels']['k_shade']
    print('%-22s Mrays/s %.0f ms %.2f trace %.2f shade %.2f'%(sys.argv[1],d['value'],d['ms_per_step'],r['avg_launch_ms']*r['launches_per_step'],k['avg_launch_ms']*k['launches_per_step']))
except Exception as e: print(sys.argv[1],'FAILED',e, open('gpurun_out/tmp.err').read()[-300:])
PY
}
for wl in c4 c3; do
run "base" $wl X=1
run "other last" $wl RT_AMD_LIB=$V/ol.so
run "other last, 6 cls" $wl RT_AMD_LIB=$V/ol6.so
run "6 classes" $wl RT_AMD_LIB=$V/c6.so
done

#!/bin/bash
# bench lines of the default library and of a variant library, alternating.  usage: tools/ab_lib.sh <variant name> workloads...
V=$PWD/rustraytracer_amd/csrc/build/variants/$1.so; shift
for wl in "$@"; do for lib in "" $V "" $V; do
  RT_AMD_LIB=$lib timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 3 --warmup 1 > gpurun_out/tmp.json 2>gpurun_out/tmp.err
  python - $wl "${lib:-default}" <<'PY'
import json,sys,os
d=json.load(open('gpurun_out/tmp.json'))
r=d['roofline']; K=r['kernels']
print('%s %-10s Mrays/s %.0f ms %.2f trace %.2f classify %.2f shade %.2f'%(sys.argv[1],os.path.basename(sys.argv[2])[:10],d['value'],d['ms_per_step'],K['k_trace']['ms_per_step'],K['k_classify']['ms_per_step'],K['k_shade']['ms_per_step']))
PY
done; done

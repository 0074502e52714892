#!/bin/bash
# A/B a variant library: parity subset, then bench lines for both.  usage: tools/gpu_ab.sh <variant name> workloads...
V=$PWD/rustraytracer_amd/csrc/build/variants/$1.so; shift
RT_AMD_LIB=$V timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_arms.py -x -q -m gpu $PYTEST_EXTRA 2>&1 | tail -${TAIL:-2}
for wl in "$@"; do for lib in "" $V; do
  RT_AMD_LIB=$lib timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 3 --warmup 1 > gpurun_out/tmp.json 2>gpurun_out/tmp.err
  python - $wl "${lib:-base}" <<'PY'
import json,sys,os
d=json.load(open('gpurun_out/tmp.json'))
r=d['roofline']; K=r['kernels']
print('%s %-8s Mrays/s %.0f ms %.2f trace %.2f classify %.2f shade %.2f | nodes/ray %.3f prims/ray %.3f'%(sys.argv[1],os.path.basename(sys.argv[2])[:8],d['value'],d['ms_per_step'],K['k_trace']['ms_per_step'],K['k_classify']['ms_per_step'],K['k_shade']['ms_per_step'],r['k_trace_detail']['nodes_per_ray'],r['k_trace_detail']['tris_per_ray']))
PY
done; done

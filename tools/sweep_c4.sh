#!/bin/bash
# parameter sweeps on one workload (env knobs of abi.hip); one line per setting.  usage: tools/sweep_c4.sh [workload] [out tag]
WL=${1:-c4}; TAG=${2:-r04}
mkdir -p gpurun_out
run() { # label, env...
  label=$1; shift
  env "$@" timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-extra --steps 2 --warmup 1 > gpurun_out/tmp.json 2>gpurun_out/tmp.err
  python - "$label" <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/tmp.json'))
    r=d['roofline']; K=r['kernels']
    print('%-34s Mrays/s %.0f ms %.1f trace %.1f classify %.1f shade %.1f launches %d'%(sys.argv[1],d['value'],d['ms_per_step'],K['k_trace']['ms_per_step'],K['k_classify']['ms_per_step'],K['k_shade']['ms_per_step'],r['k_trace_detail']['launches_per_step']))
except Exception as e:
    print(sys.argv[1],'FAILED',e)
PY
}
{
run base X=1
for rf in 16 32 40; do run "refill $rf" RT_TRACE_REFILL=$rf; done
for nb in 3 5 6; do run "node_bias $nb" RT_TRACE_NODE_BIAS=$nb; done
for rs in 64 256 512; do run "reserve $rs" RT_TRACE_RESERVE=$rs; done
for tp in 262144 2097152; do run "tail paths $tp" RT_TAIL_PATHS=$tp; done
for lg in 1 3; do run "mirror lag $lg" RT_MIRROR_LAG=$lg; done
run base X=1
} 2>&1 | tee gpurun_out/${TAG}_sweep_$WL.txt

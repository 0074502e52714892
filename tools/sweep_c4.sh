#!/bin/bash
# parameter sweeps on one workload (env knobs of abi.hip); one line per setting.  usage: tools/sweep_c4.sh [workload] [out tag]
WL=${1:-c4}; TAG=${2:-r03}
mkdir -p gpurun_out
run() { # label, env...
  label=$1; shift
  env "$@" timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-extra --steps 2 --warmup 1 > gpurun_out/tmp.json 2>gpurun_out/tmp.err
  python - "$label" <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/tmp.json'))
    r=d['roofline']; k=r['kernels']['k_shade']
    print('%-34s Mrays/s %.0f ms %.1f trace %.1f shade %.1f launches %d'%(sys.argv[1],d['value'],d['ms_per_step'],r['avg_launch_ms']*r['launches_per_step'],k['avg_launch_ms']*k['launches_per_step'],r['launches_per_step']))
except Exception as e:
    print(sys.argv[1],'FAILED',e)
PY
}
{
run base X=1
for rf in 8 16 32 40 48; do run "refill $rf" RT_TRACE_REFILL=$rf; done
for nb in 2 3 5 6 8; do run "node_bias $nb" RT_TRACE_NODE_BIAS=$nb; done
for rf in 16 32; do for nb in 3 6; do run "refill $rf node_bias $nb" RT_TRACE_REFILL=$rf RT_TRACE_NODE_BIAS=$nb; done; done
for b in 2 3; do run "trace blocks/CU $b" RT_TRACE_BLOCKS_PER_CU=$b; done
run "lanes 2" RT_LANES=2
run "pool 2^24" RT_POOL=1
} 2>&1 | tee gpurun_out/${TAG}_sweep_$WL.txt

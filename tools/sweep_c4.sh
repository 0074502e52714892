#!/bin/bash
# parameter sweeps on the headline workload (env knobs of abi.hip); one line per setting
mkdir -p gpurun_out
run() { # label, env...
  label=$1; shift
  env "$@" timeout 600 python bench.py --workload c4 --no-cpu-baseline --no-extra --steps 2 --warmup 1 > gpurun_out/tmp.json 2>gpurun_out/tmp.err
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
for rf in 16 32 40; do run "refill $rf" RT_TRACE_REFILL=$rf; done
for nb in 3 5 6; do run "node_bias $nb" RT_TRACE_NODE_BIAS=$nb; done
run "batch 2^29" RT_BATCH_LOG2=29
run "batch 2^27" RT_BATCH_LOG2=27
run "tail 262144" RT_TAIL_PATHS=262144
run "tail 2097152" RT_TAIL_PATHS=2097152
run "lanes 2" RT_LANES=2
} 2>&1 | tee gpurun_out/r02_sweep_c4.txt

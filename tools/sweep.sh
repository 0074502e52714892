#!/bin/bash
# usage: tools/sweep.sh "VAR=a,b VAR2=c" ... each arg is an env assignment set; runs bench quick for each
for cfgs in "$@"; do
  out=$(env $cfgs python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print(f\"{d['value']:.1f} Mrays/s dev_ms {d['device_ms_per_step']:.1f} trace_avg {r['avg_launch_ms']:.3f}\")")
  echo "$cfgs -> $out"
done

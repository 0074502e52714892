#!/bin/bash
# usage: tools/sweep3.sh "<env assignments> -- <bench args>" ...   (like sweep2.sh, plus the traversal counters)
for cfgs in "$@"; do
  envs="${cfgs%%--*}"; args="${cfgs#*--}"
  [ "$envs" == "$cfgs" ] && args=""
  out=$(env $envs python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print(f\"{d['value']:.1f} Mrays/s ms/step {d['ms_per_step']:.1f} trace_avg {r['avg_launch_ms']:.3f} launches {r['launches_per_step']:.0f} nodes/ray {r['nodes_per_ray']:.3f} tris/ray {r['tris_per_ray']:.3f} B/ray {r['bytes_per_ray']:.0f} frac {r['frac']:.3f} trace_share {r['trace_share_of_device_time']:.3f}\")")
  echo "$cfgs -> $out"
done

#!/bin/bash
# On the GPU box: parity tests, then the C2 bench line (no CPU baseline) -> gpurun_out/
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/gputests.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/bench_quick.json'))
    r=d['roofline']
    print(f"Mrays/s {d['value']:.1f}  ms/step {d['ms_per_step']:.1f} dev_ms {d['device_ms_per_step']:.1f} trace_avg_ms {r['avg_launch_ms']:.3f} launches {r['launches_per_step']} frac {r['frac']:.4f} GB/s {r['achieved']:.0f} trace_share {r['trace_share_of_device_time']:.3f} nodes/ray {r['nodes_per_ray']:.2f} tris/ray {r['tris_per_ray']:.2f}")
except Exception as e:
    print("bench parse failed", e); print(open('gpurun_out/bench_quick.err').read()[-2000:])
PY

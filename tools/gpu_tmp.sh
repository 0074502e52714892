#!/bin/bash
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
python - <<'PY'
import time, torch, bench, rustraytracer_amd as rr
b = bench.Bench("c4", 0, 1, 0, "cuda")
b.step(); torch.cuda.synchronize()
t0=time.time(); st=b.step(); torch.cuda.synchronize(); t1=time.time()
stc=b.counted(); t2=time.time()
print("plain %.2f s, counted %.2f s; nodes/ray %.3f tris/ray %.3f" % (t1-t0, t2-t1, stc.nodes_fetched/stc.rays, stc.tris_tested/stc.rays))
PY
RT_DIAG=1 python bench.py --workload c2 --no-cpu-baseline --no-extra --steps 2 2>&1 | grep "rt diag" | tail -2

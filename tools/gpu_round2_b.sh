#!/bin/bash
# sharded lists: parity first, then the bench
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02b_gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02b_gputests.log )
tail -5 gpurun_out/r02b_gputests.log
( timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r02b_bench_c4.json 2> gpurun_out/r02b_bench_c4.err; echo "bench rc=$?" >> gpurun_out/r02b_bench_c4.err )
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02b_bench_c4.json'))
print('C4', d['value'], d['ms_per_step'], 'trace', d['roofline']['avg_launch_ms']*d['roofline']['launches_per_step'], 'shade', d['roofline']['kernels']['k_shade']['avg_launch_ms']*d['roofline']['kernels']['k_shade']['launches_per_step'])
for e in d.get('extra',[]): print(e['workload'][:3], e['value'], e['ms_per_step'], 'trace', e['k_trace_ms_per_step'], 'shade', e['k_shade_ms_per_step'])
PY
tail -3 gpurun_out/r02b_bench_c4.err

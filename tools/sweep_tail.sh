for t in 262144 786432 2097152 6291456; do
  RT_TAIL_PATHS=$t timeout 900 python bench.py --workload c4 --no-cpu-baseline --no-extra --steps 2 --warmup 1 > gpurun_out/tmp.json 2>gpurun_out/tmp.err
  python - $t <<'PY'
import json,sys
d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; k=r['kernels']['k_shade']
print('c4 tail_paths %s Mrays/s %.0f ms %.2f trace %.2f shade %.2f launches %d'%(sys.argv[1],d['value'],d['ms_per_step'],r['avg_launch_ms']*r['launches_per_step'],k['avg_launch_ms']*k['launches_per_step'],r['launches_per_step']))
PY
done

for l in 2 1 2 1; do
  RT_MIRROR_LAG=$l timeout 900 python bench.py --workload ${WL:-c4} --no-cpu-baseline --no-extra --steps 3 --warmup 1 > gpurun_out/tmp.json 2>gpurun_out/tmp.err
  python - $l ${WL:-c4} <<'PY'
import json,sys
d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; k=r['kernels']['k_shade']
print('%s lag %s Mrays/s %.0f ms %.2f trace %.2f shade %.2f launches %d'%(sys.argv[2],sys.argv[1],d['value'],d['ms_per_step'],r['avg_launch_ms']*r['launches_per_step'],k['avg_launch_ms']*k['launches_per_step'],r['launches_per_step']))
PY
done

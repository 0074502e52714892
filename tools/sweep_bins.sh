for b in 8 32 64; do
  RT_BVH_BINS=$b timeout 900 python bench.py --workload c4 --no-cpu-baseline --no-extra --steps 2 --warmup 1 > gpurun_out/tmp.json 2>gpurun_out/tmp.err
  python - $b <<'PY'
import json,sys
d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; k=r['kernels']['k_shade']
print('bins %s Mrays/s %.0f ms %.2f trace %.2f shade %.2f | nodes/ray %.3f prims/ray %.3f commit %.0f ms'%(sys.argv[1],d['value'],d['ms_per_step'],r['avg_launch_ms']*r['launches_per_step'],k['avg_launch_ms']*k['launches_per_step'],r['nodes_per_ray'],r['tris_per_ray'],d['config']['scene_commit_ms']))
PY
done

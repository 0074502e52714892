#!/bin/bash
mkdir -p gpurun_out
V=rustraytracer_amd/csrc/build/variants
run() { # name lib
  for wl in c2 c3; do
    RT_AMD_LIB=$2 timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 4 --warmup 1 > gpurun_out/tmp.json 2>gpurun_out/tmp.err
    python - "$1" "$wl" <<'PY'
import json,sys
d=json.load(open('gpurun_out/tmp.json'))
r=d['roofline']; k=r['kernels']['k_shade']
print('%-10s %s Mrays/s %.0f ms %.2f trace %.2f shade %.2f'%(sys.argv[1],sys.argv[2],d['value'],d['ms_per_step'],r['avg_launch_ms']*r['launches_per_step'],k['avg_launch_ms']*k['launches_per_step']))
PY
  done
}
{
run base ""
for v in condnee sort0 both lds32 lds64 lds128s12; do run $v $PWD/$V/$v.so; done
for v in both lds64; do
  RT_AMD_LIB=$PWD/$V/$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "render_matches_oracle or committed_oracle_films or intersect_batch_bit_exact or full_size" 2>&1 | tail -2
done
} 2>&1 | tee gpurun_out/r02d_variants.txt

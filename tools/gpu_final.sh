#!/bin/bash
# end-of-round evidence: refreshed profiles for c4 / c3 / c2, the default bench line, the 2-rank gloo dry run
mkdir -p gpurun_out
for tag in c4 c3 c2; do
  timeout 1500 bash tools/refresh_profiles.sh $tag > gpurun_out/refresh_$tag.log 2>&1
  tail -c 200 gpurun_out/refresh_$tag.log; echo
done
( timeout 1200 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; echo "rc=$?" >> gpurun_out/r02_bench_default.err )
( timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --backend gloo --workload c3 --steps 2 --warmup 1 > gpurun_out/r02_bench_c3_2ranks_gloo.json 2> gpurun_out/r02_bench_c3_2ranks_gloo.err; echo "rc=$?" >> gpurun_out/r02_bench_c3_2ranks_gloo.err )
tail -c 300 gpurun_out/r02_bench_default.json; echo; tail -2 gpurun_out/r02_bench_default.err; tail -c 300 gpurun_out/r02_bench_c3_2ranks_gloo.json

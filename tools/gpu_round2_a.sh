#!/bin/bash
# first GPU call of round 2: tests, smoke, C4 bench, 2-rank gloo dry run of bench.py
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_gputests.log )
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r02_smoke.log )
( timeout 900 python bench.py > gpurun_out/r02_bench_c4.json 2> gpurun_out/r02_bench_c4.err; echo "bench rc=$?" >> gpurun_out/r02_bench_c4.err )
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --workload c2 --steps 3 --warmup 1 > gpurun_out/r02_bench_c2_2ranks_gloo.json 2> gpurun_out/r02_bench_c2_2ranks_gloo.err; echo "rc=$?" >> gpurun_out/r02_bench_c2_2ranks_gloo.err )
tail -5 gpurun_out/r02_gputests.log; tail -2 gpurun_out/r02_smoke.log; tail -c 600 gpurun_out/r02_bench_c4.json; tail -3 gpurun_out/r02_bench_c4.err; tail -c 400 gpurun_out/r02_bench_c2_2ranks_gloo.json

#!/bin/bash
# On the GPU box: everything profiles/ is built from at the end of a round.  usage: tools/gpu_final.sh rNN
R=${1:-r04}
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/${R}_gputests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/${R}_gputests.log
for wl in c4 c3 c2; do bash tools/refresh_profiles.sh $wl > gpurun_out/${R}_refresh_$wl.log 2>&1; echo "refresh $wl rc=$?"; done
python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err; echo "default bench rc=$?"
for wl in c5 c3p hdr hdr1 teapot; do
  extra=""; [ $wl = c5 ] && extra="--steps 1 --warmup 1"
  python bench.py --workload $wl --no-cpu-baseline --no-extra $extra > gpurun_out/${R}_bench_$wl.json 2> gpurun_out/${R}_bench_$wl.err; echo "bench $wl rc=$?"
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --workload c3 --steps 2 --warmup 1 > gpurun_out/${R}_bench_c3_2ranks_gloo_one_gpu.json 2> gpurun_out/${R}_bench_2ranks.err; echo "2-rank gloo bench rc=$?"
python tools/fuzz_parity.py 600 5004 > gpurun_out/${R}_fuzz_parity_600_seed5004.txt 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/${R}_fuzz_parity_600_seed5004.txt
{ python tools/bigparity.py cornell_box_statue 400000 512 64 0; python tools/bigparity.py plastic_dragon 871414 512 32 1; } > gpurun_out/${R}_bigparity.txt 2>&1; echo "bigparity rc=$?"; grep -E "pixels differing|rmse" gpurun_out/${R}_bigparity.txt

#!/bin/bash
# FETCH_SIZE calibration + refreshed profiles for c4 / c3 / c2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/fetchsize
cd /tmp && export TMPDIR=/tmp
U=$REPO/rustraytracer_amd/csrc/build/ubench_fetchsize
timeout 300 $U > $REPO/gpurun_out/fetchsize/ubench_stdout.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/fetchsize/p1 -- $U > /dev/null 2>&1; echo "p1 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $REPO/gpurun_out/fetchsize/p2 -- $U > /dev/null 2>&1; echo "p2 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $REPO/gpurun_out/fetchsize/p3 -- $U > /dev/null 2>&1; echo "p3 rc=$?"
cd $REPO
python3 tools/fetchsize_factors.py gpurun_out/fetchsize/ubench_stdout.txt gpurun_out/fetchsize/p1 gpurun_out/fetchsize/p2 gpurun_out/fetchsize/p3 > gpurun_out/fetchsize/factors.json
cat gpurun_out/fetchsize/factors.json | grep -v "bytes_as\|requested_bytes" | head -80
for tag in c4 c3 c2; do
  timeout 1500 bash tools/refresh_profiles.sh $tag > gpurun_out/refresh_$tag.log 2>&1
  tail -c 300 gpurun_out/refresh_$tag.log; echo
done

#!/bin/bash
# On the GPU box: rocprofv3 PMC passes over one bench step (separate passes, no tracing domains
# besides --kernel-trace), summarised per kernel into gpurun_out/pmc_summary.json
# usage: tools/prof_pmc.sh <tag> [bench args]
TAG=${1:-c2}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_pass() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra ${BENCH_ARGS} > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
BENCH_ARGS="$*"
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_FLAT
run_pass tcc1 FETCH_SIZE TCC_HIT_sum
run_pass tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
run_pass grbm GRBM_GUI_ACTIVE
python3 $REPO/tools/pmc_summary.py $OUT > $REPO/gpurun_out/pmc_summary_$TAG.json
cat $REPO/gpurun_out/pmc_summary_$TAG.json | head -c 6000

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
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra ${BENCH_ARGS} > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
BENCH_ARGS="$*"
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_FLAT
run_pass tcc1 FETCH_SIZE TCC_HIT_sum
run_pass tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
# LDS traffic of the kernels (traversal stack, staged top-of-tree nodes, k_shade's dealing): instructions, busy / wait
# cycles, bank conflicts.  A pass whose counter names this rocprofv3 does not know fails on its own (rc != 0).
run_pass lds1 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run_pass lds2 SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_FLAT_LDS_ONLY
run_pass grbm GRBM_GUI_ACTIVE
python3 $REPO/tools/pmc_summary.py $OUT > $REPO/gpurun_out/pmc_summary_$TAG.json
cat $REPO/gpurun_out/pmc_summary_$TAG.json | head -c 6000

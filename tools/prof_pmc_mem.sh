#!/bin/bash
# On the GPU box: PMC passes that look at the vector-memory pipeline (TA / TCP) and the issue mix of the kernels.
# Same rules as tools/prof_pmc.sh (separate passes, --kernel-trace only).  Output: gpurun_out/pmc_mem_<tag>.json
# usage: tools/prof_pmc_mem.sh <tag> [bench args]
TAG=${1:-c2}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcmem_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS="$*"
run_pass() {
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra ${BENCH_ARGS} > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
run_pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS
run_pass sq3 SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64
#(aborts in rocprofv3 on this image, then hangs until the limit) run_pass ta1 TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum
#run_pass tcp1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
#run_pass tcp2 TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
#run_pass tcp3 TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
run_pass grbm GRBM_GUI_ACTIVE
python3 $REPO/tools/pmc_summary.py $OUT > $REPO/gpurun_out/pmc_mem_$TAG.json
grep -c . $REPO/gpurun_out/pmc_mem_$TAG.json

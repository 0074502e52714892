#!/bin/bash
# On the GPU box: texture-addresser / L1 (TA, TCP) counters of one bench step per kernel -- is k_trace bound by the
# vector-memory pipeline?  Separate --pmc passes, kernel trace only.  usage: tools/prof_ta.sh <tag> [bench args]
TAG=${1:-c3}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/ta_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "\b\(TA\|TCP\|TD\)_[A-Z0-9_a-z]*" | sort -u > $OUT/avail.txt
BENCH_ARGS="$*"
run_pass() {
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra ${BENCH_ARGS} > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
run_pass ta1 TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE
run_pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
run_pass ta3 TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum
run_pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run_pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
run_pass tcp3 TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        agg[k]["calls_" + r["Counter_Name"]] += 1
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        if not c.startswith("calls_"):
            print("   %-44s %.4g" % (c, x))
PY

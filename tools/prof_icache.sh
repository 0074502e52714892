#!/bin/bash
# On the GPU box: instruction-cache counters per kernel over one bench step.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_icache
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/ic -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/ic.log 2>&1; echo "ic rc=$?"
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQC_TC_INST_REQ SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/if -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/if.log 2>&1; echo "if rc=$?"
python3 - <<PY
import csv,glob,collections
for d in ("ic","if"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float))
    for p in glob.glob("$OUT/%s/*/*counter_collection.csv"%d):
        for r in csv.DictReader(open(p)):
            k=r["Kernel_Name"].split("(")[0].replace("void ","")
            if k.startswith("rtd::"): acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in acc.items(): print(d,k,dict(v))
PY

#!/bin/bash
# On the GPU box: everything profiles/ is built from, for one workload tag.
#   1. rocprofv3 --kernel-trace --stats of the default bench command (no cpu baseline)
#   2. the PMC passes (tools/prof_pmc.sh)
#   3. the full bench line (with cpu_baseline)
# Outputs land in gpurun_out/refresh_<tag>/ ; tools/collect_profiles.py copies them into profiles/.
TAG=${1:-c2}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/refresh_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --no-cpu-baseline --no-extra --workload $TAG "$@" > $OUT/stats_bench.json 2> $OUT/stats_bench.err
echo "stats rc=$?"
$REPO/tools/prof_pmc.sh $TAG --workload $TAG "$@" > $OUT/pmc.log 2>&1
echo "pmc rc=$?"
cp $REPO/gpurun_out/pmc_summary_$TAG.json $OUT/pmc_summary.json
cd $REPO
python3 tools/collect_profiles.py --stage $TAG   # writes profiles/trace_pmc_<tag>.json on the box so the bench line below carries traffic
python3 bench.py --workload $TAG --no-extra "$@" > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench rc=$?"
cp profiles/trace_pmc_$TAG.json $OUT/
cat $OUT/bench_full.json

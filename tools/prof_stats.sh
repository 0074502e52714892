#!/bin/bash
# On the GPU box: rocprofv3 kernel statistics of a short bench run, printed as a table.  usage: tools/prof_stats.sh <workload> [bench args]
WL=${1:-c4}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/stats_$WL
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/bench.py --no-cpu-baseline --no-extra --workload $WL --steps 2 --warmup 1 "$@" > $OUT/bench.json 2> $OUT/bench.err
echo "rc=$?"
python3 - $OUT <<'PY'
import csv,glob,sys,re,os
f=sorted(glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True),key=os.path.getmtime)
rows=list(csv.DictReader(open(f[-1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:12]:
    n=re.sub(r'\(.*','',r['Name'])[:44]
    print('%-44s calls %5s total %9.2f ms avg %9.1f us %5.1f%%'%(n,r['Calls'],float(r['TotalDurationNs'])/1e6,float(r['AverageNs'])/1e3,100*float(r['TotalDurationNs'])/tot))
PY

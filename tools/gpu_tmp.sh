#!/bin/bash
# PMC passes only (the kernel stats and bench lines of this state are already collected)
for tag in c4 c3 c2; do
  tools/prof_pmc.sh $tag --workload $tag > gpurun_out/pmc_$tag.log 2>&1
  mkdir -p gpurun_out/refresh_$tag
  cp gpurun_out/pmc_summary_$tag.json gpurun_out/refresh_$tag/pmc_summary.json
  python3 tools/collect_profiles.py --stage $tag
  cp profiles/trace_pmc_$tag.json gpurun_out/refresh_$tag/
  python3 bench.py --workload $tag --no-extra > gpurun_out/refresh_$tag/bench_full.json 2> gpurun_out/refresh_$tag/bench_full.err
  tail -c 150 gpurun_out/refresh_$tag/bench_full.json; echo
done

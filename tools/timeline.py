#!/usr/bin/env python3
"""Print the kernel timeline of the first render in a rocprofv3 --kernel-trace csv directory."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
seq = [(r['Kernel_Name'].split('(')[0].replace('void ', '').replace('rtd::', ''),
        (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows]
i0 = [i for i, (n, _) in enumerate(seq) if n == 'k_plan'][0]
res = [i for i, (n, _) in enumerate(seq) if n == 'k_resolve'][0]
print(" ".join(f"{n.replace('k_', '')}:{d:.0f}" for n, d in seq[i0:res + 1]))
tot = collections.Counter()
for n, d in seq[i0:res + 1]:
    tot[n] += d
print({k: round(v) for k, v in tot.items()})

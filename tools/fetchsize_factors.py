#!/usr/bin/env python3
"""FETCH_SIZE calibration: tools/ubench_fetchsize.hip's known byte counts against rocprofv3's counters.
usage: fetchsize_factors.py <ubench stdout> <rocprof pmc dir> [<second pmc dir> ...]  -> JSON on stdout"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

known = {}
for line in open(sys.argv[1]):
    if line.startswith("UBENCH "):
        f = line.split()
        # the kernel name may contain a space ("k_node<7, 1>"): fields are name..., then key value pairs
        i = f.index("requested_bytes")
        name = " ".join(f[1:i])
        kv = dict(zip(f[i::2], f[i + 1::2]))
        known[name] = {k: float(v) for k, v in kv.items()}
ctr = defaultdict(lambda: defaultdict(float))
for root in sys.argv[2:]:
    for p in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            ctr[k][r["Counter_Name"]] += float(r["Counter_Value"])
out = {}
for name, kb in known.items():
    c = ctr.get(name) or ctr.get(name.replace(", ", ","), {})
    row = dict(kb)
    row.update(c)
    if "FETCH_SIZE" in c and c["FETCH_SIZE"] > 0:
        fs = c["FETCH_SIZE"] * 1024.0
        row["FETCH_SIZE_bytes"] = fs
        row["factor_vs_requested"] = kb["requested_bytes"] / fs
        row["factor_vs_64B_sectors"] = kb["bytes_as_64B_sectors"] / fs
        row["factor_vs_128B_lines"] = kb["bytes_as_128B_lines"] / fs
        row["GBps_requested"] = kb["requested_bytes"] / kb["ms"] / 1e6
    out[name] = row
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (one directory per pass) into per-kernel sums."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
dur = defaultdict(float)
for p in sorted(glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv"))):
    seen = set()
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
for p in sorted(glob.glob(os.path.join(root, "sq1", "*", "*kernel_trace.csv"))):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        calls[k] += 1
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
res = {}
for k in out:
    if not (k.startswith("rtd::") or k.startswith("rtd32::")):
        continue
    d = dict(out[k])
    d["calls"] = calls.get(k, 0)
    d["total_ms_profiled"] = dur.get(k, 0.0)
    # derived, with the gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests as 64 B)
    if "FETCH_SIZE" in d:
        d["hbm_read_bytes_fetch_size_x1024"] = d["FETCH_SIZE"] * 1024
        d["hbm_read_bytes_corrected_x2"] = d["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in d:
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d and d["TCC_HIT_sum"] + d["TCC_MISS_sum"] > 0:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    if "SQ_THREAD_CYCLES_VALU" in d and d.get("SQ_ACTIVE_INST_VALU", 0) > 0:
        d["valu_lane_utilisation"] = d["SQ_THREAD_CYCLES_VALU"] / (d["SQ_ACTIVE_INST_VALU"] * 64 / 4) if False else None
    if d.get("SQ_WAVE_CYCLES", 0) > 0:
        for c in ("SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY"):
            if c in d:
                d[c + "_per_wave_cycle"] = d[c] / d["SQ_WAVE_CYCLES"]
    res[k] = d
print(json.dumps(res, indent=1))

#!/usr/bin/env python3
"""Per launch: HBM-side traffic (rocprofv3 --pmc FETCH_SIZE x 2 on gfx950, WRITE_SIZE) over the launch's duration in the
counter pass, for the first batch of the TIMED step of tools/prof_pmc.sh's command (bench.py --steps 1 --warmup 0).
rocprofv3 serialises the dispatches while it counts, so every kernel is measured ALONE (in the product the class kernels
run two at a time and the light kernel beside them and the next k_trace: DESIGN.md 12.8).

    python tools/hbm_per_launch.py gpurun_out/pmc_c4 > profiles/rNN_hbm_per_launch_c4.txt
"""
import csv
import glob
import os
import re
import sys

root = sys.argv[1]


def load(pass_name, counter):
    f = sorted(glob.glob(os.path.join(root, pass_name, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1]
    out = []
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            out.append((int(r["Dispatch_Id"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", ""), float(r["Counter_Value"]),
                        int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    out.sort()
    return out


rd = load("tcc1", "FETCH_SIZE")
wr = load("tcc2", "WRITE_SIZE")
assert [x[1] for x in rd] == [x[1] for x in wr], "the two passes dispatched different kernels"
# the timed step = the dispatches before the first counting k_trace<true, ...>; its first batch = up to the first k_resolve
rows = []
for (i, name, fs, t0, t1), (_, _, ws, _, _) in zip(rd, wr):
    if "k_trace<true" in name:
        break
    rows.append((name, fs * 2048.0, ws * 1024.0, (t1 - t0) / 1e6))
first = next(i for i, r in enumerate(rows) if "k_resolve" in r[0])
rows = rows[:first + 1]
print("# C4, first batch of the timed step (2^30 camera samples through a 2^28-path pool): per launch, HBM-side bytes and the launch's")
print("# duration ALONE (counter passes serialise the dispatches); share = bytes / duration / 8 TB/s")
print("%-34s %-5s %8s %8s %8s %7s %6s" % ("kernel", "it", "read_GB", "write_GB", "ms", "TB/s", "share"))
it = -1
for name, r, w, ms in rows:
    if "k_plan" in name:
        it += 1
        continue
    if ms < 0.02 and r + w < 1e8:
        continue
    short = name.replace("rtd::", "").replace("rtd32::", "")
    tbs = (r + w) / (ms / 1e3) / 1e12 if ms > 0 else 0.0
    print("%-34s it%-3d %8.1f %8.1f %8.2f %7.2f %5.1f %%" % (short[:34], it, r / 1e9, w / 1e9, ms, tbs, 100.0 * tbs / 8.0))

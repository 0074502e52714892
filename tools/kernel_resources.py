#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from `make -C rustraytracer_amd/csrc asm` remarks.
usage: make -C rustraytracer_amd/csrc asm 2>&1 | python tools/kernel_resources.py [filter]"""
import re
import subprocess
import sys

flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]+\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
names = list(rows)
try:
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, stdout=subprocess.PIPE, text=True).stdout.split("\n")
except Exception:
    dem = names
print("%-58s %5s %5s %6s %7s %4s %6s" % ("kernel", "VGPR", "AGPR", "spillV", "scratch", "occ", "LDS"))
for n, d in zip(names, dem):
    r = rows[n]
    short = re.sub(r"\(.*", "", d).replace("rtd::", "")
    if flt and flt not in short:
        continue
    print("%-58s %5d %5d %6d %7d %4d %6d" % (short[:58], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", -1),
                                          r.get("ScratchSize", -1), r.get("Occupancy", -1), r.get("LDS Size", -1)))

#!/usr/bin/env python3
"""Register / scratch / LDS use of the kernels inside a built librt_amd.so (or a variant), read from the code
object's own metadata note -- no recompilation.  usage: python tools/kernel_resources_so.py [lib.so] [filter]"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_hash import _device_images, ROOT

lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "rustraytracer_amd", "librt_amd.so")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blob = open(lib, "rb").read()
print("%-64s %5s %5s %6s %7s %6s" % ("kernel", "VGPR", "AGPR", "spillV", "scratch", "LDS"))
for img in _device_images(blob):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(img)
        f.flush()
        txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], stdout=subprocess.PIPE, text=True).stdout
    for blk in txt.split("  - .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "-1"])[1]
        name = g("name")
        try:
            dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
        except Exception:
            dem = name
        short = re.sub(r"\(.*", "", dem).replace("void ", "")
        if flt and flt not in short:
            continue
        agpr = re.match(r"\s*(\d+)", blk)
        print("%-64s %5s %5s %6s %7s %6s" % (short[:64], g("vgpr_count"), agpr.group(1) if agpr else "?", g("vgpr_spill_count"),
                                            g("private_segment_fixed_size"), g("group_segment_fixed_size")))

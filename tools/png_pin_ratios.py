#!/usr/bin/env python3
"""Per-region ratios of the current kernels against the reference's examples/cornell_statue.png (the statistical pin of
tests/test_reference_png_pin.py), printed instead of asserted.  usage: python tools/png_pin_ratios.py [gpu]
  (no argument: the CPU oracle at 270x270 @ 64 spp; `gpu`: the HIP path at 540x540 @ 256 spp, on an MI355X)"""
import ctypes as C
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import rustraytracer_amd as rr
from tests import png_pin as PP
from tests.test_reference_png_pin import FIX, _compare

fix = json.load(open(FIX))
tmp = tempfile.mkdtemp()
sc = rr.cornell_box_statue(mesh_path=PP.statue_proxy_obj(os.path.join(tmp, "proxy.obj")), variant=1)
if len(sys.argv) > 1 and sys.argv[1] == "gpu":
    ctx = rr.Context(0)
    gs = ctx.upload(sc)
    rgb, n, st = ctx.render(gs, sc.camera, rr.make_cfg(540, 540, 256, seed=0))
    img = ctx.resolve_rgb8(rgb, n)
    what = "HIP path, 540x540 @ 256 spp"
else:
    from tests import oracle_ffi as O
    W = H = 270
    rgb, n, _ = O.OracleScene(sc).render(sc.camera, rr.make_cfg(W, H, 64, seed=0), O.ORDERED, os.cpu_count() or 8)
    img = np.zeros((H, W, 3), dtype=np.uint8)
    O.lib().oracle_resolve_rgb8(rgb.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p), W * H, img.ctypes.data_as(C.c_void_p))
    what = "CPU oracle, 270x270 @ 64 spp"
worst = _compare(img, fix)
print(f"# {what} against examples/cornell_statue.png (sha256 {fix['pictures']['cornell_statue'].get('sha256', '?')[:16]}...)")
print("# region: worst channel |linear ratio - 1| (tolerance); regions with tolerance 0 are compared as 8-bit means (<= 0.5)")
for name, w in worst.items():
    print(f"{name:28s} {w:8.4f}   (tol {fix['tolerance_linear_rel'][name]})")

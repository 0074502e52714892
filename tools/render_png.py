"""Render presets on the GPU and save tone-mapped PNGs (rows f1 + f4).  Run on the GPU box:
   python tools/render_png.py [size] [spp]  -> gpurun_out/renders/*.png
The library's own writer (rrh_write_png, stored deflate) produces the file; it is then re-packed with zlib
level 9 so that the copies committed under profiles/renders/ stay small."""
import os
import struct
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustraytracer_amd as rr
from tests.test_host_cpu import _read_png


def repack(path):
    img = _read_png(path)
    h, w, _ = img.shape
    raw = np.concatenate([np.zeros((h, 1), np.uint8), img.reshape(h, w * 3)], axis=1).tobytes()

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
out = "gpurun_out/renders"
os.makedirs(out, exist_ok=True)
ctx = rr.Context(0)
CASES = [("cornell_statue_matte", lambda: rr.cornell_box_statue(mesh_faces=400000, variant=0), 1.0),
         ("dragon_metal", lambda: rr.plastic_dragon(mesh_faces=871414, variant=1), 1.0),
         ("two_dragons", lambda: rr.two_dragons(16 / 9, mesh_faces=871414, variant=0), 16 / 9),
         ("material_hdr_rough_glass", lambda: rr.material_hdr(3, mesh_faces=150000), 1.0),
         ("material_hdr_rosegold", lambda: rr.material_hdr(1, mesh_faces=150000), 1.0)]
for name, make, aspect in CASES:
    sc = make()
    gs = ctx.upload(sc)
    w, h = int(size * aspect), size
    t0 = time.time()
    rgb, n, st = ctx.render(gs, sc.camera, rr.make_cfg(w, h, spp))
    dt = time.time() - t0
    p = f"{out}/{name}_{w}x{h}_{spp}spp.png"
    rr.write_png(p, ctx.resolve_rgb8(rgb, n))
    repack(p)
    print(f"{name}: {w}x{h} @ {spp} spp, {st.rays / 1e6:.0f} Mrays in {dt:.2f} s -> {p} ({os.path.getsize(p) // 1024} KiB)", flush=True)
    gs.close()

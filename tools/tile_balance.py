#!/usr/bin/env python3
"""Work balance of the tile -> rank assignment on one GPU: renders the workload once per (rank, world) at a reduced
spp and prints rays and device time per rank (max / mean = the strong-scaling loss the split alone causes).
usage: python tools/tile_balance.py [workload] [spp] [world ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import rustraytracer_amd as rr  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
worlds = [int(x) for x in sys.argv[3:]] or [2, 4, 8]
preset, kw, W, H, _, desc = bench.WORKLOADS[name]
scene = rr.Scene(preset, W / H, **kw)
ctx = rr.Context(0)
gs = ctx.upload(scene)
d_rgb = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
d_n = torch.zeros((H, W), dtype=torch.int32, device="cuda")
for world in worlds:
    rays, ms = [], []
    for r in range(world):
        cfg = rr.make_cfg(W, H, spp, seed=0, tile_rank=r, tile_world=world)
        ctx.render_device(gs, scene.camera, cfg, d_rgb.data_ptr(), d_n.data_ptr())  # warm
        st = ctx.render_device(gs, scene.camera, cfg, d_rgb.data_ptr(), d_n.data_ptr())
        torch.cuda.synchronize()
        rays.append(st.rays)
        ms.append(st.kernel_ms)
    mr, mm = sum(rays) / world, sum(ms) / world
    print(f"{name} @ {spp} spp, world {world}: rays max/mean {max(rays) / mr:.4f} min/mean {min(rays) / mr:.4f}; "
          f"device ms max/mean {max(ms) / mm:.4f} ({' '.join('%.1f' % m for m in ms)})", flush=True)
gs.close()
ctx.close()

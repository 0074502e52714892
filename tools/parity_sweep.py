"""Larger-than-test parity sweep over every shading-kernel instance (run on the GPU box): GPU film == oracle film."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustraytracer_amd as rr
from tests import oracle_ffi as O
ctx = rr.Context(0)
cases = [("c3 metal", lambda: rr.plastic_dragon(mesh_faces=871414, variant=1), 384, 32),
         ("c5 glass", lambda: rr.plastic_dragon(mesh_faces=871414, variant=2), 256, 32),
         ("plastic", lambda: rr.plastic_dragon(mesh_faces=300000, variant=0), 384, 16),
         ("statue plastic", lambda: rr.cornell_box_statue(mesh_faces=100000, variant=3), 256, 32),
         ("hdr rough glass", lambda: rr.material_hdr(3, mesh_faces=100000), 256, 32),
         ("hdr plastic", lambda: rr.material_hdr(0, mesh_faces=100000), 256, 32),
         ("spheres", lambda: rr.sphere_roughness(), 384, 32)]
for name, make, W, spp in cases:
    sc = make(); gs = ctx.upload(sc); cfg = rr.make_cfg(W, W, spp, seed=3)
    rg, ng, sg = ctx.render(gs, sc.camera, cfg)
    ro, no, so = O.OracleScene(sc).render(sc.camera, cfg, O.ORDERED, 16)
    same = np.array_equal(rg, ro, equal_nan=True) and np.array_equal(ng, no)
    print(name, "rays", sg.rays, so.rays, "identical", same, "counts", (sg.rays_extension, sg.rays_shadow, sg.rays_probe) == (so.rays_extension, so.rays_shadow, so.rays_probe), flush=True)
    gs.close()

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustraytracer_amd as rr
faces = int(sys.argv[1]) if len(sys.argv) > 1 else 871414
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
sc = rr.plastic_dragon(mesh_faces=faces, variant=1)
ctx = rr.Context(0); gs = ctx.upload(sc)
prev = None
for pif in (0, 0, 1 << 22, 1 << 20):
    r, n, st = ctx.render(gs, sc.camera, rr.make_cfg(W, W, spp, paths_in_flight=pif))
    h = hash(r.tobytes())
    print("pif", pif, "rays", st.rays_extension, st.rays_shadow, st.rays_probe, "vertices", st.vertices_shaded, "film hash", h, "nan", int(np.isnan(r).sum()), flush=True)

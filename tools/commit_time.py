"""Host commit time of the bench scenes (rt_scene_commit: BVH build + leaf layout + uploads), three commits each.
usage: python tools/commit_time.py   (on the GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustraytracer_amd as rr

ctx = rr.Context(0)
for name, make in (("c2", lambda: rr.cornell_box_statue(mesh_faces=400000, variant=0)),
                   ("c3", lambda: rr.plastic_dragon(mesh_faces=871414, variant=1)),
                   ("c4", lambda: rr.two_dragons(1920 / 1080, mesh_faces=871414, variant=0))):
    sc = make()
    ts = []
    for k in range(3):
        t0 = time.time()
        gs = ctx.upload(sc)
        ts.append((time.time() - t0) * 1e3)
        info = gs.info()
        gs.close()
    print(name, "n_prims", info["n_prims"], "commit wall ms", " ".join("%.0f" % t for t in ts), "build_ms(last)", round(info["build_ms"]), flush=True)

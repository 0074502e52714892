"""Row f3 quality sweep of the device builder's knobs on one scene (run on the GPU box).
usage: python tools/lbvh_sweep.py [c2|c3|c4] [bottom|ploc]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustraytracer_amd as rr

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
make, W, H, spp = {"c2": (lambda: rr.cornell_box_statue(mesh_faces=400000, variant=0), 512, 512, 32),
                   "c3": (lambda: rr.plastic_dragon(mesh_faces=871414, variant=1), 1024, 1024, 8),
                   "c4": (lambda: rr.two_dragons(1920 / 1080, mesh_faces=871414, variant=0), 1920, 1080, 4)}[which]
sc = make()
ctx = rr.Context(0)
def run(label, dev, env):
    for k, v in env.items():
        os.environ[k] = str(v)
    gs = ctx.upload(sc, device_build=dev)
    inf = gs.info()
    cfg = rr.make_cfg(W, H, spp)
    ctx.render(gs, sc.camera, cfg)
    _, _, st = ctx.render(gs, sc.camera, cfg)
    _, _, stc = ctx.render(gs, sc.camera, rr.make_cfg(W, H, spp, count_traversal=True))
    print("%s %-34s build_dev_ms %6.1f nodes %7d depth %2d nodes/ray %.3f tris/ray %.3f trace_ms %.2f" % (
        which, label, inf["build_device_ms"], inf["n_bvh_nodes"], inf["bvh_depth"], stc.nodes_fetched / stc.rays,
        stc.tris_tested / stc.rays, st.trace_ms), flush=True)
    gs.close()
mode = sys.argv[2] if len(sys.argv) > 2 else "bottom"
run("host_sah", False, {})
if mode == "ploc":
    for radius in (8, 16, 32):
        for rot in (0, 1, 2):
            run(f"ploc radius {radius} rot {rot}", True, {"RT_DEVICE_BUILDER": "ploc", "RT_PLOC_RADIUS": radius, "RT_PLOC_ROTATE_PASSES": rot})
    os.environ["RT_DEVICE_BUILDER"] = "lbvh"
    for rot in (2,):
        for sah in (0, 256, 64):
            run(f"lbvh rot{rot} sah_cluster {sah}", True, {"RT_LBVH_ROTATE_PASSES": rot, "RT_LBVH_SAH_CLUSTER": sah, "RT_LBVH_SAH_BOTTOM": 0})
    run("lbvh plain", True, {"RT_LBVH_ROTATE_PASSES": 0, "RT_LBVH_SAH_CLUSTER": 0})
else:  # the SAH bottom (kb_cluster_sah) against the Morton-order bottom, by cluster size and rotation passes
    os.environ["RT_DEVICE_BUILDER"] = "lbvh"
    run("lbvh cluster 256 morton bottom rot 2", True, {"RT_LBVH_ROTATE_PASSES": 2, "RT_LBVH_SAH_CLUSTER": 256, "RT_LBVH_SAH_BOTTOM": 0})
    for sah in (128, 256, 512, 1024):
        for rot in (0, 1, 2):
            run(f"lbvh cluster {sah} sah bottom rot {rot}", True,
                {"RT_LBVH_ROTATE_PASSES": rot, "RT_LBVH_SAH_CLUSTER": sah, "RT_LBVH_SAH_BOTTOM": 1})

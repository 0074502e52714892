"""Row f3: BVH build time and traversal quality, host binned SAH vs device LBVH (run on the GPU box)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustraytracer_amd as rr

CASES = [("c2", lambda: rr.cornell_box_statue(mesh_faces=400000, variant=0), 512, 512, 64),
         ("c3", lambda: rr.plastic_dragon(mesh_faces=871414, variant=1), 1024, 1024, 16),
         ("c4", lambda: rr.two_dragons(1920 / 1080, mesh_faces=871414, variant=0), 1920, 1080, 8)]
ctx = rr.Context(0)
out = []
for name, make, W, H, spp in CASES:
    sc = make()
    row = {"scene": name}
    # device builder: plain Morton-order tree, and with 1 / 2 / 3 refit passes that apply tree rotations (bvh_gpu.hip)
    # also: the SAH top over clusters of <= N primitives (RT_LBVH_SAH_CLUSTER, 0 = none; default 256)
    for label, dev, rot, sah in (("host_sah", False, None, None), ("device_lbvh_plain", True, 0, 0), ("device_lbvh_rot2", True, 2, 0),
                                 ("device_lbvh_rot2_sah1024", True, 2, 1024), ("device_lbvh", True, 2, 256),
                                 ("device_lbvh_rot2_sah64", True, 2, 64), ("device_lbvh_rot3_sah64", True, 3, 64)):
        if rot is not None:
            os.environ["RT_LBVH_ROTATE_PASSES"] = str(rot)
            os.environ["RT_LBVH_SAH_CLUSTER"] = str(sah)
        t0 = time.time()
        gs = ctx.upload(sc, device_build=dev)
        wall = (time.time() - t0) * 1e3
        inf = gs.info()
        cfg = rr.make_cfg(W, H, spp)
        ctx.render(gs, sc.camera, cfg)  # warm
        t0 = time.time()
        _, _, st = ctx.render(gs, sc.camera, cfg)
        dt = time.time() - t0
        _, _, stc = ctx.render(gs, sc.camera, rr.make_cfg(W, H, spp, count_traversal=True))
        row[label] = {"upload_wall_ms": wall, "build_ms": inf["build_ms"], "build_device_ms": inf["build_device_ms"],
                      "n_prims": inf["n_prims"], "nodes": inf["n_bvh_nodes"], "depth": inf["bvh_depth"],
                      "Mrays_s": st.rays / dt / 1e6, "trace_ms": st.trace_ms, "kernel_ms": st.kernel_ms,
                      "nodes_per_ray": stc.nodes_fetched / stc.rays, "tris_per_ray": stc.tris_tested / stc.rays}
        gs.close()
    for k in row:
        if k.startswith("device"):
            row[k]["nodes_per_ray_vs_host"] = row[k]["nodes_per_ray"] / row["host_sah"]["nodes_per_ray"]
            row[k]["trace_ms_vs_host"] = row[k]["trace_ms"] / row["host_sah"]["trace_ms"]
    out.append(row)
    print(json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/build_bench.json", "w"), indent=1)

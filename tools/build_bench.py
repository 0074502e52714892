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
    # device builder (bvh_gpu.hip): the plain Morton-order tree; + 2 refit passes with tree rotations; + the host-built SAH top
    # over clusters of <= 256 primitives (the default until the SAH bottom existed); the default: clusters of <= 1024,
    # SAH top and SAH bottom, no rotations; the same with one rotation pass / with clusters of 256
    KNOBS = ("RT_LBVH_ROTATE_PASSES", "RT_LBVH_SAH_CLUSTER", "RT_LBVH_SAH_BOTTOM")
    for label, dev, env in (("host_sah", False, {}),
                            ("device_lbvh_plain", True, {"RT_LBVH_ROTATE_PASSES": 0, "RT_LBVH_SAH_CLUSTER": 0}),
                            ("device_lbvh_rot2", True, {"RT_LBVH_ROTATE_PASSES": 2, "RT_LBVH_SAH_CLUSTER": 0}),
                            ("device_lbvh_rot2_sahtop256", True, {"RT_LBVH_ROTATE_PASSES": 2, "RT_LBVH_SAH_CLUSTER": 256, "RT_LBVH_SAH_BOTTOM": 0}),
                            ("device_lbvh", True, {}),
                            ("device_lbvh_rot1", True, {"RT_LBVH_ROTATE_PASSES": 1}),
                            ("device_lbvh_cluster256", True, {"RT_LBVH_SAH_CLUSTER": 256})):
        for k in KNOBS:
            os.environ.pop(k, None)
        for k, v in env.items():
            os.environ[k] = str(v)
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

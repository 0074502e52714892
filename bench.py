#!/usr/bin/env python3
"""bench.py -- Mrays/s of the path-tracing hot path on N MI355X (one process per GPU).

A step = one full render of the workload through the C ABI (rt_render_device): scene, BVH and
film buffers already resident in HBM, film left on the device, then (N > 1) the RCCL gather of
the per-rank films to rank 0.  Rays = R1 + R2 + R3 root closest-hit queries,
counted by the device and identical to the CPU oracle's counters.

N = 1 workload: BASELINE.json configs[1] -- Cornell box + triangle-mesh statue (procedural
P-400k stand-in for the missing data/statue.obj), Lambertian, 512x512 @ 64 spp, max_depth 25.
N > 1: weak scaling -- the same image at N times the samples per pixel (64*N spp), its 16x16
tiles interleaved over the ranks (tile k -> rank k % N), so every rank keeps 512*512*64 paths;
each rank's own tiles are packed and gathered to rank 0 over RCCL/xGMI (rustraytracer_amd/dist.py).

One JSON line on rank 0 with `roofline` (dominant kernel k_trace, algorithmic bytes from the
device's traversal counters over HIP-event kernel time) and `cpu_baseline` (the oracle in
reference-shaped mode on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)

WORKLOADS = {
    # name: (preset, kwargs, W, H, spp, description)
    "c2": ("cornell_box_statue", dict(mesh_faces=400000, variant=0), 512, 512, 64,
           "C2 cornell_box_statue (matte) + procedural P-400k mesh, 512x512 @ 64 spp, max_depth 25"),
    "c3": ("plastic_dragon", dict(mesh_faces=871414, variant=1), 1024, 1024, 256,
           "C3 dragon (procedural P-871k) microfacet metal, 1024x1024 @ 256 spp, max_depth 25"),
    "c3p": ("plastic_dragon", dict(mesh_faces=871414, variant=0), 1024, 1024, 256,
            "plastic_dragon() as committed in scenes.rs:310-375 (plastic, two lobes), procedural P-871k, 1024x1024 @ 256 spp"),
    "c4": ("two_dragons", dict(mesh_faces=871414, variant=0), 1920, 1080, 1024,
           "C4 two_dragons (glass + metal, 2 x procedural P-871k), 1920x1080 @ 1024 spp, max_depth 25"),
    "c5": ("plastic_dragon", dict(mesh_faces=871414, variant=2), 2048, 2048, 4096,
           "C5 dragon (procedural P-871k) smooth glass, 2048x2048 @ 4096 spp, max_depth 25"),
    "hdr": ("material_hdr", dict(variant=3, mesh_faces=150000), 512, 512, 64,
            "row f4: material_hdr(3) rough glass under the procedural environment map, 3 x P-150k, 512x512 @ 64 spp"),
    "hdr1": ("material_hdr", dict(variant=1, mesh_faces=150000), 512, 512, 64,
             "row f4: material_hdr(1) rose-gold metal under the procedural environment map, 3 x P-150k, 512x512 @ 64 spp"),
    "c1": ("cornell_box", dict(), 256, 256, 16, "C1 cornell_box 256x256 @ 16 spp"),
    "tiny": ("cornell_box_statue", dict(mesh_faces=20000, variant=0), 128, 128, 8, "tiny smoke workload"),
}


def algorithmic_bytes(st, info):
    """Bytes the traversal kernel requests (cache hits included), f64 parity layout:
    128 B per BVH4 node fetched, 76 B per triangle tested (72 B vertices + 4 B id), 52 B per
    sphere/rect tested (48 B parameters in the leaf slot + 4 B id), 56 B per ray (24 B origin +
    24 B dir/target + 4 B queue entry + 4 B result)."""
    rays = st.rays_extension + st.rays_shadow + st.rays_probe
    return (info["node_bytes"] * st.nodes_fetched + info["tri_bytes"] * st.tris_tested +
            info["other_bytes"] * st.others_tested + 56 * rays)


def cpu_baseline(scene, W, H, target_s=14.0):
    """Oracle, reference-shaped (exhaustive traversal, 6 threads = consts.rs:8 NUM_THREADS), on a
    bounded sample of the same workload: the full image at a reduced spp chosen for ~target_s."""
    import rustraytracer_amd as rr
    from tests import oracle_ffi as O
    osc = O.OracleScene(scene)
    threads = 6
    share = min(os.cpu_count() or 1, 16)  # the GPU box gives 16 host cores per GPU
    t0 = time.time()
    _, _, st = osc.render(scene.camera, rr.make_cfg(W, H, 1, seed=0), O.EXHAUSTIVE, threads)
    dt = max(time.time() - t0, 1e-3)
    spp = 1
    while spp * 2 * dt <= target_s and spp < 64:
        spp *= 2
    if spp > 1:
        t0 = time.time()
        _, _, st = osc.render(scene.camera, rr.make_cfg(W, H, spp, seed=0), O.EXHAUSTIVE, threads)
        dt = time.time() - t0
    rays = st.rays_extension + st.rays_shadow + st.rays_probe
    out = {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": f"the workload's full {W}x{H} image at {spp} spp (throughput is linear in spp), oracle in "
                     f"reference-shaped mode (exhaustive BVH traversal, hittable.rs:591-634, 16x16 tile workers), "
                     f"{rays} rays in {dt:.1f} s",
           "host_cores_share": share}
    # the same sample on the whole CPU share, reference-shaped and with the oracle's pruned traversal
    t0 = time.time()
    _, _, st2 = osc.render(scene.camera, rr.make_cfg(W, H, spp, seed=0), O.EXHAUSTIVE, share)
    dt2 = time.time() - t0
    out["reference_shaped_all_cores"] = {"value": st2.rays / dt2 / 1e6, "cores": share}
    t0 = time.time()
    _, _, st3 = osc.render(scene.camera, rr.make_cfg(W, H, spp, seed=0), O.ORDERED, share)
    dt3 = time.time() - t0
    out["ordered_all_cores"] = {"value": st3.rays / dt3 / 1e6, "cores": share}
    osc.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--paths-in-flight", type=int, default=0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: dry run of the N>1 control flow with several ranks sharing one GPU (films gathered on the host)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import rustraytracer_amd as rr

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback exists)")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"

    preset, kw, W, H, spp0, desc = WORKLOADS[args.workload]
    spp = spp0 * world  # weak scaling: N times the samples, tiles interleaved over ranks
    scene = rr.Scene(preset, W / H, **kw)
    ctx = rr.Context(local_rank)
    gs = ctx.upload(scene)
    info = gs.info()
    d_rgb = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    d_n = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    cfg = rr.make_cfg(W, H, spp, seed=0, tile_rank=rank, tile_world=world, paths_in_flight=args.paths_in_flight)
    from rustraytracer_amd import dist as rd
    gather = rd.FilmGather(W, H, coll_dev) if world > 1 else None

    def step():
        # on torch's current stream, so that the film zeroing of the next step is ordered after this step's gather
        st = ctx.render_device(gs, scene.camera, cfg, d_rgb.data_ptr(), d_n.data_ptr(),
                               stream=torch.cuda.current_stream().cuda_stream)
        if gather is not None:
            # the path's only exchange step: each rank's own tiles go straight to rank 0 (RCCL over xGMI)
            if coll_dev == "cuda":
                gather.gather(d_rgb, d_n)
            else:  # gloo dry run: stage through the host
                h_rgb, h_n = d_rgb.cpu(), d_n.cpu()
                gather.gather(h_rgb, h_n)
                if rank == 0:
                    d_rgb.copy_(h_rgb)
                    d_n.copy_(h_n)
        return st

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    rays = 0
    trace_ms = 0.0
    launches = 0
    kernel_ms = 0.0
    for _ in range(args.steps):
        st = step()
        rays += st.rays
        trace_ms += st.trace_ms
        kernel_ms += st.kernel_ms
        launches += st.trace_launches
    barrier()
    dt = time.perf_counter() - t0
    # max over ranks of the elapsed time, sum of rays
    tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
    rr_ = torch.tensor([float(rays)], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rr_, op=dist.ReduceOp.SUM)
    dt_max, rays_all = float(tt.item()), float(rr_.item())

    # instrumented pass (outside the timed region): the same kernel and BVH with traversal counters
    cfg_c = rr.make_cfg(W, H, spp, seed=0, tile_rank=rank, tile_world=world, paths_in_flight=args.paths_in_flight,
                        count_traversal=True)
    stc = ctx.render_device(gs, scene.camera, cfg_c, d_rgb.data_ptr(), d_n.data_ptr())
    torch.cuda.synchronize()

    if rank == 0:
        alg = algorithmic_bytes(stc, info)  # per render, this rank
        avg_launch_s = (trace_ms / 1e3) / max(launches, 1)
        launches_per_render = launches / max(args.steps, 1)
        bytes_per_launch = alg / max(launches_per_render, 1)
        achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", f"trace_pmc_{args.workload}.json")
        if world == 1 and os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/sec (primary+secondary)", "value": rays_all / dt_max / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc + (f"; weak scaling: {spp} spp, 16x16 tiles interleaved over {world} ranks, "
                                            "own tiles gathered to rank 0 over RCCL" if world > 1 else ""),
                       "width": W, "height": H, "spp": spp, "max_depth": rr.MAX_DEPTH, "seed": 0,
                       "triangles": info["n_triangles"], "bvh_nodes": info["n_bvh_nodes"],
                       "rays_per_step": rays_all / args.steps, "paths_per_step": W * H * spp,
                       "parallelism": f"tiles{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_trace", "avg_launch_ms": avg_launch_s * 1e3,
                         "launches_per_step": launches_per_render,
                         "algorithmic_bytes_per_step": alg,
                         "bytes_per_ray": alg / max(stc.rays, 1),
                         "nodes_per_ray": stc.nodes_fetched / max(stc.rays, 1),
                         "tris_per_ray": stc.tris_tested / max(stc.rays, 1),
                         "trace_share_of_device_time": trace_ms / max(kernel_ms, 1e-9)},
            "device_ms_per_step": kernel_ms / args.steps,
            # also paths/s and the mean path length (rays per camera sample), whole job
            "mpaths_per_s": W * H * spp * args.steps / dt_max / 1e6,
            "mean_rays_per_path": rays_all / args.steps / (W * H * spp),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, W, H)
        print(json.dumps(out), flush=True)
    gs.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Several ranks (one process each) on ONE GPU: each renders its interleaved tiles through the C ABI
(rt_render_cfg.tile_rank / tile_world), FilmGather (rustraytracer_amd/dist.py) collects the own-tile pixels on rank
0 over gloo, and rank 0 compares the gathered film with its own one-rank render bit for bit.

Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
             --master-port 29517 tools/mp_film_check.py
Prints one JSON line on rank 0; exit code 0 = identical.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist

    import rustraytracer_amd as rr
    from rustraytracer_amd import dist as rd

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # --backend nccl (= RCCL): one GPU per rank (LOCAL_RANK), the film gathered device to device over xGMI -- what
    # bench.py --gpus N does; needs world distinct GPUs.  Default gloo: every rank on GPU 0, films gathered on the host.
    backend = "nccl" if "--backend" in sys.argv and sys.argv[sys.argv.index("--backend") + 1] == "nccl" else "gloo"
    dev = int(os.environ.get("LOCAL_RANK", rank)) if backend == "nccl" else 0
    if backend == "nccl" and torch.cuda.device_count() < world:
        print(f"mp_film_check: --backend nccl needs {world} GPUs, found {torch.cuda.device_count()}", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend=backend)
    W, H, spp = 200, 120, 16
    sc = rr.two_dragons(W / H, mesh_faces=20000)
    ctx = rr.Context(dev)
    gs = ctx.upload(sc)
    d_rgb = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    d_n = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    cfg = rr.make_cfg(W, H, spp, seed=4, tile_rank=rank, tile_world=world)
    st = ctx.render_device(gs, sc.camera, cfg, d_rgb.data_ptr(), d_n.data_ptr(),
                           stream=torch.cuda.current_stream().cuda_stream)
    if backend == "nccl":
        gather = rd.FilmGather(W, H, torch.device("cuda", dev))
        gather.gather(d_rgb, d_n)
        torch.cuda.synchronize()
        h_rgb, h_n = d_rgb.cpu(), d_n.cpu()
        rays = torch.tensor([float(st.rays)], dtype=torch.float64, device="cuda")
        dist.all_reduce(rays)
        rays = rays.cpu()
    else:
        h_rgb, h_n = d_rgb.cpu(), d_n.cpu()
        gather = rd.FilmGather(W, H, "cpu")
        gather.gather(h_rgb, h_n)
        rays = torch.tensor([float(st.rays)], dtype=torch.float64)
        dist.all_reduce(rays)
    ok = True
    if rank == 0:
        full, nfull, sfull = ctx.render(gs, sc.camera, rr.make_cfg(W, H, spp, seed=4))
        ok = bool(np.array_equal(h_rgb.numpy(), full) and np.array_equal(h_n.numpy().astype(np.uint32), nfull)
                  and int(rays.item()) == sfull.rays)
        print(json.dumps({"ranks": world, "backend": dist.get_backend(), "devices": sorted({dev, 0}) if backend == "gloo" else list(range(world)), "bit_identical": ok, "rays_sum": int(rays.item()), "rays_one_rank": sfull.rays,
                          "image": [W, H, spp]}), flush=True)
    gs.close()
    ctx.close()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

"""N > 1 path on CPU: world_size-2 gloo processes, tiles interleaved, film gathered to rank 0.

The per-rank films come from the CPU oracle (test infrastructure); what is under test is the
product's sharding + gather code (rustraytracer_amd/dist.py) and the tile ownership rule that
rt_render_cfg.tile_rank / tile_world implement on the GPU.
"""
import math
import os
import socket
import subprocess
import sys

import numpy as np

import rustraytracer_amd as rr
from rustraytracer_amd import dist as rd
from tests import oracle_ffi as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["RR_ROOT"])
import rustraytracer_amd as rr
from rustraytracer_amd import dist as rd
from tests import oracle_ffi as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
W, H, SPP = 72, 40, 2
sc = rr.cornell_box(W / H)
osc = O.OracleScene(sc)
rgb, n, st = osc.render(sc.camera, rr.make_cfg(W, H, SPP, seed=3, tile_rank=rank, tile_world=world), threads=2)
mode = os.environ["RR_MODE"]
t_rgb, t_n = torch.from_numpy(rgb.copy()), torch.from_numpy(n.astype(np.int32))
if mode == "packed":
    g = rd.FilmGather(W, H, "cpu")
    assert g.count[rank] == int((n > 0).sum())
    g.gather(t_rgb, t_n)
else:
    rd.reduce_film(t_rgb, t_n)
rays = torch.tensor([float(st.rays)], dtype=torch.float64)
dist.all_reduce(rays)
if rank == 0:
    full, nfull, sfull = osc.render(sc.camera, rr.make_cfg(W, H, SPP, seed=3), threads=2)
    assert np.array_equal(t_rgb.numpy(), full), "gathered film differs from the 1-rank film"
    assert np.array_equal(t_n.numpy().astype(np.uint32), nfull)
    assert int(rays.item()) == sfull.rays
    print("OK", mode, flush=True)
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(mode, world=2):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RR_ROOT=ROOT, RR_MODE=mode)
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    assert f"OK {mode}" in outs[0]


def test_packed_gather_world2_gloo():
    _run("packed")


def test_reduce_world2_gloo():
    _run("reduce")


def test_packed_gather_world3_gloo():
    _run("packed", world=3)


def test_owned_pixels_partition():
    W, H = 70, 50
    for world in (1, 2, 3, 8):
        parts = [rd.owned_pixels(W, H, r, world) for r in range(world)]
        allp = np.concatenate(parts)
        assert len(allp) == W * H and len(np.unique(allp)) == W * H
        # same ownership rule as the oracle's / library's tile loop
        sc = rr.cornell_box(W / H)
        osc = O.OracleScene(sc)
        _, n, _ = osc.render(sc.camera, rr.make_cfg(W, H, 1, max_depth=0, tile_rank=world - 1, tile_world=world), threads=2)
        assert np.array_equal(np.sort(parts[world - 1]), np.flatnonzero(n.reshape(-1)))
    win = (8, 4, 40, 30)
    p = rd.owned_pixels(W, H, 1, 2, window=win)
    ys, xs = p // W, p % W
    assert xs.min() >= 8 and xs.max() < 40 and ys.min() >= 4 and ys.max() < 30


def test_tile_owner_rule_matches_the_exported_function():
    """include/rt_abi.h: rt_tile_owner -- the library's export, dist.py's mirror and the documented properties:
    every tile row and every tile column deals its tiles to all ranks in turn (a row of 120 tiles, as in the
    1920-wide headline image, must not hand whole columns to one rank)."""
    from rustraytracer_amd import _ffi as F
    L = F.lib()
    for world in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 64):
        s = rd.tile_stride(world)
        assert world == 1 or math.gcd(s, world) == 1
        for ty in range(0, 70, 7):
            for tx in range(0, 130, 11):
                assert L.rt_tile_owner(tx, ty, world) == rd.tile_owner(tx, ty, world) < max(world, 1)
        row = [rd.tile_owner(tx, 5, world) for tx in range(world)]
        col = [rd.tile_owner(9, ty, world) for ty in range(world)]
        assert sorted(row) == sorted(col) == list(range(world))
    # 120 tiles per row, 8 ranks: a rank's tiles are not whole columns
    cols = {tx for ty in range(68) for tx in range(120) if rd.tile_owner(tx, ty, 8) == 3}
    assert len(cols) == 120


FAILING_WORKER = r'''
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["RR_ROOT"])
from rustraytracer_amd import dist as rd
rank = int(os.environ["RANK"])
def job():
    rd.init_process_group("gloo")
    g = rd.FilmGather(64, 32, "cpu")
    rgb, n = torch.zeros((32, 64, 3), dtype=torch.float64), torch.zeros((32, 64), dtype=torch.int32)
    if rank == 1:
        raise RuntimeError("simulated HIP error on rank 1")  # (what rt_render's RtError looks like to the caller)
    g.gather(rgb, n)  # rank 0 waits for a payload that never comes (a non-root rank only sends)
    dist.barrier()    # ... and every rank for rank 1, as bench.py's timing barrier does
    print("UNREACHABLE", flush=True)
rd.run_guarded(job)
'''


def test_a_failing_rank_takes_the_job_down_within_the_timeout():
    """one rank's failure must end every process of the job with a non-zero exit code in bounded time
    (no rank may sit in the film gather forever).  Three gloo ranks, rank 1 raises before the gather."""
    import time
    port = _free_port()
    t0 = time.time()
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RR_ROOT=ROOT,
                   RT_DIST_TIMEOUT_S="20")
        procs.append(subprocess.Popen([sys.executable, "-c", FAILING_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    took = time.time() - t0
    for p, o in zip(procs, outs):
        assert p.returncode != 0, o[-2000:]
        assert "UNREACHABLE" not in o
    assert "simulated HIP error on rank 1" in outs[1]
    assert took < 200, took


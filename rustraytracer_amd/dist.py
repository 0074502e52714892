"""Multi-GPU image-plane sharding: one process per GPU, tiles interleaved over ranks, film gather.

The reference already farms 16x16 tiles to worker threads (src/render.rs:49-71); here tile
(tx, ty) belongs to rank rt_tile_owner(tx, ty, world) = (tx + ty * stride) % world (include/rt_abi.h;
rt_render_cfg.tile_rank / tile_world).  Pixels are
independent (RNG keyed by seed, pixel, sample), so the only exchange step of the whole path is the
final framebuffer gather to rank 0.  `gather_film` packs each rank's own pixels and sends them
straight to the root (torch.distributed.gather = point-to-point sends on RCCL, one xGMI link per
peer) -- 1/world of the bytes a full-frame reduce would move; `reduce_film` is the simple variant.
Backends: "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

TILE_SIZE = 16  # src/consts.rs:10


def init_process_group(backend, device_id=None, timeout_s=None):
    """torch.distributed.init_process_group with a BOUNDED collective timeout (RT_DIST_TIMEOUT_S, default 600 s): when a
    rank dies -- a HIP or RCCL error, an exception anywhere -- the others' next collective fails within that time instead of
    waiting for it forever, and every process of the job exits non-zero (bench.py: run_guarded).  Under torchrun the agent
    also ends the other workers as soon as one exits with an error."""
    import datetime
    import os
    t = float(timeout_s if timeout_s is not None else os.environ.get("RT_DIST_TIMEOUT_S", "600"))
    kw = {"timeout": datetime.timedelta(seconds=t)}
    if device_id is not None:
        kw["device_id"] = device_id
    dist.init_process_group(backend=backend, **kw)


def run_guarded(fn):
    """Run fn(); any exception -> its message with the rank on stderr, exit code 1 at once (os._exit: no atexit handler may
    sit in a collective the failed rank will never join)."""
    import os
    import sys
    import traceback
    try:
        return fn()
    except SystemExit:
        raise
    except BaseException:  # noqa: BLE001 -- the job must come down whatever it was
        sys.stderr.write("[rank %s] failed:\n%s" % (os.environ.get("RANK", "0"), traceback.format_exc()))
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(1)


def tile_stride(world):
    """include/rt_abi.h: rt_tile_stride -- the integer nearest to world / golden ratio that is coprime to world."""
    if world <= 1:
        return 0
    s = max((world * 618034 + 500000) // 1000000, 1)
    while math.gcd(s, world) != 1:
        s += 1
    return s % world


def tile_owner(tx, ty, world):
    """include/rt_abi.h: rt_tile_owner -- rank-1 lattice: every tile row and every tile column visits all ranks in turn."""
    return (tx + ty * tile_stride(world)) % world if world > 1 else 0


def owned_pixels(width, height, rank, world, tile_size=TILE_SIZE, window=None):
    """Linear pixel indices (py*W+px) this rank renders, in the library's tile order."""
    x0, y0, x1, y1 = window if window else (0, 0, width, height)
    tw = (width + tile_size - 1) // tile_size
    th = (height + tile_size - 1) // tile_size
    out = []
    for k in range(tw * th):
        tx, ty = k % tw, k // tw
        if tile_owner(tx, ty, world) != rank:
            continue
        ys = np.arange(ty * tile_size, min((ty + 1) * tile_size, height))
        xs = np.arange(tx * tile_size, min((tx + 1) * tile_size, width))
        ys = ys[(ys >= y0) & (ys < y1)]
        xs = xs[(xs >= x0) & (xs < x1)]
        if len(ys) and len(xs):
            out.append((ys[:, None] * width + xs[None, :]).reshape(-1))
    return np.concatenate(out) if out else np.zeros(0, dtype=np.int64)


def reduce_film(rgb_sum, n, dst=0):
    """Full-frame SUM-reduce to `dst` (films are zero outside a rank's own tiles, so x + 0 = x)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(rgb_sum, dst=dst, op=dist.ReduceOp.SUM)
        dist.reduce(n, dst=dst, op=dist.ReduceOp.SUM)
    return rgb_sum, n


class FilmGather:
    """Packed gather of the per-rank films to the root.  Index tensors are built once per image shape."""

    def __init__(self, width, height, device, tile_size=TILE_SIZE, window=None):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.width, self.height = width, height
        idx = [owned_pixels(width, height, r, self.world, tile_size, window) for r in range(self.world)]
        self.count = [len(i) for i in idx]
        self.max_count = max(self.count) if self.count else 0
        self.own = torch.as_tensor(idx[self.rank], dtype=torch.int64, device=device)
        self.all_idx = [torch.as_tensor(i, dtype=torch.int64, device=device) for i in idx] if self.rank == 0 else None
        self.device = device

    def gather(self, rgb_sum, n, dst=0):
        """rgb_sum [H,W,3] f64, n [H,W] i32 (this rank's film).  On `dst` they become the full film."""
        if self.world == 1:
            return rgb_sum, n
        assert dst == 0
        flat_rgb = rgb_sum.view(-1, 3)
        flat_n = n.view(-1)
        # one payload per rank: [max_count, 4] doubles = r, g, b, n (n is exact in f64)
        pay = torch.zeros((self.max_count, 4), dtype=torch.float64, device=self.device)
        k = self.count[self.rank]
        pay[:k, :3] = flat_rgb[self.own]
        pay[:k, 3] = flat_n[self.own].to(torch.float64)
        bufs = [torch.empty_like(pay) for _ in range(self.world)] if self.rank == dst else None
        dist.gather(pay, bufs, dst=dst)
        if self.rank == dst:
            for r in range(self.world):
                if r == dst:
                    continue
                kr = self.count[r]
                flat_rgb[self.all_idx[r]] = bufs[r][:kr, :3]
                flat_n[self.all_idx[r]] = bufs[r][:kr, 3].to(flat_n.dtype)
        return rgb_sum, n

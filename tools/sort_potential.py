#!/usr/bin/env python3
"""What would ordering the rays of the deep bounces buy k_trace?  Secondary rays that start on the surfaces of C4's scene
with uniformly random directions go through the render's traversal kernel (rt_intersect_batch_ex, RT_INTERSECT_WAVEFRONT)
(a) in random order, (b) ordered by the Morton code of their origin (a perfect spatial sort), (c) by origin cell AND
direction octant.  Run under rocprofv3 --kernel-trace --stats: the k_trace launches appear in this order.
usage (GPU box): rocprofv3 --kernel-trace --stats -d gpurun_out/sortpot -- python3 tools/sort_potential.py [n_rays]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustraytracer_amd as rr
from rustraytracer_amd import _ffi as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
rng = np.random.default_rng(3)
sc = rr.two_dragons(1920 / 1080, mesh_faces=871414, variant=0)
ctx = rr.Context(0)
gs = ctx.upload(sc)


def unit(v):
    return v / np.linalg.norm(v, axis=1, keepdims=True)


# camera-like rays from a shell around the scene towards it: their hit points are the origins of the secondary rays
o0 = unit(rng.normal(size=(n, 3))) * 14.0 + np.array([2.5, 3.0, 0.0])
target = np.array([2.5, 1.0, 0.0]) + rng.uniform(-3.5, 3.5, size=(n, 3)) * np.array([1.0, 0.5, 0.6])
d0 = unit(target - o0)
t, prim = ctx.intersect_batch(gs, o0, d0, F.RT_SMALL, flags=F.RT_INTERSECT_WAVEFRONT)
hit = prim >= 0
o1 = (o0 + d0 * t[:, None])[hit]
d1 = unit(rng.normal(size=o1.shape))
print("secondary rays:", o1.shape[0], "of", n, flush=True)


def morton(p, bits=10):
    lo, hi = p.min(axis=0), p.max(axis=0)
    q = np.clip(((p - lo) / np.maximum(hi - lo, 1e-30) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(p.shape[0], dtype=np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return code


def run(label, order):
    t1, p1 = ctx.intersect_batch(gs, o1[order], d1[order], F.RT_SMALL, flags=F.RT_INTERSECT_WAVEFRONT)
    cost = ctx.last_intersect_cost
    print("%-34s hits %d nodes/ray %.2f" % (label, int((p1 >= 0).sum()), float((cost & 0xff).mean())), flush=True)
    return p1


m = morton(o1)
base = run("random order", rng.permutation(o1.shape[0]))
run("random order (again)", rng.permutation(o1.shape[0]))
run("by origin (Morton, 30 bits)", np.argsort(m, kind="stable"))
octant = (d1[:, 0] < 0).astype(np.int64) | ((d1[:, 1] < 0).astype(np.int64) << 1) | ((d1[:, 2] < 0).astype(np.int64) << 2)
run("by origin cell (9 bits), octant", np.argsort(((m >> 21) << 3) | octant, kind="stable"))
run("by origin cell (9 bits) only", np.argsort(m >> 21, kind="stable"))
gs.close()

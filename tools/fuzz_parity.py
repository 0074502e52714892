"""Randomised parity sweep (run on the GPU box): presets x random image size / spp / depth / seed / pool size / tile
split / BVH builder, GPU film and counters against the oracle.  usage: python tools/fuzz_parity.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

import numpy as np

import rustraytracer_amd as rr
from rustraytracer_amd import _ffi as F
from tests import oracle_ffi as O
from tests.test_gpu_arms import emitter_scene

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = rr.Context(0)
makers = [
    ("cornell_box", lambda f, v: rr.cornell_box()),
    ("cornell_box_spheres", lambda f, v: rr.cornell_box_spheres()),
    ("cornell_box_statue", lambda f, v: rr.cornell_box_statue(mesh_faces=f, variant=v % 4)),
    ("plastic_dragon", lambda f, v: rr.plastic_dragon(mesh_faces=f, variant=v % 3)),
    ("two_dragons", lambda f, v: rr.two_dragons(mesh_faces=f, variant=v % 2)),
    ("sphere_roughness", lambda f, v: rr.sphere_roughness()),
    ("material_hdr", lambda f, v: rr.material_hdr(v % 4, mesh_faces=f)),
    # round 2: triangle / sphere emitters, a uv-mapped checkered mesh, a mesh without normals (tests/test_gpu_arms.py)
    ("emitters", lambda f, v: emitter_scene(bool(v & 1), sphere_light=bool(v & 2) or not (v & 4), tri_lights=bool(v & 4) or not (v & 2),
                                            two_sided=bool(v & 1))),
]
bad = 0
for case in range(n_cases):
    name, make = makers[int(rng.integers(len(makers)))]
    faces, variant = int(rng.integers(2000, 60000)), int(rng.integers(8))
    W, H = int(rng.integers(24, 200)), int(rng.integers(24, 160))
    spp = int(rng.choice([1, 2, 3, 5, 8, 16, 24, 40]))
    depth = int(rng.choice([1, 2, 5, 25]))
    pif = int(rng.choice([0, 64, 1000, 50000]))
    world = int(rng.choice([1, 1, 2, 3]))
    dev_build = bool(rng.integers(2))
    seed = int(rng.integers(1 << 30))
    sc = make(faces, variant)
    gs = ctx.upload(sc, device_build=dev_build)
    # round 3: a third of the cases through the lens arm of Camera::get_ray (geometry.rs:177-190): the preset's camera
    # with lens_radius > 0 (the frame u, v and the focus plane stay the preset's)
    cam = F.rt_camera()
    src = sc.camera  # a pointer for the host presets, a struct for the tests' scene builder
    C.memmove(C.byref(cam), src if isinstance(src, C._Pointer) else C.byref(src), C.sizeof(cam))
    lens = float(rng.choice([0.0, 0.0, 0.08, 2.5]))
    cam.lens_radius = lens
    acc = nacc = None
    tot = np.zeros(5, dtype=np.int64)
    for rank in range(world):
        cfg = rr.make_cfg(W, H, spp, max_depth=depth, seed=seed, paths_in_flight=pif, tile_rank=rank, tile_world=world)
        r, n, s = ctx.render(gs, cam, cfg)
        acc = r if acc is None else acc + r
        nacc = n if nacc is None else nacc + n
        tot += np.array([s.rays, s.rays_extension, s.rays_shadow, s.rays_probe, s.vertices_shaded])
    ro, no, so = O.OracleScene(sc).render(cam, rr.make_cfg(W, H, spp, max_depth=depth, seed=seed), O.ORDERED, 16)
    same = np.array_equal(acc, ro, equal_nan=True) and np.array_equal(nacc, no)
    cnt = tuple(tot) == (so.rays, so.rays_extension, so.rays_shadow, so.rays_probe, so.vertices_shaded)
    bad += not (same and cnt)
    print(f"{case:3d} {name:20s} faces {faces:6d} v{variant} {W}x{H}@{spp} depth {depth} pool {pif} ranks {world} lens {lens} "
          f"{'lbvh' if dev_build else 'sah '} rays {int(tot[0]):9d} film {'==' if same else 'DIFF'} counters {'==' if cnt else 'DIFF'}",
          flush=True)
    gs.close()
print("cases", n_cases, "failures", bad)
sys.exit(1 if bad else 0)

"""Regression fixtures for the ORACLE itself (not reference output: the reference cannot run here).
Small oracle films at fixed seeds, stored bit for bit, so that an accidental change of the restatement (or of
rt_detmath.h, the RNG, a preset) shows up in the CPU-only test run.  Regenerate only on purpose:
    python tests/golden/make_oracle_films.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rustraytracer_amd as rr
from tests import oracle_ffi as O

CASES = {
    "cornell_box": (lambda: rr.cornell_box(), 24, 24, 4, 1),
    "cornell_statue_plastic": (lambda: rr.cornell_box_statue(mesh_faces=2000, variant=3), 24, 24, 4, 2),
    "dragon_glass": (lambda: rr.plastic_dragon(mesh_faces=2000, variant=2), 24, 24, 4, 3),
    "two_dragons": (lambda: rr.two_dragons(16 / 9, mesh_faces=1500, variant=0), 32, 18, 4, 4),
    "material_hdr_rough_glass": (lambda: rr.material_hdr(3, mesh_faces=1000), 24, 24, 4, 5),
    "sphere_roughness": (lambda: rr.sphere_roughness(), 32, 18, 4, 6),
}


def render(name):
    make, w, h, spp, seed = CASES[name]
    sc = make()
    rgb, n, st = O.OracleScene(sc).render(sc.camera, rr.make_cfg(w, h, spp, seed=seed), O.ORDERED)
    return rgb, np.array([st.rays_extension, st.rays_shadow, st.rays_probe, st.vertices_shaded], dtype=np.uint64)


if __name__ == "__main__":
    out = {}
    for name in CASES:
        rgb, counts = render(name)
        out[name + "_rgb"] = rgb
        out[name + "_counts"] = counts
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_films.npz"), **out)
    print("wrote", len(CASES), "films")

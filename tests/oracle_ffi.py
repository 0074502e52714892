"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os

import numpy as np

from rustraytracer_amd import _ffi as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# RT_ORACLE_LIB: another build of the same checker (tests/test_host_sanitizers.py loads oracle/liboracle_asan.so)
LIB_PATH = os.environ.get("RT_ORACLE_LIB") or os.path.join(ROOT, "oracle", "liboracle.so")

EXHAUSTIVE, ORDERED, BRUTE = 0, 1, 2


class oracle_hit_record(C.Structure):
    _fields_ = [("hit", C.c_int32), ("front", C.c_int32), ("t", C.c_double), ("uv", C.c_double * 2),
                ("p", C.c_double * 3), ("n", C.c_double * 3), ("sh_n", C.c_double * 3),
                ("sh_dpdu", C.c_double * 3), ("sh_dpdv", C.c_double * 3)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} missing: run `make -C oracle` (or __graft_entry__.build())")
    L = C.CDLL(LIB_PATH)
    vp, d, dp = C.c_void_p, C.c_double, C.POINTER(C.c_double)
    L.oracle_scene_create.argtypes = [C.POINTER(F.rt_scene_desc), C.POINTER(vp)]
    L.oracle_scene_destroy.argtypes = [vp]
    L.oracle_render.argtypes = [vp, C.POINTER(F.rt_camera), C.POINTER(F.rt_render_cfg), C.c_int, C.c_int, vp, vp,
                                C.POINTER(F.rt_stats)]
    L.oracle_intersect_batch.argtypes = [vp, C.POINTER(F.rt_ray), C.c_uint64, C.c_int, C.POINTER(F.rt_hit)]
    L.oracle_prim_intersect.argtypes = [vp, C.c_int32, C.POINTER(F.rt_ray), C.POINTER(oracle_hit_record)]
    L.oracle_sample.argtypes = [vp, C.POINTER(F.rt_camera), C.POINTER(F.rt_render_cfg), C.c_uint32, C.c_uint32,
                                C.c_uint32, C.c_int, dp, C.POINTER(F.rt_stats)]
    L.oracle_sample_rays.restype = C.c_int64
    L.oracle_sample_rays.argtypes = [vp, C.POINTER(F.rt_camera), C.POINTER(F.rt_render_cfg), C.c_uint32, C.c_uint32,
                                     C.c_uint32, C.c_int, C.POINTER(F.rt_ray), C.POINTER(F.rt_hit), C.c_uint64]
    L.oracle_fr_dielectric.restype = d
    L.oracle_fr_dielectric.argtypes = [d, d, d]
    L.oracle_fr_conductor.argtypes = [d, dp, dp, dp]
    L.oracle_power_heuristic.restype = d
    L.oracle_power_heuristic.argtypes = [C.c_int, d, C.c_int, d]
    for name in ("oracle_tr_d", "oracle_tr_lambda"):
        getattr(L, name).restype = d
        getattr(L, name).argtypes = [d, d, dp]
    for name in ("oracle_tr_g", "oracle_tr_pdf"):
        getattr(L, name).restype = d
        getattr(L, name).argtypes = [d, d, dp, dp]
    L.oracle_tr_sample_wh.argtypes = [d, d, dp, d, d, dp]
    L.oracle_tr_roughness_to_alpha.restype = d
    L.oracle_tr_roughness_to_alpha.argtypes = [d]
    L.oracle_concentric_sample_disk.argtypes = [d, d, dp]
    L.oracle_rand_cosine_dir.argtypes = [d, d, dp]
    L.oracle_rng_draws.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, dp]
    L.oracle_box_intersects.argtypes = [dp, dp, C.POINTER(F.rt_ray)]
    L.oracle_refract.argtypes = [dp, dp, d, dp]
    L.oracle_lambert_f_pdf.argtypes = [dp, dp, dp, dp, dp]
    L.oracle_microfacet_f_pdf.argtypes = [d, d, dp, dp, dp, dp, dp, dp]
    L.oracle_micro_trans.argtypes = [d, d, d, dp, dp, dp, d, d, dp, dp, dp, dp, dp]
    L.oracle_env.argtypes = [vp, C.c_int, dp, dp]
    L.oracle_prim_area.restype = d
    L.oracle_prim_area.argtypes = [vp, C.c_int32]
    L.oracle_prim_pdf.restype = d
    L.oracle_prim_pdf.argtypes = [vp, C.c_int32, dp, dp]
    L.oracle_texture_value.argtypes = [vp, C.c_uint32, d, d, dp]
    L.oracle_detmath.argtypes = [C.c_int, dp, dp, C.c_uint64, dp]
    L.oracle_resolve_rgb8.argtypes = [vp, vp, C.c_uint64, vp]
    _lib = L
    return L


def vec(*xs):
    return (C.c_double * len(xs))(*xs)


def make_ray(o, d, tmin=F.RT_SMALL, tmax=F.RT_INFINITY):
    r = F.rt_ray()
    r.origin[:] = o
    r.dir[:] = d
    r.tmin, r.tmax = tmin, tmax
    return r


class OracleScene:
    def __init__(self, scene):
        """scene: rustraytracer_amd.Scene (host-side flattened preset)."""
        self.scene = scene  # keeps the arrays alive
        h = C.c_void_p()
        rc = lib().oracle_scene_create(scene.desc, C.byref(h))
        assert rc == 0
        self._h = h

    def render(self, camera, cfg, mode=ORDERED, threads=8):
        rgb = np.zeros((cfg.height, cfg.width, 3), dtype=np.float64)
        n = np.zeros((cfg.height, cfg.width), dtype=np.uint32)
        st = F.rt_stats()
        rc = lib().oracle_render(self._h, camera, C.byref(cfg), mode, threads, rgb.ctypes.data_as(C.c_void_p),
                                 n.ctypes.data_as(C.c_void_p), C.byref(st))
        assert rc == 0, rc
        return rgb, n, st

    def intersect_batch(self, origins, dirs, tmin, tmax=F.RT_INFINITY, mode=ORDERED):
        origins = np.ascontiguousarray(origins, dtype=np.float64)
        dirs = np.ascontiguousarray(dirs, dtype=np.float64)
        n = origins.shape[0]
        rays = (F.rt_ray * n)()
        ra = np.frombuffer(rays, dtype=np.float64).reshape(n, 8)
        ra[:, 0:3] = origins
        ra[:, 3:6] = dirs
        ra[:, 6] = tmin
        ra[:, 7] = tmax
        hits = (F.rt_hit * n)()
        rc = lib().oracle_intersect_batch(self._h, rays, n, mode, hits)
        assert rc == 0
        ha = np.frombuffer(hits, dtype=np.dtype([("t", "<f8"), ("prim", "<i4"), ("r", "<u4")]))
        return ha["t"].copy(), ha["prim"].copy()

    def prim_intersect(self, prim, o, d, tmin=F.RT_SMALL, tmax=F.RT_INFINITY):
        rec = oracle_hit_record()
        r = make_ray(o, d, tmin, tmax)
        rc = lib().oracle_prim_intersect(self._h, prim, C.byref(r), C.byref(rec))
        assert rc == 0
        return rec

    def sample(self, camera, cfg, px, py, s, mode=ORDERED):
        rgb = (C.c_double * 3)()
        st = F.rt_stats()
        rc = lib().oracle_sample(self._h, camera, C.byref(cfg), px, py, s, mode, rgb, C.byref(st))
        assert rc == 0
        return np.array(rgb[:]), st

    def sample_rays(self, camera, cfg, px, py, s, mode=ORDERED, capacity=256):
        """Every root closest-hit query of one camera sample: (origins, dirs, tmin, t, prim)."""
        rays = (F.rt_ray * capacity)()
        hits = (F.rt_hit * capacity)()
        n = lib().oracle_sample_rays(self._h, camera, C.byref(cfg), px, py, s, mode, rays, hits, capacity)
        assert 0 <= n <= capacity, n
        ra = np.frombuffer(rays, dtype=np.float64).reshape(capacity, 8)[:n]
        ha = np.frombuffer(hits, dtype=np.dtype([("t", "<f8"), ("prim", "<i4"), ("r", "<u4")]))[:n]
        return ra[:, 0:3].copy(), ra[:, 3:6].copy(), ra[:, 6].copy(), ha["t"].copy(), ha["prim"].copy()

    def env(self, what, *xs):
        """Light::Infinite tables: 0 sample(u0,u1)->(uv0,uv1,pdf); 1 pdf(p0,p1); 2 le(dir); 3 pdf_li(dir); 4 sizes"""
        out = (C.c_double * 3)()
        rc = lib().oracle_env(self._h, what, vec(*xs) if xs else vec(0.0), out)
        assert rc == 0, rc
        return np.array(out[:])

    def close(self):
        if self._h:
            lib().oracle_scene_destroy(self._h)
            self._h = None

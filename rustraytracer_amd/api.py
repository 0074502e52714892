"""Thin host-side wrappers over the C ABI (librt_amd.so).

Mirrors the reference's call shape for the hot path:
  scenes.<preset>()                     -> Scene  (src/scenes.rs presets, flattened `Objects`)
  render.tile_multithread(path, camera, sampler, int_type)   (src/render.rs:13)
                                        -> Context.render(scene, width, height, spp, max_depth)
All compute goes through the HIP library; nothing here computes pixels on the CPU.
"""
import ctypes as C

import numpy as np

from . import _ffi as F

MAX_DEPTH = 25  # src/consts.rs:7
TILE_SIZE = 16  # src/consts.rs:10


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt error {code}: {msg}")
        self.code = code


def _check(code, host=False):
    if code != F.RT_OK:
        L = F.lib()
        msg = (L.rrh_last_error() if host else L.rt_last_error()) or b""
        raise RtError(code, msg.decode(errors="replace"))


class Scene:
    """A flattened scene (`Objects`, src/geometry.rs:13-21) plus its preset camera."""

    def __init__(self, preset, aspect_ratio=1.0, mesh_faces=0, mesh_path=None, variant=0):
        L = F.lib()
        h = C.c_void_p()
        _check(L.rrh_scene_build(preset.encode(), float(aspect_ratio), int(mesh_faces),
                                 mesh_path.encode() if mesh_path else None, int(variant), C.byref(h)), host=True)
        self._h = h
        self.preset = preset

    @property
    def desc(self):
        return F.lib().rrh_scene_desc(self._h)

    @property
    def camera(self):
        return F.lib().rrh_scene_camera(self._h)

    @property
    def name(self):
        return F.lib().rrh_scene_name(self._h).decode()

    def close(self):
        if self._h:
            F.lib().rrh_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_cfg(width, height, spp, max_depth=MAX_DEPTH, seed=0, window=None, tile_rank=0, tile_world=1,
             paths_in_flight=0, count_traversal=False, sample_first=0, sample_count=0, accumulate=False, cancel=None,
             precision=0):
    cfg = F.rt_render_cfg()
    cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed = width, height, spp, max_depth, seed
    if window:
        cfg.x0, cfg.y0, cfg.x1, cfg.y1 = window
    cfg.tile_size = TILE_SIZE
    cfg.tile_rank, cfg.tile_world = tile_rank, tile_world
    cfg.precision = precision  # F.RT_PRECISION_F64 (parity mode) / F.RT_PRECISION_F32 (fast mode)
    if cancel is not None:  # a ctypes c_int32 the caller may set from another thread (render.rs:93 stop_render)
        cfg.cancel = C.pointer(cancel)
    cfg.paths_in_flight = paths_in_flight
    cfg.flags = (F.RT_RENDER_COUNT_TRAVERSAL if count_traversal else 0) | (F.RT_RENDER_ACCUMULATE if accumulate else 0)
    cfg.sample_first, cfg.sample_count = sample_first, sample_count  # progressive pass (0, 0 = every sample)
    return cfg


class GpuScene:
    def __init__(self, ctx, scene, device_build=False):
        self.ctx = ctx
        h = C.c_void_p()
        flags = F.RT_COMMIT_DEVICE_LBVH if device_build else F.RT_COMMIT_HOST_SAH
        _check(F.lib().rrh_scene_upload_ex(ctx._h, scene.desc, flags, C.byref(h)))
        self._h = h

    def info(self):
        inf = F.rt_scene_info()
        _check(F.lib().rt_scene_get_info(self._h, C.byref(inf)))
        return {k: getattr(inf, k) for k, _ in inf._fields_}

    def close(self):
        if self._h:
            F.lib().rt_scene_destroy(self._h)
            self._h = None


class Context:
    """rt_context: one device (an int) or several GPUs of the node (a list of device ids; films live on the first)."""

    def __init__(self, device=0):
        L = F.lib()
        h = C.c_void_p()
        ids = list(device) if isinstance(device, (list, tuple)) else [device]
        dev = (C.c_int * len(ids))(*ids)
        _check(L.rt_context_create(dev, len(ids), C.byref(h)))
        self._h = h
        self.device = ids[0]
        self.devices = ids

    def upload(self, scene, device_build=False):
        """rt_scene_create + set_* + rt_scene_commit_ex; device_build=True builds the BVH on the GPU (row f3)."""
        return GpuScene(self, scene, device_build)

    def render(self, gscene, camera, cfg, film=None):
        """rt_render: returns (rgb_sum[H,W,3] f64, n[H,W] u32, rt_stats).  `film` = (rgb_sum, n) of the earlier
        passes when cfg carries RT_RENDER_ACCUMULATE (progressive rendering); it is updated in place."""
        if film is not None:
            rgb, n = film
            assert rgb.dtype == np.float64 and n.dtype == np.uint32 and rgb.flags.c_contiguous and n.flags.c_contiguous
        else:
            rgb = np.zeros((cfg.height, cfg.width, 3), dtype=np.float64)
            n = np.zeros((cfg.height, cfg.width), dtype=np.uint32)
        st = F.rt_stats()
        _check(F.lib().rt_render(self._h, gscene._h, camera, C.byref(cfg), rgb.ctypes.data_as(C.c_void_p),
                                 n.ctypes.data_as(C.c_void_p), C.byref(st)))
        return rgb, n, st

    def render_device(self, gscene, camera, cfg, d_rgb_ptr, d_n_ptr, stream=None):
        """rt_render_device: film stays in caller-owned device buffers (torch tensors' data_ptr())."""
        st = F.rt_stats()
        _check(F.lib().rt_render_device(self._h, gscene._h, camera, C.byref(cfg), C.c_void_p(d_rgb_ptr),
                                        C.c_void_p(d_n_ptr), C.c_void_p(stream or 0), C.byref(st)))
        return st

    def intersect_batch(self, gscene, origins, dirs, tmin, tmax=F.RT_INFINITY, flags=0):
        """rt_intersect_batch_ex; flags: F.RT_INTERSECT_F32 | F.RT_INTERSECT_WAVEFRONT"""
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
        _check(F.lib().rt_intersect_batch_ex(self._h, gscene._h, rays, n, hits, flags))
        ha = np.frombuffer(hits, dtype=np.dtype([("t", "<f8"), ("prim", "<i4"), ("r", "<u4")]))
        self.last_intersect_cost = ha["r"].copy()  # nodes | tris << 8 | others << 16 per ray (saturating bytes)
        return ha["t"].copy(), ha["prim"].copy()

    def resolve_rgb8(self, rgb_sum, n):
        h, w = n.shape
        out = np.zeros((h, w, 3), dtype=np.uint8)
        rgb_sum = np.ascontiguousarray(rgb_sum, dtype=np.float64)
        n = np.ascontiguousarray(n, dtype=np.uint32)
        _check(F.lib().rt_resolve_rgb8(self._h, rgb_sum.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p), w, h,
                                       out.ctypes.data_as(C.c_void_p)))
        return out

    def gpu_tile(self, gscene, camera, width, height, samples_per_pixel, max_depth=MAX_DEPTH, seed=0):
        """GPU sibling of render::tile_multithread (src/render.rs:13) via rrh_gpu_tile."""
        rgb = np.zeros((height, width, 3), dtype=np.float64)
        n = np.zeros((height, width), dtype=np.uint32)
        st = F.rt_stats()
        _check(F.lib().rrh_gpu_tile(self._h, gscene._h, camera, width, height, samples_per_pixel, max_depth, seed,
                                    rgb.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p), C.byref(st)))
        return rgb, n, st

    def close(self):
        if self._h:
            F.lib().rt_context_destroy(self._h)
            self._h = None


# scenes.rs preset names
def cornell_box(aspect_ratio=1.0):
    return Scene("cornell_box", aspect_ratio)


def cornell_box_spheres(aspect_ratio=1.0):
    return Scene("cornell_box_spheres", aspect_ratio)


def cornell_box_statue(aspect_ratio=1.0, mesh_faces=0, mesh_path=None, variant=0):
    return Scene("cornell_box_statue", aspect_ratio, mesh_faces, mesh_path, variant)


def plastic_dragon(aspect_ratio=1.0, mesh_faces=0, mesh_path=None, variant=0):
    return Scene("plastic_dragon", aspect_ratio, mesh_faces, mesh_path, variant)


def sphere_roughness(aspect_ratio=1.0):
    return Scene("sphere_roughness", aspect_ratio)


def two_dragons(aspect_ratio=1.0, mesh_faces=0, mesh_path=None, variant=0):
    return Scene("two_dragons", aspect_ratio, mesh_faces, mesh_path, variant)


def material_hdr(mat_num=0, aspect_ratio=1.0, mesh_faces=0, data_dir=None):
    """scenes.rs:627-741 (row f4): environment-lit material test; mat_num 0 plastic / 1 metal / 2 mirror /
    3 rough glass.  data_dir = the reference's data/material directory; missing files -> procedural stand-ins."""
    return Scene("material_hdr", aspect_ratio, mesh_faces, data_dir, mat_num)


def teapot_hdr(aspect_ratio=1.0, mesh_faces=0, data_dir=None):
    """scenes.rs:744-808 (row f4): smooth-plastic teapot (two meshes) on a checkered floor under the environment map.
    data_dir = a directory with models/Mesh000.obj, models/Mesh001.obj, textures/envmap.hdr (the reference's data/teapot
    holds only the environment map); missing files -> procedural stand-ins."""
    return Scene("teapot_hdr", aspect_ratio, mesh_faces, data_dir, 0)


def write_png(path, rgb8):
    """rgb8: (H, W, 3) uint8 (Context.resolve_rgb8) -> PNG file, the last step of util::draw_picture (row f1)."""
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _ = rgb8.shape
    _check(F.lib().rrh_write_png(str(path).encode(), rgb8.ctypes.data_as(C.c_void_p), w, h), host=True)

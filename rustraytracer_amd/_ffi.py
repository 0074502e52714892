"""ctypes view of include/rt_abi.h and include/rt_host.h (librt_amd.so).

The library is the product: if it is missing this module raises -- there is no
Python or CPU fallback for any compute entry point.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB: load another build of the same library (kernel-variant experiments, tools/variants.sh)
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")

RT_OK = 0
RT_ERR_INVALID_ARG = -1
RT_ERR_NO_DEVICE = -2
RT_ERR_HIP = -3
RT_ERR_UNSUPPORTED = -4
RT_ERR_STATE = -5
RT_ERR_OOM = -6
RT_ERR_CANCELLED = -7
RT_PRECISION_F64, RT_PRECISION_F32 = 0, 1
RT_INTERSECT_F32, RT_INTERSECT_WAVEFRONT = 1, 2

RT_PRIM_SPHERE, RT_PRIM_TRIANGLE, RT_PRIM_XY_RECT, RT_PRIM_XZ_RECT, RT_PRIM_YZ_RECT = range(5)
RT_TEX_SOLID, RT_TEX_CHECKERED = 0, 1
RT_MAT_MATTE, RT_MAT_LIGHT, RT_MAT_PLASTIC, RT_MAT_GLASS, RT_MAT_METAL, RT_MAT_MIRROR = range(6)
RT_NO_TEXTURE = 0xFFFFFFFF
RT_RENDER_COUNT_TRAVERSAL = 1
RT_RENDER_ACCUMULATE = 2
RT_COMMIT_HOST_SAH = 0
RT_COMMIT_DEVICE_LBVH = 1
RT_INFINITY = 1e308
RT_SMALL = 0.001


class rt_xform(C.Structure):
    _fields_ = [("fwd", C.c_double * 12), ("inv", C.c_double * 12)]


class rt_primitive(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32), ("flip", C.c_uint32), ("mat_index", C.c_uint32), ("light_index", C.c_int32),
        ("mesh_index", C.c_uint32), ("tri_ind", C.c_uint32), ("xform_index", C.c_int32), ("reserved", C.c_uint32),
        ("v", C.c_double * 5), ("bbox_min", C.c_double * 3), ("bbox_max", C.c_double * 3),
    ]


class rt_mesh(C.Structure):
    _fields_ = [
        ("p", C.POINTER(C.c_double)), ("n", C.POINTER(C.c_double)), ("uv", C.POINTER(C.c_double)),
        ("ind", C.POINTER(C.c_uint32)),
        ("n_p", C.c_uint64), ("n_n", C.c_uint64), ("n_uv", C.c_uint64), ("n_ind", C.c_uint64),
    ]


class rt_texture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("odd", C.c_uint32), ("even", C.c_uint32), ("reserved", C.c_uint32),
                ("color", C.c_double * 3), ("frequency", C.c_double),
                ("rgbe", C.POINTER(C.c_uint8)), ("width", C.c_uint32), ("height", C.c_uint32)]


class rt_material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("remap_roughness", C.c_uint32), ("tex", C.c_uint32 * 5),
                ("reserved", C.c_uint32), ("f", C.c_double * 3)]


class rt_light(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("prim_index", C.c_uint32), ("two_sided", C.c_uint32),
                ("tex_index", C.c_uint32), ("xform_index", C.c_int32), ("reserved", C.c_uint32),
                ("color", C.c_double * 3), ("area", C.c_double), ("world_radius", C.c_double)]


class rt_scene_desc(C.Structure):
    _fields_ = [
        ("meshes", C.POINTER(rt_mesh)), ("n_meshes", C.c_uint64),
        ("prims", C.POINTER(rt_primitive)), ("n_prims", C.c_uint64),
        ("xforms", C.POINTER(rt_xform)), ("n_xforms", C.c_uint64),
        ("materials", C.POINTER(rt_material)), ("n_materials", C.c_uint64),
        ("textures", C.POINTER(rt_texture)), ("n_textures", C.c_uint64),
        ("lights", C.POINTER(rt_light)), ("n_lights", C.c_uint64),
    ]


class rt_camera(C.Structure):
    _fields_ = [
        ("origin", C.c_double * 3), ("upper_left_corner", C.c_double * 3), ("horizontal_offset", C.c_double * 3),
        ("vertical_offset", C.c_double * 3), ("lens_radius", C.c_double), ("t0", C.c_double), ("t1", C.c_double),
        ("u", C.c_double * 3), ("v", C.c_double * 3), ("w", C.c_double * 3),
    ]


class rt_render_cfg(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("max_depth", C.c_uint32),
        ("seed", C.c_uint64),
        ("x0", C.c_uint32), ("y0", C.c_uint32), ("x1", C.c_uint32), ("y1", C.c_uint32),
        ("tile_size", C.c_uint32), ("tile_rank", C.c_uint32), ("tile_world", C.c_uint32),
        ("precision", C.c_uint32), ("paths_in_flight", C.c_uint32), ("flags", C.c_uint32),
        ("sample_first", C.c_uint32), ("sample_count", C.c_uint32),
        ("cancel", C.POINTER(C.c_int32)),
    ]


class rt_stats(C.Structure):
    _fields_ = [
        ("paths", C.c_uint64), ("rays_extension", C.c_uint64), ("rays_shadow", C.c_uint64),
        ("rays_probe", C.c_uint64), ("vertices_shaded", C.c_uint64), ("nodes_fetched", C.c_uint64),
        ("tris_tested", C.c_uint64), ("others_tested", C.c_uint64),
        ("kernel_ms", C.c_double), ("trace_ms", C.c_double), ("trace_launches", C.c_uint64),
        ("tail_rays", C.c_uint64), ("tail_nodes_fetched", C.c_uint64), ("tail_tris_tested", C.c_uint64),
        ("tail_others_tested", C.c_uint64),
        ("gather_ms", C.c_double), ("n_devices", C.c_uint64),
        ("shade_ms", C.c_double), ("shade_launches", C.c_uint64), ("classify_ms", C.c_double), ("light_ms", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}

    @property
    def rays(self):
        return self.rays_extension + self.rays_shadow + self.rays_probe


class rt_ray(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("dir", C.c_double * 3), ("tmin", C.c_double), ("tmax", C.c_double)]


class rt_hit(C.Structure):
    _fields_ = [("t", C.c_double), ("prim", C.c_int32), ("reserved", C.c_uint32)]


class rt_scene_info(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in (
        "n_prims", "n_triangles", "n_others", "n_bvh_nodes", "bvh_depth", "node_bytes", "tri_bytes",
        "other_bytes", "device_bytes_total", "build_flags")] + [("build_ms", C.c_double),
                                                                     ("build_device_ms", C.c_double),
                                                                     ("build_from_cache", C.c_uint64),
                                                                     ("n_classes", C.c_uint64)]


# every symbol include/rt_abi.h and include/rt_host.h declare
ABI_SYMBOLS = [
    "rt_context_create", "rt_context_destroy", "rt_scene_create", "rt_scene_set_meshes",
    "rt_scene_set_primitives", "rt_scene_set_transforms", "rt_scene_set_materials", "rt_scene_set_textures",
    "rt_scene_set_lights", "rt_scene_commit", "rt_scene_commit_ex", "rt_scene_destroy", "rt_scene_get_info", "rt_render",
    "rt_render_device", "rt_intersect_batch", "rt_intersect_batch_ex", "rt_resolve_rgb8", "rt_last_error", "rt_abi_version", "rt_tile_owner",
]
HOST_SYMBOLS = [
    "rrh_scene_build", "rrh_scene_destroy", "rrh_scene_desc", "rrh_scene_camera", "rrh_scene_name",
    "rrh_last_error", "rrh_camera_new", "rrh_scene_upload", "rrh_scene_upload_ex", "rrh_gpu_tile", "rrh_write_png",
]

_lib = None


def _one_hip_runtime():
    """Keep ONE HIP runtime in the process when PyTorch is installed.  The torch wheel bundles its own
    libamdhip64.so (SONAME libamdhip64.so.7) and its libraries ask for it as "libamdhip64.so"; librt_amd.so asks for
    "libamdhip64.so.7".  With torch imported first the loader hands librt_amd.so torch's copy (SONAME match); the
    other way round /opt/rocm's copy is loaded first, torch's request does not match its SONAME, a second runtime
    comes in beside it and torch then finds "No HIP GPUs" -- and device pointers / streams could not be shared
    anyway.  Loading torch's copy by path first makes both orders end in the same single runtime (the loader
    recognises the file when torch asks for it).  Without torch nothing happens and /opt/rocm's runtime is used."""
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            path = os.path.join(libdir, name)
            if os.path.exists(path):
                C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        pass  # a torch without a usable HIP runtime: librt_amd.so loads /opt/rocm's


def lib():
    """Load librt_amd.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no fallback path.")
    _one_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rt_last_error.restype = C.c_char_p
    L.rrh_last_error.restype = C.c_char_p
    L.rrh_scene_name.restype = C.c_char_p
    L.rrh_scene_name.argtypes = [vp]
    L.rt_abi_version.restype = C.c_int
    L.rt_context_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.rt_context_destroy.argtypes = [vp]
    L.rt_scene_create.argtypes = [vp, C.POINTER(vp)]
    L.rt_scene_set_meshes.argtypes = [vp, C.POINTER(rt_mesh), C.c_uint64]
    L.rt_scene_set_primitives.argtypes = [vp, C.POINTER(rt_primitive), C.c_uint64]
    L.rt_scene_set_transforms.argtypes = [vp, C.POINTER(rt_xform), C.c_uint64]
    L.rt_scene_set_materials.argtypes = [vp, C.POINTER(rt_material), C.c_uint64]
    L.rt_scene_set_textures.argtypes = [vp, C.POINTER(rt_texture), C.c_uint64]
    L.rt_scene_set_lights.argtypes = [vp, C.POINTER(rt_light), C.c_uint64]
    L.rt_scene_commit.argtypes = [vp]
    L.rt_scene_commit_ex.argtypes = [vp, C.c_uint32]
    L.rt_scene_destroy.argtypes = [vp]
    L.rt_scene_get_info.argtypes = [vp, C.POINTER(rt_scene_info)]
    L.rt_render.argtypes = [vp, vp, C.POINTER(rt_camera), C.POINTER(rt_render_cfg), vp, vp, C.POINTER(rt_stats)]
    L.rt_render_device.argtypes = [vp, vp, C.POINTER(rt_camera), C.POINTER(rt_render_cfg), vp, vp, vp,
                                   C.POINTER(rt_stats)]
    L.rt_intersect_batch.argtypes = [vp, vp, C.POINTER(rt_ray), C.c_uint64, C.POINTER(rt_hit)]
    L.rt_tile_owner.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    L.rt_tile_owner.restype = C.c_uint32
    L.rt_intersect_batch_ex.argtypes = [vp, vp, C.POINTER(rt_ray), C.c_uint64, C.POINTER(rt_hit), C.c_uint32]
    L.rt_resolve_rgb8.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, vp]
    L.rrh_scene_build.argtypes = [C.c_char_p, C.c_double, C.c_uint64, C.c_char_p, C.c_int, C.POINTER(vp)]
    L.rrh_scene_destroy.argtypes = [vp]
    L.rrh_scene_desc.restype = C.POINTER(rt_scene_desc)
    L.rrh_scene_desc.argtypes = [vp]
    L.rrh_scene_camera.restype = C.POINTER(rt_camera)
    L.rrh_scene_camera.argtypes = [vp]
    L.rrh_camera_new.argtypes = [C.POINTER(C.c_double)] * 3 + [C.c_double] * 6 + [C.POINTER(rt_camera)]
    L.rrh_scene_upload.argtypes = [vp, C.POINTER(rt_scene_desc), C.POINTER(vp)]
    L.rrh_write_png.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
    L.rrh_scene_upload_ex.argtypes = [vp, C.POINTER(rt_scene_desc), C.c_uint32, C.POINTER(vp)]
    L.rrh_gpu_tile.argtypes = [vp, vp, C.POINTER(rt_camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                               C.c_uint64, vp, vp, C.POINTER(rt_stats)]
    _lib = L
    return L

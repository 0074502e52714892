"""rustraytracer_amd -- MI355X-native path-tracing core behind the C ABI of include/rt_abi.h.

Only what the hot path needs: csrc/ (HIP kernels + C ABI + host-side scene mirror) and
the ctypes wrappers used by tests/ and bench.py.  Importing the package does not load the
library; the first call does, and raises if librt_amd.so has not been built.
"""
from . import _ffi  # noqa: F401
from .api import (  # noqa: F401
    MAX_DEPTH, TILE_SIZE, Context, GpuScene, RtError, Scene, cornell_box, cornell_box_spheres,
    cornell_box_statue, make_cfg, material_hdr, plastic_dragon, sphere_roughness, teapot_hdr, two_dragons, write_png,
)

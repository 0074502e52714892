"""Hand-made flattened scenes for tests: a triangle soup (+ optional spheres) as rt_scene_desc."""
import ctypes as C

import numpy as np

from rustraytracer_amd import _ffi as F


class SoupScene:
    """tris: (n, 3, 3) float64 vertices; spheres: (m, 4) centre + radius.  One matte material, no lights."""

    def __init__(self, tris, spheres=()):
        tris = np.ascontiguousarray(np.asarray(tris, dtype=np.float64).reshape(-1, 3, 3))
        spheres = np.asarray(spheres, dtype=np.float64).reshape(-1, 4)
        n, m = tris.shape[0], spheres.shape[0]
        self.p = np.ascontiguousarray(tris.reshape(-1, 3))
        self.ind = np.arange(3 * n, dtype=np.uint32)
        self.mesh = (F.rt_mesh * 1)()
        self.mesh[0].p = self.p.ctypes.data_as(C.POINTER(C.c_double))
        self.mesh[0].ind = self.ind.ctypes.data_as(C.POINTER(C.c_uint32))
        self.mesh[0].n_p, self.mesh[0].n_ind = 3 * n, 3 * n
        self.prims = (F.rt_primitive * max(n + m, 1))()
        for i in range(n):
            pr = self.prims[i]
            pr.kind, pr.mat_index, pr.light_index, pr.xform_index = 1, 0, -1, -1
            pr.mesh_index, pr.tri_ind = 0, 3 * i
            lo, hi = tris[i].min(axis=0), tris[i].max(axis=0)
            pr.bbox_min[:] = lo
            pr.bbox_max[:] = hi
        for j in range(m):
            pr = self.prims[n + j]
            pr.kind, pr.mat_index, pr.light_index, pr.xform_index = 0, 0, -1, -1
            c, r = spheres[j, :3], spheres[j, 3]
            pr.v[0], pr.v[1], pr.v[2], pr.v[3] = c[0], c[1], c[2], r
            pr.bbox_min[:] = c - r   # Primitive::new_sphere's box: centre -/+ r (primitive.rs:66-68)
            pr.bbox_max[:] = c + r
        self.tex = (F.rt_texture * 1)()
        self.tex[0].kind = 0
        self.tex[0].color[:] = (0.5, 0.5, 0.5)
        self.mat = (F.rt_material * 1)()
        self.mat[0].kind = 0
        self.mat[0].tex[0] = 0
        self._desc = F.rt_scene_desc()
        self._desc.meshes, self._desc.n_meshes = self.mesh, 1 if n else 0
        self._desc.prims, self._desc.n_prims = self.prims, n + m
        self._desc.materials, self._desc.n_materials = self.mat, 1
        self._desc.textures, self._desc.n_textures = self.tex, 1
        self.n_prims = n + m

    @property
    def desc(self):
        return C.pointer(self._desc)

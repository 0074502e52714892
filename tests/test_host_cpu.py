"""CPU-side tests: deterministic math, host scene mirror, C-ABI surface (no GPU compute)."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import rustraytracer_amd as rr
from rustraytracer_amd import _ffi as F
from tests import oracle_ffi as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PI = 3.14159265358979


def _dm(fn, x, y=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), dtype=np.float64)
    out = np.empty_like(x)
    dp = C.POINTER(C.c_double)
    O.lib().oracle_detmath(fn, x.ctypes.data_as(dp), y.ctypes.data_as(dp), x.size, out.ctypes.data_as(dp))
    return out


def _ulps(a, b):
    return np.abs(a - b) / np.spacing(np.abs(b))


def test_detmath_close_to_libm():
    # include/rt_detmath.h pins ONE implementation; it must stay within 2 ulp of the platform libm
    rng = np.random.default_rng(0)
    x = rng.uniform(-7, 7, 200000)
    assert _ulps(_dm(0, x), np.sin(x)).max() <= 2
    assert _ulps(_dm(1, x), np.cos(x)).max() <= 2
    big = rng.uniform(0, 70000, 200000)  # Checkered: frequency 1e4 * u * 2*pi
    assert _ulps(_dm(0, big), np.sin(big)).max() <= 2
    assert _ulps(_dm(1, big), np.cos(big)).max() <= 2
    lx = np.exp(rng.uniform(-30, 30, 200000))
    assert _ulps(_dm(2, lx), np.log(lx)).max() <= 2
    a = rng.uniform(-1, 1, 200000)
    b = rng.uniform(-1, 1, 200000)
    assert _ulps(_dm(3, a), np.arccos(a)).max() <= 2
    assert _ulps(_dm(4, a, b), np.arctan2(a, b)).max() <= 2
    ex = rng.uniform(-50, 50, 100000)
    assert _ulps(_dm(5, ex), np.exp(ex)).max() <= 2
    p = rng.uniform(0, 1, 100000)
    assert np.abs(_dm(6, p, np.full_like(p, 1 / 2.2)) - p ** (1 / 2.2)).max() < 1e-14
    # sqrt is IEEE-exact
    assert np.array_equal(_dm(7, lx), np.sqrt(lx))


def test_detmath_special_values():
    assert _dm(0, [0.0])[0] == 0.0 and _dm(1, [0.0])[0] == 1.0
    assert _dm(2, [1.0])[0] == 0.0
    assert _dm(3, [1.0])[0] == 0.0 and _dm(3, [-1.0])[0] == pytest.approx(math.pi)
    assert _dm(4, [0.0], [1.0])[0] == 0.0
    assert _dm(4, [1.0], [0.0])[0] == pytest.approx(math.pi / 2)
    assert _dm(4, [0.0], [-1.0])[0] == pytest.approx(math.pi)
    assert np.isnan(_dm(3, [1.5])[0])
    assert _dm(6, [0.0], [0.4545])[0] == 0.0 and _dm(6, [1.0], [0.4545])[0] == 1.0


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads and exports every function include/*.h declares."""
    L = F.lib()
    declared = set()
    for hdr in ("rt_abi.h", "rt_host.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b(rt_[a-z0-9_]+|rrh_[a-z0-9_]+)\s*\(", text))
    assert declared == set(F.ABI_SYMBOLS) | set(F.HOST_SYMBOLS), declared ^ (set(F.ABI_SYMBOLS) | set(F.HOST_SYMBOLS))
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert L.rt_abi_version() == 5
    # the shared object really contains gfx950 code
    blob = open(F.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob
    # every kernel instance the launch tables of abi.hip can pick is in the library: the class kernels (three kinds of hit
    # record) and the fused tail of the nine shading-feature masks (shading.h: kFeatVariants) in both precisions, the
    # light kernel with and without the environment, the traversal kernel with and without counters / simple leaves
    for ns in (b"3rtd", b"5rtd32"):
        for feat in (0, 1, 2, 3, 7, 19, 23, 15, 31):
            for kind in (0, 1, 2):
                assert b"_ZN%s11k_shade_clsILi%dELi%dELi" % (ns, feat, kind) in blob, (ns, feat, kind)  # (+ waves per SIMD)
            assert b"_ZN%s6k_tailILi%dELb0EEE" % (ns, feat) in blob and b"_ZN%s6k_tailILi%dELb1EEE" % (ns, feat) in blob
        for env in (0, 16):
            assert b"_ZN%s13k_shade_lightILi%dEEE" % (ns, env) in blob
    for inst in (b"k_traceILb0ELb0E", b"k_traceILb0ELb1E", b"k_traceILb1ELb0E", b"k_intersect_batch", b"k_generate",
                 b"k_resolve", b"k_tonemap", b"kb_scatter", b"kb_emit", b"k_film_pack", b"k_film_unpack"):
        assert inst in blob, inst


def test_struct_layouts_match_the_header():
    # sizes the C side static_asserts or relies on (include/rt_abi.h)
    assert C.sizeof(F.rt_primitive) == 120
    assert C.sizeof(F.rt_xform) == 192
    assert C.sizeof(F.rt_texture) == 64
    assert C.sizeof(F.rt_material) == 56
    assert C.sizeof(F.rt_light) == 64
    assert C.sizeof(F.rt_camera) == 15 * 8 + 9 * 8
    assert C.sizeof(F.rt_ray) == 64 and C.sizeof(F.rt_hit) == 16
    assert C.sizeof(F.rt_render_cfg) == 80
    assert C.sizeof(F.rt_stats) == 8 * 8 + 16 + 8 + 32 + 16 + 16 + 8 + 8


def test_error_behaviour_without_compute():
    L = F.lib()
    assert L.rt_context_create(None, 1, None) == F.RT_ERR_INVALID_ARG
    assert b"null" in L.rt_last_error()
    h = C.c_void_p()
    assert L.rt_context_create(None, 2, C.byref(h)) == F.RT_ERR_INVALID_ARG  # device_ids missing
    assert L.rt_context_create((C.c_int * 2)(0, 0), -1, C.byref(h)) == F.RT_ERR_INVALID_ARG
    # without a HIP device every context creation fails with RT_ERR_NO_DEVICE (never a CPU fallback)
    import torch
    if not torch.cuda.is_available():
        assert L.rt_context_create((C.c_int * 2)(0, 0), 2, C.byref(h)) == F.RT_ERR_NO_DEVICE
    assert L.rt_scene_create(None, C.byref(h)) == F.RT_ERR_INVALID_ARG
    assert L.rt_scene_commit(None) == F.RT_ERR_INVALID_ARG
    with pytest.raises(rr.RtError) as ei:
        rr.Scene("no_such_scene")
    assert "Unknown scene" in str(ei.value)  # main.rs:355-357
    with pytest.raises(rr.RtError):
        rr.plastic_dragon(mesh_path="/nonexistent/dragon.obj")


def _prims(sc):
    d = sc.desc.contents
    return [d.prims[i] for i in range(d.n_prims)]


def test_cornell_box_preset_matches_scenes_rs():
    # scenes.rs:89-197
    sc = rr.cornell_box()
    d = sc.desc.contents
    assert (d.n_prims, d.n_materials, d.n_textures, d.n_lights, d.n_meshes, d.n_xforms) == (18, 4, 4, 1, 0, 2)
    pr = _prims(sc)
    assert [p.kind for p in pr[:6]] == [F.RT_PRIM_YZ_RECT, F.RT_PRIM_YZ_RECT, F.RT_PRIM_XZ_RECT, F.RT_PRIM_XZ_RECT,
                                        F.RT_PRIM_XZ_RECT, F.RT_PRIM_XY_RECT]
    assert [p.flip for p in pr[:6]] == [1, 0, 1, 0, 1, 1]
    assert [p.mat_index for p in pr[:6]] == [2, 0, 3, 1, 1, 1]
    assert [p.light_index for p in pr[:6]] == [-1, -1, 0, -1, -1, -1]
    assert list(pr[2].v) == [213.0, 227.0, 343.0, 332.0, 554.9]
    # rect AABB padded by SMALL along the normal (primitive.rs:161-164)
    assert list(pr[2].bbox_min) == [213.0, 554.9 - 0.001, 227.0] and list(pr[2].bbox_max) == [343.0, 554.9 + 0.001, 332.0]
    # Cube::get_sides order z0 z1 y0 y1 x0 x1 with FlipFace on the min sides (hittable.rs:788-846)
    assert [p.kind for p in pr[6:12]] == [F.RT_PRIM_XY_RECT] * 2 + [F.RT_PRIM_XZ_RECT] * 2 + [F.RT_PRIM_YZ_RECT] * 2
    assert [p.flip for p in pr[6:12]] == [1, 0, 1, 0, 1, 0]
    assert all(p.xform_index == 0 for p in pr[6:12]) and all(p.xform_index == 1 for p in pr[12:18])
    # cube1 uses second_transform = translate(130,0,65) * rotY(-18 deg)
    xf = d.xforms[0]
    c, s = math.cos(-18 * PI / 180), math.sin(-18 * PI / 180)
    assert list(xf.fwd) == pytest.approx([c, 0, s, 130, 0, 1, 0, 0, -s, 0, c, 65], abs=1e-15)
    m = np.array(xf.fwd[:]).reshape(3, 4)
    mi = np.array(xf.inv[:]).reshape(3, 4)
    assert np.allclose(m[:, :3] @ mi[:, :3], np.eye(3), atol=1e-15)
    assert np.allclose(m[:, :3] @ mi[:, 3] + m[:, 3], 0, atol=1e-12)
    lt = d.lights[0]
    assert (lt.prim_index, lt.two_sided, list(lt.color), lt.area) == (2, 0, [15.0] * 3, 130.0 * 105.0)
    assert list(d.textures[0].color) == [0.65, 0.05, 0.05]
    assert d.materials[3].kind == F.RT_MAT_LIGHT and d.materials[0].kind == F.RT_MAT_MATTE
    assert sc.name == "cornell_box.png"


def test_camera_matches_geometry_rs():
    # geometry.rs:133-175 for the Cornell camera: from (278,278,-800) to (278,278,0), vfov 40, focus 10
    sc1 = rr.cornell_box(aspect_ratio=1.0)  # keep the scene alive: camera points into it
    cam = sc1.camera.contents
    assert list(cam.origin) == [278.0, 278.0, -800.0]
    assert list(cam.w) == [0.0, 0.0, 1.0]
    assert list(cam.u) == [-1.0, 0.0, 0.0]  # u = -normalize(up x w)
    assert list(cam.v) == pytest.approx([0.0, 1.0, 0.0])
    h = math.tan(40 * PI / 180 / 2)
    assert cam.horizontal_offset[0] == pytest.approx(-2 * h * 10, rel=1e-14)
    assert cam.vertical_offset[1] == pytest.approx(2 * h * 10, rel=1e-14)
    ulc = np.array(cam.origin[:]) - np.array(cam.horizontal_offset[:]) / 2 + np.array(cam.vertical_offset[:]) / 2 + [0, 0, 10]
    assert list(cam.upper_left_corner) == pytest.approx(list(ulc), rel=1e-14)
    assert (cam.lens_radius, cam.t0, cam.t1) == (0.0, 0.0, 1.0)
    # aspect ratio widens only the horizontal offset
    sc2 = rr.cornell_box(aspect_ratio=2.0)
    cam2 = sc2.camera.contents
    assert cam2.horizontal_offset[0] == pytest.approx(2 * cam.horizontal_offset[0])
    assert cam2.vertical_offset[1] == cam.vertical_offset[1]
    # rrh_camera_new is the same constructor
    out = F.rt_camera()
    F.lib().rrh_camera_new(O.vec(278, 278, -800), O.vec(278, 278, 0), O.vec(0, 1, 0), 1.0, 40.0, 0.0, 10.0, 0.0, 1.0,
                           C.byref(out))
    assert bytes(out) == bytes(cam)


def test_mesh_presets_and_variants():
    sc = rr.cornell_box_statue(mesh_faces=500, variant=0)
    d = sc.desc.contents
    assert d.n_prims == 6 + 500 and d.n_meshes == 1
    assert d.materials[3].kind == F.RT_MAT_MATTE  # scenes.rs:243
    assert d.lights[0].two_sided == 1 and list(d.lights[0].color) == pytest.approx([0.97 * 25, 0.92 * 25, 0.23 * 25])
    assert d.prims[2].flip == 0 and d.prims[2].mat_index == 0  # bare XZRect light with a matte material
    sc_m = rr.cornell_box_statue(mesh_faces=20, variant=1)
    assert sc_m.desc.contents.materials[3].kind == F.RT_MAT_METAL
    # the stand-in mesh sits inside the box
    m = d.meshes[0]
    p = np.ctypeslib.as_array(m.p, shape=(m.n_p, 3))
    assert p.min() > 0 and p.max() < 555
    # triangles: bbox = min/max of the three vertices (hittable.rs:264-277), tri_ind = 3*face
    ind = np.ctypeslib.as_array(m.ind, shape=(m.n_ind,))
    for f in (0, 17, 499):
        pr = d.prims[6 + f]
        assert pr.kind == F.RT_PRIM_TRIANGLE and pr.tri_ind == 3 * f and pr.mat_index == 3
        v = p[ind[3 * f:3 * f + 3]]
        assert list(pr.bbox_min) == list(v.min(0)) and list(pr.bbox_max) == list(v.max(0))

    sc_dr = rr.plastic_dragon(mesh_faces=300, variant=1)
    dr = sc_dr.desc.contents  # C3: metal
    assert dr.n_prims == 1 + 300 + 1
    assert dr.materials[1].kind == F.RT_MAT_METAL and dr.materials[1].remap_roughness == 1
    assert list(dr.textures[dr.materials[1].tex[0]].color) == [0.05, 0.5, 0.75]
    assert dr.textures[2].kind == F.RT_TEX_CHECKERED and dr.textures[2].frequency == 10000.0
    assert (dr.textures[2].even, dr.textures[2].odd) == (0, 1)
    light = dr.prims[dr.n_prims - 1]
    assert light.flip == 1 and light.light_index == 0 and list(light.v) == [-5, -5, 5, 5, 15]
    assert dr.lights[0].prim_index == dr.n_prims - 1 and dr.lights[0].area == 100.0
    sc_gl = rr.plastic_dragon(mesh_faces=20, variant=2)
    gl = sc_gl.desc.contents.materials[1]
    assert gl.kind == F.RT_MAT_GLASS and gl.f[2] == 1.5
    sc_pl = rr.plastic_dragon(mesh_faces=20, variant=0)
    pl = sc_pl.desc.contents.materials[1]
    assert pl.kind == F.RT_MAT_PLASTIC and pl.f[0] == 0.001

    sc_td = rr.two_dragons(mesh_faces=100, variant=0)
    td = sc_td.desc.contents  # C4: both dragons
    assert td.n_prims == 2 + 200 and td.n_meshes == 2
    assert td.prims[1].flip == 0 and td.prims[1].light_index == 0  # bare, one-sided emitter
    assert td.prims[2].mat_index == 2 and td.prims[2 + 100].mat_index == 3
    assert td.materials[2].kind == F.RT_MAT_GLASS and td.materials[3].kind == F.RT_MAT_METAL
    sc_td1 = rr.two_dragons(mesh_faces=100, variant=1)
    assert sc_td1.desc.contents.n_prims == 2 + 100  # as committed


def test_procedural_mesh_properties():
    sc = rr.plastic_dragon(mesh_faces=1280)  # exactly 20*8^2: no trimming
    m = sc.desc.contents.meshes[0]
    assert m.n_ind == 3 * 1280 and m.n_p == 10 * 64 + 2 and m.n_n == m.n_p and m.n_uv == 0
    p = np.ctypeslib.as_array(m.p, shape=(m.n_p, 3))
    ind = np.ctypeslib.as_array(m.ind, shape=(m.n_ind,)).reshape(-1, 3)
    # closed genus-0 surface: every edge is shared by exactly two faces
    e = np.sort(np.concatenate([ind[:, [0, 1]], ind[:, [1, 2]], ind[:, [2, 0]]]), axis=1)
    _, counts = np.unique(e, axis=0, return_counts=True)
    assert (counts == 2).all()
    # bbox = 10 * [-0.5,0.5]*(1,0.7,0.45) (+0.67 in y), vertices are f32 values times the scale
    ext = p.max(0) - p.min(0)
    assert ext == pytest.approx([10.0, 7.0, 4.5], rel=1e-6)
    q = p[:, 0] / 10.0
    assert np.array_equal(q.astype(np.float32).astype(np.float64), q)
    # normals point outwards (positive dot with the face normal on average) and are scaled by the transform
    n = np.ctypeslib.as_array(m.n, shape=(m.n_n, 3))
    fn = np.cross(p[ind[:, 1]] - p[ind[:, 0]], p[ind[:, 2]] - p[ind[:, 0]])
    assert (np.einsum("ij,ij->i", fn, n[ind[:, 0]]) > 0).mean() > 0.99
    assert np.linalg.norm(n, axis=1) == pytest.approx(10.0, rel=1e-6)
    # deterministic
    sc2 = rr.plastic_dragon(mesh_faces=1280)
    m2 = sc2.desc.contents.meshes[0]
    assert np.array_equal(np.ctypeslib.as_array(m2.p, shape=(m2.n_p, 3)), p)
    # trimmed to exactly N
    sc3 = rr.plastic_dragon(mesh_faces=1000)
    assert sc3.desc.contents.meshes[0].n_ind == 3000


def test_obj_parser_tobj_semantics(tmp_path):
    # parser.rs:8-87 / tobj: first model only, fan triangulation, negative indices, transform baked in
    obj = tmp_path / "quad.obj"
    obj.write_text(
        "# quad + a second object that must be ignored\n"
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\n"
        "vn 0 0 1\n"
        "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
        "o first\n"
        "f 1/1/1 2/2/1 3/3/1 4/4/1\n"
        "o second\n"
        "v 5 5 5\nv 6 5 5\nv 5 6 5\n"
        "f -3 -2 -1\n")
    sc = rr.plastic_dragon(mesh_path=str(obj))
    d = sc.desc.contents
    m = d.meshes[0]
    assert m.n_ind == 6 and m.n_p == 4 and m.n_n == 4 and m.n_uv == 4
    assert list(np.ctypeslib.as_array(m.ind, shape=(6,))) == [0, 1, 2, 0, 2, 3]
    p = np.ctypeslib.as_array(m.p, shape=(4, 3))
    assert np.array_equal(p, np.array([[0, 0, 0], [10, 0, 0], [10, 10, 0], [0, 10, 0]], dtype=float))  # Similarity scale 10
    n = np.ctypeslib.as_array(m.n, shape=(4, 3))
    assert np.array_equal(n[0], [0, 0, 10.0])  # transform_vector, not renormalised (parser.rs:45)
    uv = np.ctypeslib.as_array(m.uv, shape=(4, 2))
    assert np.array_equal(uv[2], [1.0, 1.0])
    assert d.n_prims == 1 + 2 + 1
    # the oracle uses the mesh uvs (hittable.rs:462-466)
    osc = O.OracleScene(sc)
    rec = osc.prim_intersect(1, (7.5, 2.5, 5.0), (0, 0, -1.0))
    assert rec.hit == 1 and list(rec.uv) == pytest.approx([0.75, 0.25])


def test_resolve_rgb8_reference_formula():
    # util.rs:400-408, 441-471 via the oracle: ACES approx on 0.6*x, gamma 1/2.2, round(.*256) saturating
    rgb = np.array([[0.0, 0.18, 1.0], [4.0, 100.0, 0.5]], dtype=np.float64) * 2
    n = np.array([2, 2], dtype=np.uint32)
    out = np.zeros((2, 3), dtype=np.uint8)
    O.lib().oracle_resolve_rgb8(rgb.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p), 2, out.ctypes.data_as(C.c_void_p))

    def ref(x):
        x *= 0.6
        y = min(max((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14), 0.0), 1.0)
        return min(255, max(0, round(y ** (1 / 2.2) * 256)))

    want = [[ref(v) for v in row] for row in [[0.0, 0.18, 1.0], [4.0, 100.0, 0.5]]]
    assert out.tolist() == want
    assert out[0, 0] == 0 and out[1, 1] == 255


def _write_hdr(path, w, h, px):
    """Radiance RGBE file with new-style RLE scanlines; px: (h, w, 4) uint8."""
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        for y in range(h):
            f.write(bytes([2, 2, w >> 8, w & 255]))
            for c in range(4):
                row = px[y, :, c]
                x = 0
                while x < w:
                    run = 1
                    while x + run < w and run < 127 and row[x + run] == row[x]:
                        run += 1
                    if run >= 3:
                        f.write(bytes([128 + run, int(row[x])]))
                        x += run
                    else:
                        lit = min(w - x, 20)
                        f.write(bytes([lit]) + bytes(int(v) for v in row[x:x + lit]))
                        x += lit


def test_material_hdr_preset_and_hdr_reader(tmp_path):
    """scenes.rs:627-741 + Texture::new_hdr (material.rs:631-641): file -> Rgb<f32> -> to_rgbe8 texels."""
    sc = rr.material_hdr(1, mesh_faces=1000)
    d = sc.desc.contents
    assert d.n_lights == 1 and d.lights[0].kind == 1 and d.lights[0].tex_index == 0
    assert d.lights[0].world_radius == 10000.0 and d.lights[0].xform_index == -1
    assert d.textures[0].kind == 2 and d.textures[0].width == 512 and d.textures[0].height == 256
    assert d.n_prims == 3 * 1000 + 1 and d.n_materials == 3 and d.n_textures == 1 + 3 + 4
    assert d.prims[d.n_prims - 1].kind == 2 and d.prims[d.n_prims - 1].xform_index >= 0   # the transformed floor rect
    assert d.materials[0].kind == 4 and d.materials[2].kind == 0
    variants = [rr.material_hdr(k, mesh_faces=500) for k in range(4)]   # kept alive: desc points into them
    assert [v.desc.contents.materials[0].kind for v in variants] == [2, 4, 5, 3]
    g = variants[3].desc.contents.materials[0]
    assert (g.f[0], g.f[1], g.f[2]) == (0.01, 0.01, 1.5) and g.remap_roughness == 1
    with pytest.raises(Exception):
        rr.material_hdr(7)
    # a data directory with an envmap: normalised RGBE texels survive the f32 round trip unchanged
    root = tmp_path / "material"
    (root / "textures").mkdir(parents=True)
    rng = np.random.default_rng(2)
    w, h = 40, 12
    px = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    px[..., 3] = rng.integers(100, 150, size=(h, w))
    px[..., 0] |= 128                     # normalised: the largest mantissa has its top bit set
    px[:, 5:15, :] = px[:, 5:6, :]        # a run, to exercise the RLE branch
    px[3, 20] = (0, 0, 0, 0)              # black texel
    px[4, 21] = (7, 3, 1, 130)            # denormalised: to_rgbe8 renormalises (7 -> 224 = x 32, e - 5)
    _write_hdr(root / "textures" / "envmap.hdr", w, h, px)
    sc2 = rr.material_hdr(0, mesh_faces=500, data_dir=str(root))
    t = sc2.desc.contents.textures[0]
    assert (t.width, t.height) == (w, h)
    got = np.ctypeslib.as_array(t.rgbe, shape=(h, w, 4))
    expect = px.copy()
    expect[4, 21] = (7 * 32, 3 * 32, 1 * 32, 125)
    assert np.array_equal(got, expect)


def test_reference_envmap_if_present():
    """The reference's own data/material/textures/envmap.hdr (exists only in the build container)."""
    root = "/root/reference/data/material"
    if not os.path.exists(root + "/textures/envmap.hdr"):
        pytest.skip("reference data not on this machine")
    sc = rr.material_hdr(2, mesh_faces=500, data_dir=root)
    d = sc.desc.contents
    t = d.textures[0]
    assert t.width >= 256 and t.height * 2 == t.width
    got = np.ctypeslib.as_array(t.rgbe, shape=(t.height, t.width, 4))
    assert (got[..., 3] > 0).mean() > 0.9     # a studio map: a few black texels, the rest lit
    # Mesh000/001.obj are in the checkout, Mesh002.obj is not (-> procedural): 3 meshes either way
    assert d.n_meshes == 3 and d.meshes[0].n_ind > 3 * 500


def _read_png(path):
    """Minimal reader for what rrh_write_png emits (8-bit RGB, filter 0) using zlib only."""
    import struct
    import zlib
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(typ + body) == crc
        if typ == b"IHDR":
            w, h, depth, ctype, comp, filt, inter = struct.unpack(">IIBBBBB", body)
            assert (depth, ctype, comp, filt, inter) == (8, 2, 0, 0, 0)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)


def test_png_writer_round_trip(tmp_path):
    """next-row f1: the 8-bit picture leaves as a PNG (util.rs:387-398); checked with zlib's own crc/adler/inflate."""
    rng = np.random.default_rng(4)
    for w, h in ((1, 1), (37, 19), (300, 200)):   # the last one spans several 64 KiB stored blocks
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        p = tmp_path / f"t{w}.png"
        rr.write_png(p, img)
        assert np.array_equal(_read_png(p), img)
    with pytest.raises(Exception):
        rr.write_png(tmp_path / "no_such_dir" / "x.png", img)


def test_teapot_hdr_preset():
    """scenes.rs:744-808 teapot_hdr(): camera, one plastic material (roughness 1e-5, remapped) on two meshes, the
    checkered matte floor rect with its transform, the infinite light on texture 0."""
    sc = rr.teapot_hdr(16 / 9, mesh_faces=4000)
    d = sc.desc.contents
    assert sc.name == "teapot_hdr.png"
    assert d.n_meshes == 2 and d.n_lights == 1 and d.lights[0].kind == 1 and d.lights[0].tex_index == 0
    assert d.n_prims == 4000 + 4000 // 8 + 1
    assert d.n_textures == 1 + 2 + 4 and d.n_materials == 3
    m = d.materials[0]
    assert m.kind == 2 and m.f[0] == 0.00001 and m.remap_roughness == 1          # make_plastic(.., 0.00001, true)
    assert [d.prims[i].mat_index for i in (0, 4000 // 8, d.n_prims - 1)] == [0, 0, 2]
    fl = d.prims[d.n_prims - 1]
    assert fl.kind == 2 and fl.xform_index >= 0 and tuple(fl.v[:5]) == (-1.0, -1.0, 1.0, 1.0, 0.0)
    assert d.textures[6].kind == 1 and (d.textures[6].even, d.textures[6].odd) == (4, 5) and d.textures[6].frequency == 10.0  # new_checkered(even, odd, f), material.rs:621
    cam = sc.camera.contents
    assert tuple(cam.origin) == (23.895, 11.2207, 0.0400773) and cam.lens_radius == 0.0
    # the oracle renders it (environment-lit plastic): finite, and the teapot darkens the middle of the frame
    from tests import oracle_ffi as O
    r, n, s = O.OracleScene(sc).render(sc.camera, rr.make_cfg(48, 27, 4, seed=2))
    assert np.isfinite(r).all() and s.rays > 48 * 27 * 4


def test_committed_reference_assets_round_trip(tmp_path):
    """tests/golden/assets: the reference's Mesh000/001.obj and envmap.hdr as committed fixtures (tests/assets.py) go
    through parse_obj / Texture::new_hdr exactly like the files of the checkout."""
    from tests import assets
    root = assets.material_dir(tmp_path)
    sc = rr.material_hdr(1, data_dir=root, mesh_faces=2000)
    d = sc.desc.contents
    assert d.n_meshes == 3
    # Mesh001.obj: 17536 faces / 35072 vertices with normals and uvs; Mesh000.obj: 13312 / 26624; Mesh002 = stand-in
    assert (d.meshes[0].n_ind // 3, d.meshes[0].n_p, d.meshes[0].n_n, d.meshes[0].n_uv) == (17536, 35072, 35072, 35072)
    assert (d.meshes[2].n_ind // 3, d.meshes[2].n_p) == (13312, 26624)
    assert d.meshes[1].n_ind // 3 == 2000 and d.meshes[1].n_uv == 0
    assert (d.textures[0].width, d.textures[0].height) == (1024, 512)
    if os.path.exists("/root/reference/data/material/models/Mesh000.obj"):   # build container: identical to the checkout
        sc_ref = rr.material_hdr(1, data_dir="/root/reference/data/material", mesh_faces=2000)
        dr = sc_ref.desc.contents
        for k in (0, 2):
            a = np.ctypeslib.as_array(d.meshes[k].p, shape=(d.meshes[k].n_p, 3))
            b = np.ctypeslib.as_array(dr.meshes[k].p, shape=(dr.meshes[k].n_p, 3))
            assert np.array_equal(a, b)
        ta = np.ctypeslib.as_array(d.textures[0].rgbe, shape=(512, 1024, 4))
        tb = np.ctypeslib.as_array(dr.textures[0].rgbe, shape=(512, 1024, 4))
        assert np.array_equal(ta, tb)


def _build_c_example(tmp_path):
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "gpu_tile")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O2", "-I" + os.path.join(root, "include"),
           os.path.join(root, "examples", "gpu_tile.c"), "-o", exe, "-L" + os.path.join(root, "rustraytracer_amd"),
           "-l:librt_amd.so", "-Wl,-rpath," + os.path.join(root, "rustraytracer_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def test_c_host_compiles_as_c99_and_fails_loudly_without_a_device(tmp_path):
    """include/rt_abi.h + rt_host.h are plain C (the reference's Rust side binds them through `extern "C"`): a complete
    host in C99 (examples/gpu_tile.c: scene -> rt_render -> PNG) compiles with -pedantic -Werror against them and
    links librt_amd.so.  Without a HIP device the product path has no fallback: RT_ERR_NO_DEVICE, exit code 1."""
    import subprocess
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe, "cornell_box", "16", "16", "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:   # (a box with a GPU renders the 16x16 picture instead)
        assert r.returncode == 1 and "no HIP device" in r.stdout, r.stdout
    else:
        assert "film_fnv" in r.stdout


def test_bench_names_the_kernel_with_the_largest_share_and_never_reports_a_fraction_above_one():
    """bench.py: pick_roofline -- `kernel` = largest share of device time; `frac` = the measured HBM
    fraction when the counter profile is current, else the algorithmic one, and never an algorithmic figure above 1."""
    import bench
    ks = {"k_trace": {"share_of_device_time": 0.45, "frac_algorithmic": 1.2, "hbm_frac_rocprof": 0.47, "hbm_read_frac_rocprof": 0.44, "achieved": 9600.0},
          "k_shade": {"share_of_device_time": 0.51, "frac_algorithmic": 0.46, "hbm_frac_rocprof": 0.31, "hbm_read_frac_rocprof": 0.2, "achieved": 3700.0},
          "k_classify": {"share_of_device_time": 0.04, "frac_algorithmic": 0.5, "hbm_frac_rocprof": None, "achieved": 4000.0}}
    top = bench.pick_roofline(ks)
    assert top["kernel"] == "k_shade" and top["frac"] == 0.31 and top["frac_algorithmic_incl_cache_hits"] == 0.46
    ks["k_trace"]["share_of_device_time"] = 0.6
    top = bench.pick_roofline(ks)
    assert top["kernel"] == "k_trace" and top["frac"] == 0.47 and top["frac_algorithmic_incl_cache_hits"] == 1.2
    ks["k_trace"]["hbm_frac_rocprof"] = None  # stale / missing counter profile: an algorithmic 1.2 is not a fraction
    top = bench.pick_roofline(ks)
    assert top["frac"] is None and top["frac_algorithmic_incl_cache_hits"] == 1.2 and "withheld" in top["frac_is"]
    ks["k_trace"]["frac_algorithmic"] = 0.8
    assert bench.pick_roofline(ks)["frac"] == 0.8


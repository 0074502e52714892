"""GPU tests (-m gpu) of the ABI-v2 additions: multi-device contexts, cancellation, stream ordering of
rt_render_device, and the fused tail launch's share of the counters."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

import rustraytracer_amd as rr
from rustraytracer_amd import _ffi as F

pytestmark = pytest.mark.gpu


def _need_gpus(ids):
    import torch
    have = torch.cuda.device_count()
    if max(ids) >= have:
        pytest.skip(f"needs {max(ids) + 1} GPUs, this box has {have}")


# (0, 1) and (0, 1, 2, 3): DISTINCT devices -- hipDeviceEnablePeerAccess, hipMemcpyPeerAsync between two GPUs and the
# cross-device hipStreamWaitEvent of render_multi; they arm themselves on a multi-GPU box and skip on a one-GPU box
@pytest.mark.parametrize("ids", [(0, 0), (0, 0, 0), (0, 1), (0, 1, 2, 3)], ids=["2ctx", "3ctx", "gpu01", "gpu0123"])
def test_multi_device_context_is_bit_identical(gpu_ctx, ids):
    _need_gpus(ids)
    """rt_context_create(device_ids, n > 1) (SURVEY.md 8b): one host thread per device, scene replicated at commit,
    tiles interleaved, peers' pixels packed + copied peer-to-peer + scattered into the caller's film.  A one-GPU box
    exercises it with the same device id repeated (each entry gets its own context, streams and pools)."""
    sc = rr.cornell_box_statue(mesh_faces=6000, variant=3)
    gs1 = gpu_ctx.upload(sc)
    cfg = rr.make_cfg(80, 56, 8, seed=12)
    r1, n1, s1 = gpu_ctx.render(gs1, sc.camera, cfg)
    mc = rr.Context(list(ids))
    gsm = mc.upload(sc)
    rm, nm, sm = mc.render(gsm, sc.camera, cfg)
    assert sm.n_devices == len(ids) and s1.n_devices == 1
    assert np.array_equal(rm, r1) and np.array_equal(nm, n1)
    assert (sm.paths, sm.rays_extension, sm.rays_shadow, sm.rays_probe, sm.vertices_shaded) == \
           (s1.paths, s1.rays_extension, s1.rays_shadow, s1.rays_probe, s1.vertices_shaded)
    assert sm.gather_ms > 0.0
    # combined with a caller-level split (two "processes" of a multi-device context each)
    acc, nacc = np.zeros_like(r1), np.zeros_like(n1)
    for r in range(2):
        a, b, _ = mc.render(gsm, sc.camera, rr.make_cfg(80, 56, 8, seed=12, tile_rank=r, tile_world=2))
        acc += a
        nacc += b
    assert np.array_equal(acc, r1) and np.array_equal(nacc, n1)
    # progressive passes across devices: running sums continue in sample order
    film = (np.zeros_like(r1), np.zeros_like(n1))
    for first in (0, 2, 6):
        mc.render(gsm, sc.camera, rr.make_cfg(80, 56, 8, seed=12, sample_first=first,
                                               sample_count={0: 2, 2: 4, 6: 2}[first], accumulate=True), film=film)
    assert np.array_equal(film[0], r1) and np.array_equal(film[1], n1)
    # the device-built tree is replicated as well
    gsd = mc.upload(sc, device_build=True)
    rd, nd, _ = mc.render(gsd, sc.camera, cfg)
    assert np.array_equal(rd, r1)
    for g in (gsd, gsm, gs1):
        g.close()
    mc.close()


def test_multi_device_empty_progressive_pass(gpu_ctx):
    """A progressive pass past the last sample (sample_first >= next_pow2(spp)) renders nothing; on a multi-device
    context the gather still runs over every peer's pixel list, which therefore has to be this render's list even
    though no kernel of the pass used it (a fresh peer had none, a reused one a stale list of another
    image size).  The caller's film must come back unchanged."""
    sc = rr.cornell_box()
    mc = rr.Context([0, 0, 0])
    gsm = mc.upload(sc)
    # a first render of ANOTHER size leaves a longer pixel list on the peers
    mc.render(gsm, sc.camera, rr.make_cfg(96, 80, 2, seed=1))
    cfg = rr.make_cfg(48, 32, 4, seed=1)
    r1, n1, _ = mc.render(gsm, sc.camera, cfg)
    film = (r1.copy(), n1.copy())
    _, _, st = mc.render(gsm, sc.camera, rr.make_cfg(48, 32, 4, seed=1, sample_first=4, sample_count=2, accumulate=True),
                         film=film)
    assert st.paths == 0 and st.rays == 0
    assert np.array_equal(film[0], r1) and np.array_equal(film[1], n1)
    # ... also as the very first call on a fresh multi-device context (peers without any pixel list yet)
    mc2 = rr.Context([0, 0])
    gs2 = mc2.upload(sc)
    film2 = (r1.copy(), n1.copy())
    mc2.render(gs2, sc.camera, rr.make_cfg(48, 32, 4, seed=1, sample_first=8, accumulate=True), film=film2)
    assert np.array_equal(film2[0], r1) and np.array_equal(film2[1], n1)
    for g, c in ((gs2, mc2), (gsm, mc)):
        g.close()
        c.close()


def test_multi_device_render_device_film(gpu_ctx):
    import torch
    sc = rr.cornell_box()
    cfg = rr.make_cfg(64, 64, 4, seed=1)
    gs1 = gpu_ctx.upload(sc)
    r1, n1, _ = gpu_ctx.render(gs1, sc.camera, cfg)
    mc = rr.Context([0, 0])
    gsm = mc.upload(sc)
    d_rgb = torch.full((64, 64, 3), 7.0, dtype=torch.float64, device="cuda")
    d_n = torch.full((64, 64), 9, dtype=torch.int32, device="cuda")
    st = mc.render_device(gsm, sc.camera, cfg, d_rgb.data_ptr(), d_n.data_ptr(),
                          stream=torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(d_rgb.cpu().numpy(), r1) and np.array_equal(d_n.cpu().numpy().astype(np.uint32), n1)
    assert st.n_devices == 2
    gsm.close()
    gs1.close()
    mc.close()


def test_cancel_flag(gpu_ctx):
    """rt_render_cfg.cancel (render.rs:93 stop_render): set before the call, and from another thread during it."""
    sc = rr.cornell_box_statue(mesh_faces=20000, variant=0)
    gs = gpu_ctx.upload(sc)
    ref = gpu_ctx.render(gs, sc.camera, rr.make_cfg(64, 64, 8, seed=2))
    flag = C.c_int32(1)
    with pytest.raises(rr.RtError) as ei:
        gpu_ctx.render(gs, sc.camera, rr.make_cfg(64, 64, 8, seed=2, cancel=flag))
    assert ei.value.code == F.RT_ERR_CANCELLED
    # the context stays usable and exact after a cancelled call
    again = gpu_ctx.render(gs, sc.camera, rr.make_cfg(64, 64, 8, seed=2))
    assert np.array_equal(again[0], ref[0]) and np.array_equal(again[1], ref[1])
    # an unset flag changes nothing
    flag = C.c_int32(0)
    same = gpu_ctx.render(gs, sc.camera, rr.make_cfg(64, 64, 8, seed=2, cancel=flag))
    assert np.array_equal(same[0], ref[0])
    # mid-render: a long render (many batches of a small pool), cancelled from a second thread
    flag = C.c_int32(0)
    big = rr.make_cfg(512, 512, 256, seed=2, cancel=flag, paths_in_flight=1 << 18)

    def stopper():
        time.sleep(0.05)
        flag.value = 1

    th = threading.Thread(target=stopper)
    t0 = time.time()
    th.start()
    with pytest.raises(rr.RtError) as ei:
        gpu_ctx.render(gs, sc.camera, big)
    th.join()
    assert ei.value.code == F.RT_ERR_CANCELLED
    assert time.time() - t0 < 5.0  # an uncancelled run of this configuration takes much longer
    again = gpu_ctx.render(gs, sc.camera, rr.make_cfg(64, 64, 8, seed=2))
    assert np.array_equal(again[0], ref[0]) and np.array_equal(again[1], ref[1])
    gs.close()


def test_render_device_is_ordered_on_the_callers_stream(gpu_ctx):
    """rt_render_device enqueues everything that touches the film on the caller's stream (NULL = the null stream):
    work queued before the call does not clobber the film, work queued after sees it."""
    import torch
    sc = rr.cornell_box_statue(mesh_faces=5000, variant=1)
    gs = gpu_ctx.upload(sc)
    cfg = rr.make_cfg(96, 96, 8, seed=6)
    ref, nref, _ = gpu_ctx.render(gs, sc.camera, cfg)
    for stream in (None, torch.cuda.Stream()):
        d_rgb = torch.zeros((96, 96, 3), dtype=torch.float64, device="cuda")
        d_n = torch.zeros((96, 96), dtype=torch.int32, device="cuda")
        junk = torch.ones((4096, 4096), dtype=torch.float64, device="cuda")
        ctxmgr = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctxmgr:
            s = torch.cuda.current_stream()
            for _ in range(6):  # a long queue of earlier work that writes the film buffers
                junk = junk @ junk * 1e-4
                d_rgb.fill_(123.0)
                d_n.fill_(77)
            gpu_ctx.render_device(gs, sc.camera, cfg, d_rgb.data_ptr(), d_n.data_ptr(), stream=s.cuda_stream)
            doubled = d_rgb * 2.0  # queued after the call on the same stream
        torch.cuda.synchronize()
        assert np.array_equal(d_rgb.cpu().numpy(), ref)
        assert np.array_equal(d_n.cpu().numpy().astype(np.uint32), nref)
        assert np.array_equal(doubled.cpu().numpy(), ref * 2.0)
    gs.close()


def test_tail_launch_share_of_the_counters(gpu_ctx):
    sc = rr.cornell_box_statue(mesh_faces=20000, variant=0)
    gs = gpu_ctx.upload(sc)
    st = gpu_ctx.render(gs, sc.camera, rr.make_cfg(128, 128, 8, seed=3, count_traversal=True))[2]
    assert 0 < st.tail_rays <= st.rays
    assert 0 < st.tail_nodes_fetched <= st.nodes_fetched
    assert st.tail_tris_tested <= st.tris_tested and st.tail_others_tested <= st.others_tested
    assert st.shade_launches > 0 and st.shade_ms > 0.0 and st.trace_ms > 0.0
    # without the counting flag the ray split is still reported, the traversal counters are not
    st2 = gpu_ctx.render(gs, sc.camera, rr.make_cfg(128, 128, 8, seed=3))[2]
    assert st2.tail_rays == st.tail_rays and st2.nodes_fetched == 0 and st2.tail_nodes_fetched == 0
    # every ray through the fused launch (tiny image: the pool never exceeds the tail threshold after the top-up)
    st3 = gpu_ctx.render(gs, sc.camera, rr.make_cfg(16, 16, 2, seed=3, count_traversal=True))[2]
    assert st3.tail_rays + 16 * 16 * 2 >= st3.rays - st3.rays_shadow - st3.rays_probe or st3.tail_rays > 0
    gs.close()


@pytest.mark.parametrize("name", ["two_dragons", "cornell_box_statue", "hdr"])
def test_light_kernel_on_the_side_stream_changes_nothing(gpu_ctx, monkeypatch, name):
    """run_lane: the kernel for escaped / fold-only paths runs on the lane's side stream, beside the class kernels and
    the next traversal launch (default), ordered by events against the scatter that rewrites its lists, the fused tail
    and the end of the lane; the class kernels of a bounce alternate between two streams and are joined before the next
    bounce is planned.  Same film, counts and counters as the serial schedule (RT_LIGHT_OVERLAP=0), with small pools
    (many iterations, the lists rewritten every time) and the default one; rt_stats.light_ms is reported in both."""
    if name == "two_dragons":
        sc = rr.two_dragons(mesh_faces=30000)
    elif name == "hdr":
        sc = rr.material_hdr(0, mesh_faces=12000)
    else:
        sc = rr.cornell_box_statue(mesh_faces=30000, variant=0)
    gs = gpu_ctx.upload(sc)
    for pool in (0, 4096, 70000):
        cfg = rr.make_cfg(160, 120, 16, seed=5, paths_in_flight=pool)
        monkeypatch.setenv("RT_LIGHT_OVERLAP", "0")
        monkeypatch.setenv("RT_CLS_STREAMS", "1")
        r0, n0, s0 = gpu_ctx.render(gs, sc.camera, cfg)
        monkeypatch.delenv("RT_LIGHT_OVERLAP")
        monkeypatch.delenv("RT_CLS_STREAMS")  # (default: on its side stream; class kernels alternate between two streams)
        r1, n1, s1 = gpu_ctx.render(gs, sc.camera, cfg)
        assert np.array_equal(r0, r1) and np.array_equal(n0, n1)
        assert (s0.paths, s0.rays_extension, s0.rays_shadow, s0.rays_probe, s0.vertices_shaded) == \
               (s1.paths, s1.rays_extension, s1.rays_shadow, s1.rays_probe, s1.vertices_shaded)
        if s1.shade_launches:
            assert s0.light_ms > 0.0 and s1.light_ms > 0.0
    gs.close()


def test_primitive_count_limit():
    L = F.lib()
    ctx = rr.Context(0)
    h = C.c_void_p()
    assert L.rt_scene_create(ctx._h, C.byref(h)) == 0
    one = (F.rt_primitive * 1)()
    assert L.rt_scene_set_primitives(h, one, 1 << 27) == F.RT_ERR_UNSUPPORTED
    assert b"2^27" in L.rt_last_error()
    L.rt_scene_destroy(h)
    ctx.close()


def _rmse(a, b, drop_top=0.0):
    d = ((a - b) ** 2).sum(axis=-1).reshape(-1)
    if drop_top > 0.0:
        k = int(np.ceil(d.size * drop_top))
        d = np.sort(d)[:d.size - k]
    return float(np.sqrt(d.mean() / 3.0))


@pytest.mark.parametrize("name", ["cornell_box", "cornell_statue_plastic", "dragon_metal", "two_dragons", "hdr_glass"])
def test_f32_fast_mode_is_reported_not_gated(gpu_ctx, name):
    """RT_PRECISION_F32: the same kernels in binary32.  Same sample counts and camera samples; paths diverge from the
    f64 ones after a few bounces, so the per-pixel difference at low spp is Monte-Carlo noise (SURVEY.md 8d expects
    1e-3..1e-2 and asks for it to be REPORTED with and without the top 0.01 % pixels).  What is asserted here is only
    that the fast mode is a faithful estimator of the same image: image means within 2 %, and a per-pixel RMSE that
    stays at the Monte-Carlo noise level of the sample count (a few 1e-3 at 256 spp)."""
    make = {"cornell_box": lambda: rr.cornell_box(),
            "cornell_statue_plastic": lambda: rr.cornell_box_statue(mesh_faces=8000, variant=3),
            "dragon_metal": lambda: rr.plastic_dragon(mesh_faces=8000, variant=1),
            "two_dragons": lambda: rr.two_dragons(mesh_faces=6000, variant=0),
            "hdr_glass": lambda: rr.material_hdr(3, mesh_faces=4000)}[name]
    sc = make()
    gs = gpu_ctx.upload(sc)
    out = {}
    for spp in (16, 256):
        r64, n64, s64 = gpu_ctx.render(gs, sc.camera, rr.make_cfg(64, 64, spp, seed=7))
        r32, n32, s32 = gpu_ctx.render(gs, sc.camera, rr.make_cfg(64, 64, spp, seed=7, precision=F.RT_PRECISION_F32))
        assert np.array_equal(n32, n64) and s32.paths == s64.paths
        i64, i32 = r64 / n64[..., None], r32 / n32[..., None]
        ok = np.isfinite(i64).all(axis=-1) & np.isfinite(i32).all(axis=-1)
        assert ok.mean() > 0.999
        i64, i32 = np.where(ok[..., None], i64, 0.0), np.where(ok[..., None], i32, 0.0)
        m64, m32 = np.clip(i64, 0, 10).mean(), np.clip(i32, 0, 10).mean()
        out[spp] = (_rmse(np.clip(i64, 0, 10), np.clip(i32, 0, 10)), _rmse(np.clip(i64, 0, 10), np.clip(i32, 0, 10), 1e-4), m64, m32)
        assert abs(s32.rays / s64.rays - 1.0) < 0.02  # the same amount of work
    print(f"\n[f32 vs f64] {name}: RMSE @16 spp {out[16][0]:.4g} (without top 0.01 %: {out[16][1]:.4g}), "
          f"@256 spp {out[256][0]:.4g} ({out[256][1]:.4g}); image mean f64 {out[256][2]:.5g} f32 {out[256][3]:.5g}")
    assert abs(out[256][3] / out[256][2] - 1.0) < 0.02
    assert out[256][0] < 5e-3
    gs.close()


@pytest.mark.parametrize("name", ["cornell_box", "cornell_box_spheres", "sphere_roughness", "statue", "two_dragons"])
def test_f32_traversal_finds_the_same_primitives(gpu_ctx, name):
    """Kernel-level check of the fast mode: rt_intersect_batch_ex(RT_INTERSECT_F32) runs the binary32 traversal.  On rays whose f64 hit is not within rounding distance of another primitive the two modes
    must name the same primitive; a handful of edge / silhouette rays may differ.  (This is the test that found a
    compiler problem in the binary32 instance: all yz rects were missed.)"""
    make = {"cornell_box": lambda: rr.cornell_box(), "cornell_box_spheres": lambda: rr.cornell_box_spheres(),
            "sphere_roughness": lambda: rr.sphere_roughness(),
            "statue": lambda: rr.cornell_box_statue(mesh_faces=20000, variant=0),
            "two_dragons": lambda: rr.two_dragons(mesh_faces=20000)}[name]
    sc = make()
    gs = gpu_ctx.upload(sc)
    rng = np.random.default_rng(11)
    n = 200000
    lo, hi = (5.0, 550.0) if name in ("cornell_box", "cornell_box_spheres", "statue") else (-9.0, 9.0)
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32).astype(np.float64)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1)[:, None]).astype(np.float32).astype(np.float64)
    t64, p64 = gpu_ctx.intersect_batch(gs, o, d, F.RT_SMALL)
    t32, p32 = gpu_ctx.intersect_batch(gs, o, d, F.RT_SMALL, flags=F.RT_INTERSECT_F32)
    tw, pw = gpu_ctx.intersect_batch(gs, o, d, F.RT_SMALL, flags=F.RT_INTERSECT_F32 | F.RT_INTERSECT_WAVEFRONT)
    gs.close()
    assert np.array_equal(pw, p32)  # the fast mode's k_trace == its run-to-completion loop
    assert (p64 >= 0).mean() > 0.3
    same = p64 == p32
    assert same.mean() > 0.999, (name, float(same.mean()))
    hit = same & (p64 >= 0)
    assert np.all(np.abs(t32[hit] - t64[hit]) <= 2e-4 * np.maximum(1.0, np.abs(t64[hit])))


@pytest.mark.parametrize("name", ["cornell_box", "cornell_box_spheres", "statue", "dragon", "two_dragons", "hdr"])
def test_render_traversal_kernel_on_caller_rays(gpu_ctx, name):
    """rt_intersect_batch_ex(RT_INTERSECT_WAVEFRONT): the caller's rays through k_trace itself -- persistent waves,
    queue reservations, refill below 24 idle lanes, while-while majority scheduling, results written in refill rounds
    -- bit for bit against the oracle (prim and t), for ray counts that leave partial reservations and partial waves
    and for both traversal instances (with / without spheres and transformed rects).  VERDICT r1: this path was only
    checked through whole-film equality."""
    from tests import oracle_ffi as O
    make, lo, hi = {"cornell_box": (lambda: rr.cornell_box(), 5.0, 550.0),
                    "cornell_box_spheres": (lambda: rr.cornell_box_spheres(), 5.0, 550.0),
                    "statue": (lambda: rr.cornell_box_statue(mesh_faces=30000, variant=0), 5.0, 550.0),
                    "dragon": (lambda: rr.plastic_dragon(mesh_faces=60000, variant=1), -8.0, 8.0),
                    "two_dragons": (lambda: rr.two_dragons(mesh_faces=20000), -8.0, 10.0),
                    "hdr": (lambda: rr.material_hdr(1, mesh_faces=5000), -3.0, 3.0)}[name]
    sc = make()
    osc = O.OracleScene(sc)
    gs = gpu_ctx.upload(sc)
    rng = np.random.default_rng(9)
    for n in (1, 63, 129, 5000, 300001):
        o = rng.uniform(lo, hi, size=(n, 3))
        d = rng.normal(size=(n, 3)) * rng.uniform(0.1, 20.0, size=(n, 1))
        if n > 1000:
            d[:200, 0] = 0.0
            d[200:300, :2] = 0.0
        tw, pw = gpu_ctx.intersect_batch(gs, o, d, F.RT_SMALL, flags=F.RT_INTERSECT_WAVEFRONT)
        to, po = osc.intersect_batch(o, d, F.RT_SMALL)
        assert np.array_equal(pw, po), f"{name} n={n}: {(pw != po).sum()} prim mismatches"
        assert np.array_equal(tw, to)
    # the device-built tree through the same kernel
    gd = gpu_ctx.upload(sc, device_build=True)
    o = rng.uniform(lo, hi, size=(40000, 3))
    d = rng.normal(size=(40000, 3))
    tw, pw = gpu_ctx.intersect_batch(gd, o, d, F.RT_SMALL, flags=F.RT_INTERSECT_WAVEFRONT)
    to, po = osc.intersect_batch(o, d, F.RT_SMALL)
    assert np.array_equal(pw, po) and np.array_equal(tw, to)
    with pytest.raises(rr.RtError):  # a free-form interval is not what the render kernel traces
        gpu_ctx.intersect_batch(gs, o[:4], d[:4], 0.0, flags=F.RT_INTERSECT_WAVEFRONT)
    gd.close()
    gs.close()
    osc.close()


def test_host_bvh_is_shared_through_the_node_local_cache(gpu_ctx, tmp_path, monkeypatch):
    """RT_BVH_CACHE=<dir> (abi.hip: build_bvh_shared): the first process to commit a scene publishes the host-built
    tree, later commits of the same primitives read it (bench.py --gpus N: eight ranks build once, not eight times).
    Same film either way; a damaged or foreign file is ignored and rebuilt."""
    import glob
    import os
    sc = rr.cornell_box_statue(mesh_faces=9000, variant=1)
    cfg = rr.make_cfg(48, 40, 4, seed=6)
    g0 = gpu_ctx.upload(sc)
    assert g0.info()["build_from_cache"] == 0
    base = gpu_ctx.render(g0, sc.camera, cfg)
    g0.close()
    monkeypatch.setenv("RT_BVH_CACHE", str(tmp_path))
    g1 = gpu_ctx.upload(sc)
    assert g1.info()["build_from_cache"] == 0
    files = glob.glob(str(tmp_path / "rtbvh_*.bin"))
    assert len(files) == 1 and os.path.getsize(files[0]) > 9000 * 4
    g2 = gpu_ctx.upload(sc)
    assert g2.info()["build_from_cache"] == 1 and g2.info()["n_bvh_nodes"] == g1.info()["n_bvh_nodes"]
    for g in (g1, g2):
        r, n, s = gpu_ctx.render(g, sc.camera, cfg)
        assert np.array_equal(r, base[0]) and np.array_equal(n, base[1]) and s.rays == base[2].rays
        g.close()
    # another scene gets another key; a truncated file is rejected
    sc2 = rr.cornell_box_statue(mesh_faces=9001, variant=1)
    g3 = gpu_ctx.upload(sc2)
    assert g3.info()["build_from_cache"] == 0 and len(glob.glob(str(tmp_path / "rtbvh_*.bin"))) == 2
    g3.close()
    with open(files[0], "r+b") as fh:
        fh.truncate(os.path.getsize(files[0]) // 2)
    g4 = gpu_ctx.upload(sc)
    assert g4.info()["build_from_cache"] == 0
    r, n, _ = gpu_ctx.render(g4, sc.camera, cfg)
    assert np.array_equal(r, base[0])
    g4.close()
    # ... and so is one with a flipped byte in the node array (payload checksum); the rebuild republishes a good file
    size = os.path.getsize(files[0])
    with open(files[0], "r+b") as fh:
        fh.seek(size // 3)
        b = fh.read(1)
        fh.seek(size // 3)
        fh.write(bytes([b[0] ^ 0x40]))
    g4b = gpu_ctx.upload(sc)
    assert g4b.info()["build_from_cache"] == 0
    g4b.close()
    g4c = gpu_ctx.upload(sc)
    assert g4c.info()["build_from_cache"] == 1
    r, n, _ = gpu_ctx.render(g4c, sc.camera, cfg)
    assert np.array_equal(r, base[0])
    g4c.close()
    # the device builder does not use it
    g5 = gpu_ctx.upload(sc, device_build=True)
    assert g5.info()["build_from_cache"] == 0
    g5.close()


def test_plain_c_host_renders_the_same_film(gpu_ctx, tmp_path):
    """examples/gpu_tile.c -- a host in C99 against the C ABI alone, the stand-in for the reference's Rust binding
     -- renders the same film as the ctypes binding: byte-for-byte (FNV-1a of the film), and writes
    the PNG of util::draw_picture."""
    import subprocess
    from tests.test_host_cpu import _build_c_example
    exe = _build_c_example(tmp_path)
    W, H, spp = 64, 48, 4
    png = str(tmp_path / "c_host.png")
    r = subprocess.run([exe, "cornell_box_statue", str(W), str(H), str(spp), png], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    fields = r.stdout.split()
    got = dict(zip(fields[2::2], fields[3::2]))

    def fnv(buf):
        h = 0xcbf29ce484222325
        for b in bytes(buf):
            h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
        return "%016x" % h

    sc = rr.Scene("cornell_box_statue", W / H, 20000, None, 0)
    gs = gpu_ctx.upload(sc)
    rgb, n, st = gpu_ctx.render(gs, sc.camera, rr.make_cfg(W, H, spp, seed=0))
    gs.close()
    assert got["film_fnv"] == fnv(rgb.tobytes()) and got["count_fnv"] == fnv(n.tobytes())
    assert int(got["rays"]) == st.rays
    from tests.test_host_cpu import _read_png
    assert np.array_equal(_read_png(png), gpu_ctx.resolve_rgb8(rgb, n))


@pytest.mark.gpu
def test_default_pool_falls_back_when_memory_is_short(gpu_ctx):
    """rt_render_cfg.paths_in_flight = 0 asks for a whole batch of path state (up to 2^28 slots, 140 GB): when that does not
    fit, the library halves the DEFAULT until it does and renders the same film; a size the caller asked for is never
    changed -- it fails with RT_ERR_OOM.  The shortage is simulated (abi.hip: RT_TEST_POOL_OOM_ABOVE, read once per
    process, hence the child processes)."""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import hashlib, sys
sys.path.insert(0, %r)
import numpy as np
import rustraytracer_amd as rr
sc = rr.cornell_box_statue(mesh_faces=3000, variant=1)
ctx = rr.Context(0)
gs = ctx.upload(sc)
pif = int(sys.argv[1])
try:
    r, n, st = ctx.render(gs, sc.camera, rr.make_cfg(192, 160, 16, seed=5, paths_in_flight=pif))
    print("film", hashlib.sha256(r.tobytes() + n.tobytes()).hexdigest(), st.rays)
except Exception as e:
    print("error", type(e).__name__, str(e)[:200])
''' % root
    sc = rr.cornell_box_statue(mesh_faces=3000, variant=1)
    gs = gpu_ctx.upload(sc)
    r, n, st = gpu_ctx.render(gs, sc.camera, rr.make_cfg(192, 160, 16, seed=5))
    gs.close()
    want = "film %s %d" % (hashlib.sha256(r.tobytes() + n.tobytes()).hexdigest(), st.rays)

    def child(pif):
        # (the hook exists in librt_amd_testhooks.so only -- csrc/Makefile `testhooks`; the product library ignores the variable)
        hooks = os.path.join(root, "rustraytracer_amd", "librt_amd_testhooks.so")
        assert os.path.exists(hooks), "build it with: make -C rustraytracer_amd/csrc testhooks"
        env = dict(os.environ, RT_TEST_POOL_OOM_ABOVE="15", RT_AMD_LIB=hooks)  # pools above 32768 paths "do not fit"
        out = subprocess.run([sys.executable, "-c", code, str(pif)], env=env, capture_output=True, text=True, timeout=300)
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith(("film", "error"))]
        assert lines, out.stderr[-800:]
        return lines[-1]

    # 192 x 160 x 16 = 491520 camera samples: the default pool (a whole batch) is halved four times down to 30720 paths
    assert child(0) == want
    got = child(491520)
    assert got.startswith("error") and "memory" in got.lower(), got

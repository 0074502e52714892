#!/usr/bin/env python3
"""bench.py -- Mrays/s of the path-tracing hot path on N MI355X (one process per GPU).

A step = one full render of the workload through the C ABI (rt_render_device): scene, BVH and
film buffers already resident in HBM, film left on the device, then (N > 1) the RCCL gather of
the per-rank films to rank 0.  Rays = R1 + R2 + R3 root closest-hit queries,
counted by the device and identical to the CPU oracle's counters.

Workload (every N): BASELINE.json's metric config, configs[3] = C4 -- two_dragons() with both
dragons (glass + metal, 2 x procedural P-871k in place of the missing dragon.obj), 1920x1080 @
1024 spp, max_depth 25, seed 0.  It fits one GPU (the scene is 0.5 GB; path state + film staging
~40 GB of the 288 GB).
N > 1: STRONG scaling -- the same image, its 16x16 tiles interleaved over the ranks (tile (tx, ty) -> rank
rt_tile_owner(tx, ty, N), rt_render_cfg.tile_rank / tile_world), every rank renders only its tiles and the own-tile
pixels are packed and gathered to rank 0 over RCCL/xGMI (rustraytracer_amd/dist.py); the gather
time is reported separately (`gather_ms_per_step`) and is inside the timed region.

One JSON line on rank 0 with
  roofline      per kernel group (k_trace / k_shade = the class kernels + the light kernel / k_classify): share of the
                device time (HIP events), algorithmic bytes per step over that time, and -- from the rocprofv3 PMC
                passes committed under profiles/, when they belong to the library running now -- the measured HBM
                fractions (all bytes, reads only).  The top level names the group with the LARGEST share of device
                time (`kernel`); `frac` is its measured fraction when the profile is current, else the algorithmic one;
                an algorithmic fraction above 1 (cache hits counted) is never reported as `frac`
  rays          R1 / R2 / R3 of SURVEY.md 8(d) (extension incl. primary, shadow, MIS probe)
  cpu_baseline  the oracle in reference-shaped mode on the host cores, bounded sample (rank 0, N = 1 only)
  extra         (N = 1) the smaller single-GPU configs C2 and C3 measured the same way, for continuity with
                round 1's numbers
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)

WORKLOADS = {
    # name: (preset, kwargs, W, H, spp, description)
    "c2": ("cornell_box_statue", dict(mesh_faces=400000, variant=0), 512, 512, 64,
           "C2 cornell_box_statue (matte) + procedural P-400k mesh, 512x512 @ 64 spp, max_depth 25"),
    "c3": ("plastic_dragon", dict(mesh_faces=871414, variant=1), 1024, 1024, 256,
           "C3 dragon (procedural P-871k) microfacet metal, 1024x1024 @ 256 spp, max_depth 25"),
    "c3p": ("plastic_dragon", dict(mesh_faces=871414, variant=0), 1024, 1024, 256,
            "plastic_dragon() as committed in scenes.rs:310-375 (plastic, two lobes), procedural P-871k, 1024x1024 @ 256 spp"),
    "c4": ("two_dragons", dict(mesh_faces=871414, variant=0), 1920, 1080, 1024,
           "C4 two_dragons (glass + metal, 2 x procedural P-871k), 1920x1080 @ 1024 spp, max_depth 25"),
    "c5": ("plastic_dragon", dict(mesh_faces=871414, variant=2), 2048, 2048, 4096,
           "C5 dragon (procedural P-871k) smooth glass, 2048x2048 @ 4096 spp, max_depth 25"),
    "hdr": ("material_hdr", dict(variant=3, mesh_faces=150000), 512, 512, 64,
            "row f4: material_hdr(3) rough glass; the reference's Mesh000/001.obj + envmap.hdr (--assets), P-150k for the "
            "missing Mesh002.obj, 512x512 @ 64 spp"),
    "hdr1": ("material_hdr", dict(variant=1, mesh_faces=150000), 512, 512, 64,
             "row f4: material_hdr(1) rose-gold metal; the reference's Mesh000/001.obj + envmap.hdr (--assets), P-150k for "
             "the missing Mesh002.obj, 512x512 @ 64 spp"),
    "teapot": ("teapot_hdr", dict(mesh_faces=200000), 1280, 720, 64,
               "row f4: teapot_hdr() (scenes.rs:744-808) smooth plastic, procedural lid + body (the reference's teapot meshes are "
               "not in its checkout) under the reference's envmap.hdr, 1280x720 @ 64 spp"),
    "c1": ("cornell_box", dict(), 256, 256, 16, "C1 cornell_box 256x256 @ 16 spp"),
    "tiny": ("cornell_box_statue", dict(mesh_faces=20000, variant=0), 128, 128, 8, "tiny smoke workload"),
}

# Algorithmic bytes of the shading kernels (round-4 layout):
#   per path entering a bounce (= extension ray, R1): its 8-B list entry; line 0 of its record (128 B) unless it is a
#     camera sample, which has no record yet -- in the class kernel of its vertex, or in the light kernel when the ray
#     found nothing (light_bytes);
#   per shaded vertex: line 0 of the survivor's record written (128 B), its ray written to the ray arrays (48 B), the
#     winner's scene records (72 B vertices + 72 B normals + 8 B meta + 120 B primitive or material record: SURVEY.md
#     8d's B_shade doubled for f64 = 272 B);
#   per vertex with a pending MIS probe (R3): line 1 (A, Q, K) written and read back (2 x 128 B); per vertex whose only
#     pending term is a light sample (R2 - R3): its ready-made contribution, 24 B written and read back; and the shadow
#     target / probe direction (24 B per ray).
SHADE_BYTES_PER_VERTEX = 128 + 48 + 272
CLASSIFY_BYTES_PER_RAY = 2 * 8 + 8  # queue entry + hit word read by the count and by the scatter pass, one 8-B list entry


def shade_bytes(st):
    """The class kernels (vertices) -- the paths that end without a vertex are light_bytes()."""
    line1 = st.rays_probe
    lone = max(st.rays_shadow - st.rays_probe, 0)
    hits = min(st.vertices_shaded, st.rays_extension)
    stored = max(st.rays_extension - st.paths, 0) * hits // max(st.rays_extension, 1)  # (hits that are not camera samples)
    return (8 * hits + 128 * stored + SHADE_BYTES_PER_VERTEX * st.vertices_shaded +
            256 * line1 + 48 * lone + 24 * (st.rays_shadow + st.rays_probe))


def light_bytes(st):
    """The light kernel: per extension ray that found nothing its 8-B list entry, line 0 of its record (128 B) unless it
    is a camera sample, and the 24 B of film staging it retires into."""
    esc = max(st.rays_extension - st.vertices_shaded, 0)
    stored = max(st.rays_extension - st.paths, 0) * esc // max(st.rays_extension, 1)
    return (8 + 24) * esc + 128 * stored


def pick_roofline(kernels):
    """The top level of the `roofline` object from its per-kernel entries (pure: tests/test_host_cpu.py checks it).
    kernel = the entry with the largest share of device time.  frac = its measured HBM fraction when there is one
    (hbm_frac_rocprof), else its algorithmic fraction -- unless that exceeds 1 (requested bytes incl. cache hits cannot
    be a fraction of the HBM peak), in which case frac is null and the number is kept as frac_algorithmic_incl_cache_hits."""
    name = max(kernels, key=lambda k: kernels[k].get("share_of_device_time") or 0.0)
    k = kernels[name]
    alg = k.get("frac_algorithmic")
    meas = k.get("hbm_frac_rocprof")
    out = {"kernel": name, "achieved": k.get("achieved"), "share_of_device_time": k.get("share_of_device_time"),
           "frac_algorithmic_incl_cache_hits": alg, "hbm_frac_rocprof": meas, "hbm_read_frac_rocprof": k.get("hbm_read_frac_rocprof")}
    if meas is not None:
        out["frac"], out["frac_is"] = meas, "measured (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE of this kernel group / its time / peak)"
    elif alg is not None and alg <= 1.0:
        out["frac"], out["frac_is"] = alg, "algorithmic (no current counter profile)"
    else:
        out["frac"], out["frac_is"] = None, "withheld: no current counter profile and the algorithmic fraction exceeds 1"
    return out


def trace_bytes(st, info):
    """Bytes the traversal kernel k_trace requests (cache hits included), f64 parity layout:
    128 B per BVH4 node fetched, 76 B per triangle tested (72 B vertices + 4 B id), 52 B per sphere/rect
    tested (48 B parameters in the leaf slot + 4 B id), 56 B per ray (24 B origin + 24 B dir/target + 4 B
    queue entry + 4 B result).  The rays the fused tail launch traced are not k_trace's: subtracted."""
    rays = st.rays - st.tail_rays
    return (info["node_bytes"] * (st.nodes_fetched - st.tail_nodes_fetched) +
            info["tri_bytes"] * (st.tris_tested - st.tail_tris_tested) +
            info["other_bytes"] * (st.others_tested - st.tail_others_tested) + 56 * rays), rays


def cpu_baseline(scene, W, H, target_s=9.0):
    """Oracle, reference-shaped (exhaustive traversal, 6 threads = consts.rs:8 NUM_THREADS), on a
    bounded sample of the same workload: the full image at a reduced spp chosen for ~target_s."""
    import rustraytracer_amd as rr
    from tests import oracle_ffi as O
    osc = O.OracleScene(scene)
    threads = 6
    share = min(os.cpu_count() or 1, 16)  # the GPU box gives 16 host cores per GPU
    t0 = time.time()
    _, _, st = osc.render(scene.camera, rr.make_cfg(W, H, 1, seed=0), O.EXHAUSTIVE, threads)
    dt = max(time.time() - t0, 1e-3)
    spp = 1
    while spp * 2 * dt <= target_s and spp < 64:
        spp *= 2
    if spp > 1:
        t0 = time.time()
        _, _, st = osc.render(scene.camera, rr.make_cfg(W, H, spp, seed=0), O.EXHAUSTIVE, threads)
        dt = time.time() - t0
    rays = st.rays_extension + st.rays_shadow + st.rays_probe
    out = {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": f"the workload's full {W}x{H} image at {spp} spp (throughput is linear in spp), oracle in "
                     f"reference-shaped mode (exhaustive BVH traversal, hittable.rs:591-634, 16x16 tile workers), "
                     f"{rays} rays in {dt:.1f} s",
           "host_cores_share": share}
    # the same sample on the whole CPU share, reference-shaped and with the oracle's pruned traversal
    t0 = time.time()
    _, _, st2 = osc.render(scene.camera, rr.make_cfg(W, H, spp, seed=0), O.EXHAUSTIVE, share)
    dt2 = time.time() - t0
    out["reference_shaped_all_cores"] = {"value": st2.rays / dt2 / 1e6, "cores": share}
    t0 = time.time()
    _, _, st3 = osc.render(scene.camera, rr.make_cfg(W, H, spp, seed=0), O.ORDERED, share)
    dt3 = time.time() - t0
    out["ordered_all_cores"] = {"value": st3.rays / dt3 / 1e6, "cores": share}
    osc.close()
    return out


_ASSET_DIRS = {}


def asset_dir(spec, preset="material_hdr"):
    """Where the row-f4 presets (material_hdr, teapot_hdr) find the reference's data files.  "golden" (default): the
    committed fixtures tests/golden/assets/material (Mesh000/001.obj gzip-compressed, envmap.hdr), unpacked once into
    a temporary directory (removed at exit) -- they travel to the GPU box with the repo; "none": procedural stand-ins for
    everything; any other value: a directory laid out like the reference's data/material (the build container has
    /root/reference/data/material).
    teapot_hdr() gets a directory that holds the environment map ONLY: its own two meshes are not in the reference's
    checkout, and a directory with models/Mesh000.obj + Mesh001.obj in it would make the preset load the material-preview
    meshes in place of the procedural lid and body (the round-3 "teapot" row measured exactly that)."""
    if spec == "none":
        return None
    if spec != "golden":
        return spec
    key = "env_only" if preset == "teapot_hdr" else "full"
    if key not in _ASSET_DIRS:
        import atexit
        import gzip
        import shutil
        import tempfile
        src = os.path.join(ROOT, "tests", "golden", "assets", "material")
        out = tempfile.mkdtemp(prefix="rr_assets_")
        atexit.register(shutil.rmtree, out, ignore_errors=True)
        os.makedirs(os.path.join(out, "textures"))
        if key == "full":
            os.makedirs(os.path.join(out, "models"))
            for name in ("Mesh000.obj", "Mesh001.obj"):
                with gzip.open(os.path.join(src, "models", name + ".gz"), "rb") as fi, open(os.path.join(out, "models", name), "wb") as fo:
                    shutil.copyfileobj(fi, fo)
        shutil.copy(os.path.join(src, "textures", "envmap.hdr"), os.path.join(out, "textures", "envmap.hdr"))
        _ASSET_DIRS[key] = out
    return _ASSET_DIRS[key]


ASSETS = "golden"


class Bench:
    """One workload on this rank's GPU: scene resident, film on the device."""

    def __init__(self, name, rank, world, local_rank, coll_dev, paths_in_flight=0, precision=0, ctx=None):
        import torch

        import rustraytracer_amd as rr
        from rustraytracer_amd import dist as rd
        self.torch, self.rr = torch, rr
        self.name, self.rank, self.world, self.coll_dev = name, rank, world, coll_dev
        preset, kw, self.W, self.H, self.spp, self.desc = WORKLOADS[name]
        if preset in ("material_hdr", "teapot_hdr"):
            kw = dict(kw, mesh_path=asset_dir(ASSETS, preset))
        self.scene = rr.Scene(preset, self.W / self.H, **kw)
        # (the extra rows run on the headline bench's context: one pool of path state per GPU -- 238 GB at the default size)
        self.own_ctx = ctx is None
        self.ctx = rr.Context(local_rank) if ctx is None else ctx
        self.gs = self.ctx.upload(self.scene)
        self.info = self.gs.info()
        if preset == "teapot_hdr" and ASSETS == "golden":
            # the procedural body (mesh_faces) + lid (an eighth of it), not whatever meshes the asset directory happens to hold
            assert self.info["n_triangles"] == kw["mesh_faces"] + kw["mesh_faces"] // 8, (self.info["n_triangles"], kw["mesh_faces"])
        self.d_rgb = torch.zeros((self.H, self.W, 3), dtype=torch.float64, device="cuda")
        self.d_n = torch.zeros((self.H, self.W), dtype=torch.int32, device="cuda")
        self.pif = paths_in_flight
        self.precision = precision
        self.cfg = rr.make_cfg(self.W, self.H, self.spp, seed=0, tile_rank=rank, tile_world=world,
                               paths_in_flight=paths_in_flight, precision=precision)
        self.gather = rd.FilmGather(self.W, self.H, coll_dev) if world > 1 else None
        self.gather_s = 0.0

    def step(self):
        torch = self.torch
        # on torch's current stream (NULL = the null stream): the film zeroing of the next step is ordered after
        # this step's gather kernels, and the gather sees the finished film
        st = self.ctx.render_device(self.gs, self.scene.camera, self.cfg, self.d_rgb.data_ptr(), self.d_n.data_ptr(),
                                    stream=torch.cuda.current_stream().cuda_stream)
        if self.gather is not None:
            t0 = time.perf_counter()
            # the path's only exchange step: each rank's own tiles go straight to rank 0 (RCCL over xGMI)
            if self.coll_dev == "cuda":
                self.gather.gather(self.d_rgb, self.d_n)
                torch.cuda.synchronize()
            else:  # gloo dry run: stage through the host
                h_rgb, h_n = self.d_rgb.cpu(), self.d_n.cpu()
                self.gather.gather(h_rgb, h_n)
                if self.rank == 0:
                    self.d_rgb.copy_(h_rgb)
                    self.d_n.copy_(h_n)
            self.gather_s += time.perf_counter() - t0
        return st

    def counted(self):
        """Instrumented pass (outside any timed region): the same kernels and BVH with traversal counters."""
        cfg_c = self.rr.make_cfg(self.W, self.H, self.spp, seed=0, tile_rank=self.rank, tile_world=self.world,
                                 paths_in_flight=self.pif, count_traversal=True, precision=self.precision)
        st = self.ctx.render_device(self.gs, self.scene.camera, cfg_c, self.d_rgb.data_ptr(), self.d_n.data_ptr())
        self.torch.cuda.synchronize()
        return st

    def close(self):
        self.gs.close()
        if self.own_ctx:
            self.ctx.close()


def fnv1a_film(d_rgb, d_n):
    """The device film as one number: FNV-1a (64 bit) over the SHA-256 digests of its 1-MiB blocks (sums as f64, then
    counts as i32) -- byte-by-byte FNV over 70 MB is too slow in Python."""
    import hashlib
    h = 0xcbf29ce484222325
    data = d_rgb.cpu().numpy().tobytes() + d_n.cpu().numpy().tobytes()
    for off in range(0, len(data), 1 << 20):
        for byte in hashlib.sha256(data[off:off + (1 << 20)]).digest():
            h = ((h ^ byte) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def timed(b, steps, warmup, barrier):
    if warmup == 0 and b.world > 1:
        # one-off set-up that is not the measured work: RCCL connects peers at the first gather (1 spp of the same
        # image, untimed).  With --warmup >= 1 the warm-up step does it; single-GPU runs (the profiles) never need it.
        cfg = b.cfg
        b.cfg = b.rr.make_cfg(b.W, b.H, 1, seed=0, tile_rank=b.rank, tile_world=b.world, paths_in_flight=b.pif,
                              precision=b.precision)
        b.step()
        b.cfg = cfg
    for _ in range(warmup):
        b.step()
    barrier()
    b.gather_s = 0.0
    acc = {"rays": 0, "trace_ms": 0.0, "kernel_ms": 0.0, "launches": 0, "shade_ms": 0.0, "shade_launches": 0,
           "vertices": 0, "classify_ms": 0.0, "light_ms": 0.0, "r1": 0, "r2": 0, "r3": 0}
    t0 = time.perf_counter()
    for _ in range(steps):
        st = b.step()
        acc["rays"] += st.rays
        acc["trace_ms"] += st.trace_ms
        acc["kernel_ms"] += st.kernel_ms
        acc["launches"] += st.trace_launches
        acc["shade_ms"] += st.shade_ms
        acc["shade_launches"] += st.shade_launches
        acc["classify_ms"] += st.classify_ms
        acc["light_ms"] += st.light_ms
        acc["r1"] += st.rays_extension
        acc["r2"] += st.rays_shadow
        acc["r3"] += st.rays_probe
        acc["vertices"] += st.vertices_shaded
    barrier()
    acc["dt"] = time.perf_counter() - t0
    return acc


def pmc_groups(rec):
    """Per kernel group, the HBM bytes per step of a committed counter profile (profiles/trace_pmc_<workload>.json:
    per_kernel_hbm_read_bytes_per_step / ..write..): k_trace = the timed instance, k_shade = the class kernels,
    k_shade_light = the light kernel, k_classify = the three launches of the counting sort."""
    rd, wr = rec.get("per_kernel_hbm_read_bytes_per_step") or {}, rec.get("per_kernel_hbm_write_bytes_per_step") or {}
    out = {}
    for grp, needles in (("k_trace", ("k_trace<false",)), ("k_shade", ("k_shade_cls",)), ("k_shade_light", ("k_shade_light",)),
                         ("k_classify", ("k_classify_",))):
        r = sum(v for k, v in rd.items() if any(n in k for n in needles))
        w = sum(v for k, v in wr.items() if any(n in k for n in needles))
        if r or w:
            out[grp] = (r, w)
    return out


def roofline(b, acc, stc, steps):
    """Per kernel group: algorithmic bytes per step over its HIP-event time, and the measured HBM fractions of the
    committed counter profile when it was taken from the library that is running; top level: pick_roofline()."""
    steps = max(steps, 1)
    alg, trace_rays = trace_bytes(stc, b.info)  # per render, this rank
    launches_per_step = acc["launches"] / steps
    avg_launch_s = (acc["trace_ms"] / 1e3) / max(acc["launches"], 1)
    dev_ms = max(acc["kernel_ms"], 1e-9)
    groups = {
        "k_trace": {"ms_per_step": acc["trace_ms"] / steps, "algorithmic_bytes_per_step": alg},
        "k_shade": {"ms_per_step": acc["shade_ms"] / steps, "algorithmic_bytes_per_step": shade_bytes(stc),
                    "kernels_per_bounce": b.info.get("n_classes"),
                    "note": "one kernel per vertex class of the scene; f64 VALU issue and latency share the bound with "
                            "bytes (DESIGN.md section 4)"},
        "k_classify": {"ms_per_step": acc["classify_ms"] / steps, "algorithmic_bytes_per_step": CLASSIFY_BYTES_PER_RAY * stc.rays},
        # the kernel for escaped / fold-only paths runs on a side stream BESIDE the class kernels and the next k_trace launch:
        # its time (HIP events on that stream) overlaps theirs, and its rate is what it gets next to them, not alone
        "k_shade_light": {"ms_per_step": acc["light_ms"] / steps, "algorithmic_bytes_per_step": light_bytes(stc), "overlapped": True},
    }
    for g in groups.values():
        s = g["ms_per_step"] / 1e3
        g["bound"], g["peak"], g["unit"] = "hbm", HBM_PEAK_GBS, "GB/s"
        g["achieved"] = g["algorithmic_bytes_per_step"] / s / 1e9 if s > 0 else 0.0
        g["frac_algorithmic"] = g["achieved"] / HBM_PEAK_GBS
        g["share_of_device_time"] = None if g.get("overlapped") else g["ms_per_step"] * steps / dev_ms
        g["hbm_frac_rocprof"] = g["hbm_read_frac_rocprof"] = g["traffic_per_step"] = None
    # Counter figures come from a committed rocprofv3 --pmc profile (counters cannot be read inside this run); the
    # profile names the kernel build it was taken from, and the figures are WITHHELD (null, with the reason in
    # traffic_source.stale) when the machine code of the library running now hashes differently.
    traffic = frac_step = None
    source = None
    pmc = os.path.join(ROOT, "profiles", f"trace_pmc_{b.name}.json")
    if b.world == 1 and b.precision == 0 and os.path.exists(pmc):
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from kernel_hash import kernel_hash
            rec = json.load(open(pmc))
            now = kernel_hash("k_")
            prof_hash = rec.get("library_kernels_hash") and rec["library_kernels_hash"].get("sha256")
            source = {"file": os.path.relpath(pmc, ROOT), "commit": rec.get("commit"), "library_kernels_hash": prof_hash,
                      "running_library_kernels_hash": now and now["sha256"], "stale": None}
            if prof_hash is None or now is None or now["sha256"] != prof_hash:
                source["stale"] = "the kernels were rebuilt since the profile was taken: counter figures withheld"
            else:
                # bytes per LAUNCH only carry over when a step is cut into the same launches (pool size, batch size)
                prof_lps = float(rec.get("launches") or 0) / max(float(rec.get("steps_profiled") or 1), 1.0)
                source["profile_launches_per_step"] = prof_lps
                if prof_lps and abs(prof_lps - launches_per_step) > 0.05 * prof_lps:
                    source["stale"] = ("the profile was taken with %.0f k_trace launches per step, this run has %.0f "
                                       "(pool / batch size): counter figures withheld" % (prof_lps, launches_per_step))
            if source["stale"] is None:
                for name, (r, w) in pmc_groups(rec).items():
                    s = groups[name]["ms_per_step"] / 1e3
                    if s > 0:
                        groups[name]["traffic_per_step"] = r + w
                        groups[name]["hbm_frac_rocprof"] = (r + w) / s / 1e9 / HBM_PEAK_GBS
                        groups[name]["hbm_read_frac_rocprof"] = r / s / 1e9 / HBM_PEAK_GBS
                if groups["k_trace"]["traffic_per_step"]:
                    traffic = groups["k_trace"]["traffic_per_step"] / max(launches_per_step, 1)
                all_k = rec.get("hbm_bytes_per_step_all_kernels")
                if all_k:  # all kernels of a step: HBM bytes per step / ms_per_step
                    frac_step = float(all_k) / (acc["dt"] / steps) / 1e9 / HBM_PEAK_GBS
                    source["hbm_bytes_per_step_all_kernels"] = float(all_k)
        except Exception as e:  # a malformed profile must not take the bench line down
            traffic = frac_step = None
            source = {"file": os.path.relpath(pmc, ROOT), "stale": f"unreadable: {e}"}
    top = pick_roofline(groups)
    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    out.update(top)
    out.update({
        # (the contract's `traffic`: HBM bytes per launch of the traversal kernel, whose launches are what a step is cut into)
        "traffic": traffic, "hbm_frac_rocprof_step": frac_step, "traffic_source": source,
        "kernels": groups,
        "k_trace_detail": {"avg_launch_ms": avg_launch_s * 1e3, "launches_per_step": launches_per_step,
                           "algorithmic_bytes_per_launch": alg / max(launches_per_step, 1),
                           "bytes_per_ray": alg / max(trace_rays, 1),
                           "nodes_per_ray": (stc.nodes_fetched - stc.tail_nodes_fetched) / max(trace_rays, 1),
                           "tris_per_ray": (stc.tris_tested - stc.tail_tris_tested) / max(trace_rays, 1),
                           "rays_in_fused_tail_launch": stc.tail_rays},
    })
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c4", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the C2 / C3 rows")
    ap.add_argument("--paths-in-flight", type=int, default=0)
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"],
                    help="f32 = the fast mode (RT_PRECISION_F32): reported, not the headline -- the metric is defined on the f64 parity mode")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: dry run of the N>1 control flow with several ranks sharing one GPU (films gathered on the host)")
    ap.add_argument("--assets", default="golden",
                    help='row-f4 workloads (hdr, hdr1, teapot): "golden" = the committed reference data files under tests/golden/assets, '
                         '"none" = procedural stand-ins, or a directory laid out like the reference\'s data/material')
    args = ap.parse_args()
    global ASSETS
    ASSETS = args.assets

    import torch
    import torch.distributed as dist

    import rustraytracer_amd as rr

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback exists)")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from rustraytracer_amd import dist as rd_
        # (bounded collective timeout: a rank that fails takes the whole job down within RT_DIST_TIMEOUT_S, default 600 s)
        if args.backend == "nccl":
            rd_.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            rd_.init_process_group("gloo")
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"
    # N ranks commit the same scene: the host BVH is built once per node and shared through /dev/shm (abi.hip:
    # build_bvh_shared) instead of N simultaneous 16-thread SAH builds; rank 0 removes the files at the end
    cache_dir = None
    if world > 1 and "RT_BVH_CACHE" not in os.environ:
        cache_dir = f"/dev/shm/rt_bvh_cache_{os.getuid()}_{os.environ.get('MASTER_PORT', '0')}"
        os.makedirs(cache_dir, exist_ok=True)
        os.environ["RT_BVH_CACHE"] = cache_dir

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    prec = 1 if args.precision == "f32" else 0
    b = Bench(args.workload, rank, world, local_rank, coll_dev, args.paths_in_flight, prec)
    acc = timed(b, args.steps, args.warmup, barrier)
    # max over ranks of the elapsed time, sum of rays
    tt = torch.tensor([acc["dt"], b.gather_s], dtype=torch.float64, device=coll_dev)
    rr_ = torch.tensor([float(acc["rays"])], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rr_, op=dist.ReduceOp.SUM)
    dt_max, gather_max, rays_all = float(tt[0].item()), float(tt[1].item()), float(rr_.item())
    devices_all = None
    if world > 1:  # which physical device every rank used (rank 0 reports the list)
        dv = torch.zeros(world, dtype=torch.int64, device=coll_dev)
        dv[rank] = local_rank
        dist.all_reduce(dv, op=dist.ReduceOp.SUM)
        devices_all = [int(x) for x in dv.tolist()]
    r3 = torch.tensor([float(acc["r1"]), float(acc["r2"]), float(acc["r3"])], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(r3, op=dist.ReduceOp.SUM)
    rays3 = [float(x) for x in r3.tolist()]
    # the film of the last timed step on rank 0 (N > 1: gathered), as one number
    film_hash = fnv1a_film(b.d_rgb, b.d_n) if rank == 0 else None
    film_ok, per_rank_ms = None, None
    if world > 1:
        # device time of every rank's own share, and the proof that the gathered film is THE film: the same image at
        # 1 spp rendered once by all ranks (tiles + gather) and once by rank 0 alone, compared bit for bit -- outside
        # the timed region
        pr = torch.zeros(world, dtype=torch.float64, device=coll_dev)
        pr[rank] = acc["kernel_ms"] / args.steps
        dist.all_reduce(pr, op=dist.ReduceOp.SUM)
        per_rank_ms = [float(x) for x in pr.tolist()]
        cfg_keep = b.cfg
        b.cfg = rr.make_cfg(b.W, b.H, 1, seed=0, tile_rank=rank, tile_world=world, paths_in_flight=b.pif, precision=b.precision)
        b.step()
        if rank == 0:
            many = (b.d_rgb.clone(), b.d_n.clone())
            b.cfg = rr.make_cfg(b.W, b.H, 1, seed=0, paths_in_flight=b.pif, precision=b.precision)
            g = b.gather
            b.gather = None
            b.step()
            b.gather = g
            film_ok = bool(torch.equal(many[0], b.d_rgb) and torch.equal(many[1], b.d_n))
        b.cfg = cfg_keep
        barrier()
    stc = b.counted()

    if rank == 0:
        W, H, spp = b.W, b.H, b.spp
        out = {
            "metric": "Mrays/sec (primary+secondary)", "value": rays_all / dt_max / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": b.desc + (f"; strong scaling: the same image, 16x16 tiles interleaved over {world} "
                                              "ranks, own tiles gathered to rank 0 over RCCL" if world > 1 else ""),
                       "width": W, "height": H, "spp": spp, "max_depth": rr.MAX_DEPTH, "seed": 0,
                       "triangles": b.info["n_triangles"], "bvh_nodes": b.info["n_bvh_nodes"],
                       "rays_per_step": rays_all / args.steps, "paths_per_step": W * H * spp,
                       "parallelism": f"tiles{world}", "bvh_from_shared_cache": bool(b.info.get("build_from_cache", 0)),
                       # path-state slots kept alive per GPU (about 0.8 KB each): the library default is a whole batch, <= 2^28
                       "paths_in_flight": args.paths_in_flight or "library default: min(batch, 2^28 paths) = 238 GB of path state at most, less when the device has less free",
                       "scene_commit_ms": b.info.get("build_ms")},
            "roofline": roofline(b, acc, stc, args.steps),
            "device_ms_per_step": acc["kernel_ms"] / args.steps,
            "gather_ms_per_step": gather_max / args.steps * 1e3 if world > 1 else 0.0,
            # what the collective layer actually saw (N > 1): backend as torch reports it ("nccl" = RCCL on ROCm) and the
            # number of ranks in the group, plus the devices the ranks rendered on
            "rccl_ranks": ({"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                            "devices": devices_all} if world > 1 else None),
            # also paths/s and the mean path length (rays per camera sample), whole job
            "mpaths_per_s": W * H * spp * args.steps / dt_max / 1e6,
            "mean_rays_per_path": rays_all / args.steps / (W * H * spp),
            # R1 extension incl. primary (integrator.rs:388), R2 shadow (:584), R3 MIS probe (:615), per step
            "rays": {"extension": rays3[0] / args.steps, "shadow": rays3[1] / args.steps, "probe": rays3[2] / args.steps},
            "film_fnv1a": film_hash,
            "film_matches_one_rank": film_ok,
            "per_rank_device_ms": per_rank_ms,
        }
        scene_for_cpu = b.scene
    b_main = b
    if world == 1 and not args.no_extra and args.workload == "c4":
        # the smaller single-GPU configs, same accounting (they are parity-test cases, not the headline)
        extra = []
        for name in ("c2", "c3"):
            e = Bench(name, 0, 1, local_rank, coll_dev, args.paths_in_flight, ctx=b.ctx)
            ea = timed(e, 2 if name == "c3" else 5, 1, barrier)
            esteps = 2 if name == "c3" else 5
            ec = e.counted()
            rf = roofline(e, ea, ec, esteps)
            extra.append({"workload": e.desc, "value": ea["rays"] / ea["dt"] / 1e6, "unit": "Mrays/s",
                          "ms_per_step": ea["dt"] / esteps * 1e3, "steps": esteps,
                          "roofline": {k: rf[k] for k in ("kernel", "achieved", "frac", "frac_is", "frac_algorithmic_incl_cache_hits",
                                                          "hbm_frac_rocprof", "hbm_frac_rocprof_step", "traffic_source")},
                          "k_shade_ms_per_step": ea["shade_ms"] / esteps, "k_trace_ms_per_step": ea["trace_ms"] / esteps,
                          "k_classify_ms_per_step": ea["classify_ms"] / esteps})
            e.close()
        # the f32 fast mode on the headline workload: reported, never the headline (SURVEY.md 8d tolerance row)
        f = Bench(args.workload, 0, 1, local_rank, coll_dev, args.paths_in_flight, precision=1, ctx=b.ctx)
        fa = timed(f, 2, 1, barrier)
        img32 = (f.d_rgb / f.d_n[..., None].clamp(min=1)).clamp(0.0, 10.0)
        b.step()
        img64 = (b.d_rgb / b.d_n[..., None].clamp(min=1)).clamp(0.0, 10.0)
        d2 = ((img32 - img64) ** 2).sum(dim=-1).reshape(-1)
        d2 = torch.where(torch.isfinite(d2), d2, torch.zeros_like(d2))
        keep = d2.sort().values[: d2.numel() - int(d2.numel() * 1e-4 + 0.999)]
        extra.append({"workload": f.desc + " -- RT_PRECISION_F32 fast mode", "dtype": "f32",
                      "value": fa["rays"] / fa["dt"] / 1e6, "unit": "Mrays/s", "ms_per_step": fa["dt"] / 2 * 1e3, "steps": 2,
                      "k_shade_ms_per_step": fa["shade_ms"] / 2, "k_trace_ms_per_step": fa["trace_ms"] / 2,
                      "k_classify_ms_per_step": fa["classify_ms"] / 2,
                      "rmse_vs_f64_same_seed": float(torch.sqrt(d2.mean() / 3.0)),
                      "rmse_vs_f64_without_top_0.01pct": float(torch.sqrt(keep.mean() / 3.0)),
                      "image_mean_f64": float(img64.mean()), "image_mean_f32": float(img32.mean()),
                      "note": "paths diverge from the f64 ones after a few bounces: the per-pixel difference is "
                              "Monte-Carlo noise at this spp, not rounding"})
        f.close()
        out["extra"] = extra
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene_for_cpu, b_main.W, b_main.H)
        print(json.dumps(out), flush=True)
    b_main.close()
    if world > 1:
        dist.barrier()
        if rank == 0 and cache_dir:
            import shutil
            shutil.rmtree(cache_dir, ignore_errors=True)
        dist.destroy_process_group()


if __name__ == "__main__":
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from rustraytracer_amd.dist import run_guarded
        run_guarded(main)  # any rank's failure = exit code 1 of that rank at once; the others follow (timeout / torchrun)
    else:
        main()

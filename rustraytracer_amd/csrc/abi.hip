// abi.hip -- implementation of the C ABI declared in include/rt_abi.h.
// Handles, validation, HBM residency, the per-chunk launch schedule and timing.
// Nothing here falls back to a CPU path: without a HIP device every compute entry
// returns RT_ERR_NO_DEVICE / RT_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

#include "bvh_build.h"
#include "bvh_cache.h"
#include "bvh_gpu.h"
#include "env_dist.h"
#include "device/shading.h"  // geom.h (k_wf_collect re-runs a winner's own test) + the shading-feature masks
#include "kernels_api.h"     // the kernels themselves are instantiated in tu/*.hip

using namespace rtd;

namespace rtk {
KernelTable& kernel_table() {
    static KernelTable t{};
    return t;
}
}  // namespace rtk

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                            \
    } while (0)

// ------------------------------------------------------------------ handles
// One in-flight chunk: its own stream, path-state ping-pong buffers, queues and counters.  Two
// lanes run chunks concurrently so that one chunk's latency-bound traversal and short tail
// launches overlap the other's ALU-heavy shading.
struct Lane {
    hipStream_t stream = nullptr;
    uint32_t capacity = 0;
    void* pool = nullptr;
    size_t pool_bytes = 0;
    uint32_t pool_cls = 0;  // class lists the pool was laid out for
    PathState st[2] = {};
    uint32_t* queue[2] = {nullptr, nullptr};
    uint32_t* hitw = nullptr;  // the extension rays' hit words, by queue position (k_trace -> k_classify_*)
    uint32_t* cls_tab = nullptr;  // k_classify_*: [kMaxCls][kClassifyWavesMax] per-wave counts, then list offsets
    Lists lists{};          // class lists + fold lists (scene_dev.h)
    uint32_t slot_cap = 0;  // records per pool = capacity + room for the unused ends of the waves' chunks
    uint32_t q_cap = 0;     // entries per ray queue
    Ctl* ctl = nullptr;
    MirrorEntry* mirror_h = nullptr;  // pinned + mapped: counters published by k_trace
    MirrorEntry* mirror_d = nullptr;
    uint32_t seq = 1;  // next mirror sequence number (unique per iteration ever launched on this lane)
    std::vector<hipEvent_t> events;
    size_t ev_i = 0;
    // the light kernel's own stream (run_lane) and the events that order it against the lane's stream (no timing)
    hipStream_t side = nullptr;
    hipStream_t cls_side[2] = {nullptr, nullptr};  // class kernels of one bounce side by side (run_lane: RT_CLS_STREAMS)
    std::vector<hipEvent_t> sync_events;
    size_t sync_i = 0;
    // per-render results of this lane
    std::vector<std::pair<hipEvent_t, hipEvent_t>> trace_ev, classify_ev, shade_ev, light_ev;
    uint64_t trace_launches = 0, shade_launches = 0;
    int rc = RT_OK;
    char err[512] = "";
};
constexpr int kLanes = 2;

struct rt_context {
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    Lane lanes[kLanes];
    int n_lanes = 1;  // RT_LANES=2 runs two pools concurrently (same throughput once the pool is large)
    DevStats* stats = nullptr;
    BatchCtl* batch = nullptr;
    double* lf = nullptr;  // film staging of the current batch: the radiance of retired paths, 3 doubles per camera sample
    size_t lf_capacity = 0;
    uint32_t* pix_list = nullptr;
    size_t pix_capacity = 0;
    std::vector<hipEvent_t> events;
    TraceTune tune{24, 4, 0, 128};
    uint32_t tail_paths = 786432;  // switch to the fused tail kernel at or below this many live paths
    std::vector<uint32_t> last_pix;  // pixel list of the last render on this context (host copy, for the gather)
    // ---- multi-device contexts (rt_context_create with n_devices > 1)
    // The primary context owns one single-device context per further device.  rt_render splits the caller's tiles
    // over all of them (one host thread per device), and each peer's own pixels are packed, copied peer-to-peer
    // to the primary device and scattered into the caller's film.
    std::vector<rt_context*> peers;
    double* pf_rgb = nullptr;   // as a peer: scratch film on this device
    uint32_t* pf_n = nullptr;
    size_t pf_cap = 0;
    double* pack = nullptr;     // as a peer: own pixels packed as {r, g, b, n} doubles
    size_t pack_cap = 0;
    double* stage = nullptr;    // as the primary: landing buffer of the peers' payloads, one after the other
    size_t stage_cap = 0;
    uint32_t* stage_pix = nullptr;  // ... and their pixel lists
    size_t stage_pix_cap = 0;
    hipEvent_t ev_gather = nullptr;
};

struct rt_scene {
    rt_context* ctx = nullptr;
    // host copies (set_* copies; commit consumes)
    struct HostMesh {
        std::vector<double> p, n, uv;
        std::vector<uint32_t> ind;
    };
    std::vector<HostMesh> meshes;
    std::vector<rt_primitive> prims;
    std::vector<rt_xform> xforms;
    std::vector<rt_material> mats;
    std::vector<rt_texture> texs;
    std::vector<std::vector<uint8_t>> hdr;  // texel copies of the RT_TEX_HDR textures (texs[i].rgbe points here)
    std::vector<rt_light> lights;
    bool committed = false;
    int shade_variant = 0;  // index into kFeatVariants: the smallest shading-kernel instance that covers the scene
    // vertex classes (scene_dev.h): class 0 = escaped; cls[k] for k >= 1 = {kernel instance, kind of hit record}
    uint32_t n_cls = 1;
    ClsDesc cls[kMaxCls] = {};
    bool all_lambert = false;  // every class runs the Lambert-only instance (then its 2-waves/SIMD build: kernels.hip)
    // device
    std::vector<void*> allocs;
    DevScene dev{};
    rt_scene_info info{};
    std::vector<rt_scene*> replicas;  // multi-device context: the device copies on ctx->peers[i] (no host data)
};

#ifdef RT_TEST_HOOKS
// (librt_amd_testhooks.so only -- csrc/Makefile: RT_TEST_POOL_OOM_ABOVE=<log2> makes pools of more than 2^log2 paths fail as if
// the memory were not there, so that tests/test_gpu_abi2.py can walk the default pool's fall-back without 288 GB of other
// allocations.  The product library does not read the variable.)
static uint32_t test_pool_oom_above() {
    static const uint32_t v = [] {
        const char* e = getenv("RT_TEST_POOL_OOM_ABOVE");
        return e ? (1u << std::min(28, std::max(6, atoi(e)))) : 0u;
    }();
    return v;
}
#else
static uint32_t test_pool_oom_above() { return 0u; }
#endif

// bytes of one lane's pool for `cap` paths in flight and `n_cls` vertex classes
static void pool_layout(uint32_t cap, uint32_t n_cls, size_t& slots, size_t& qn, size_t& bytes) {
    // every shading wave may leave the end of its last chunk of slots / queue entries / list entries unused: at most 1/16
    // of a launch's entries per kernel (kernels.hip: pick_chunk), summed generously
    slots = (size_t)cap + std::min<size_t>(cap / 16, (size_t)16 << 20) + 65536;
    qn = 3 * slots;
    // per slot and pool: the 256-B record + 15 ray / light-term words + 2 result words + the camera sample's RNG state;
    // 2 fold lists; per queue entry: 2 queues + hit words
    bytes = slots * (2 * ((size_t)kRecBytes + 15 * 8 + 2 * 4 + 8) + 2 * sizeof(uint32_t)) + 3 * qn * sizeof(uint32_t) +
            slots * sizeof(ListEnt) + 4096;  // (the class lists share one arena: every path is in at most one of them)
    (void)n_cls;
}

static int ensure_lane_capacity(rt_context* c, Lane& ln, uint32_t cap, uint32_t n_cls) {
    if (ln.capacity >= cap && ln.pool_cls >= n_cls) {
        ln.lists.n_cls = n_cls;
        return RT_OK;
    }
    if (test_pool_oom_above() && cap > test_pool_oom_above()) return fail(RT_ERR_OOM, "simulated: no memory for a pool of %u paths", cap);
    HIP_TRY(hipSetDevice(c->device));
    cap = std::max(cap, ln.capacity);
    const uint32_t want_cls = n_cls;
    n_cls = std::max(n_cls, ln.pool_cls);
    if (ln.pool) {
        HIP_TRY(hipFree(ln.pool));
        ln.pool = nullptr;
        ln.pool_bytes = 0;
        ln.capacity = 0;
        ln.pool_cls = 0;
    }
    // one slab: 2 pools of 256-B records, 2 ray queues, the class lists, 2 fold lists
    size_t slots, qn, bytes;
    pool_layout(cap, n_cls, slots, qn, bytes);
    void* slab = nullptr;
    HIP_TRY(hipMalloc(&slab, bytes));  // (on failure ln.pool stays null and ln.capacity 0: the caller may retry smaller)
    ln.pool = slab;
    ln.pool_bytes = bytes;
    char* p = (char*)ln.pool;
    for (int b = 0; b < 2; b++) {
        PathState& st = ln.st[b];
        st.rec = p;
        p += slots * kRecBytes;
        double** dptrs[] = {&st.ox, &st.oy, &st.oz, &st.dx, &st.dy, &st.dz, &st.spx, &st.spy, &st.spz, &st.pdx, &st.pdy, &st.pdz,
                            &st.fax, &st.fay, &st.faz};
        for (auto dp : dptrs) {
            *dp = (double*)p;
            p += slots * 8;
        }
        // (the kernels address these fifteen as ox + k * stride: kernels.hip, ldr / str)
        if (st.oy - st.ox != (ptrdiff_t)slots || st.faz - st.ox != 14 * (ptrdiff_t)slots) return fail(RT_ERR_HIP, "pool layout: ray arrays not equally spaced");
        st.rng0 = (uint64_t*)p; p += slots * 8;
        st.sh_prim = (int32_t*)p; p += slots * 4;
        st.pr_prim = (int32_t*)p; p += slots * 4;
    }
    ln.lists.ent = (ListEnt*)p; p += slots * sizeof(ListEnt);
    ln.queue[0] = (uint32_t*)p; p += qn * sizeof(uint32_t);
    ln.queue[1] = (uint32_t*)p; p += qn * sizeof(uint32_t);
    ln.hitw = (uint32_t*)p; p += qn * sizeof(uint32_t);
    ln.lists.fold[0] = (uint32_t*)p; p += slots * sizeof(uint32_t);
    ln.lists.fold[1] = (uint32_t*)p; p += slots * sizeof(uint32_t);
    ln.lists.cap = (uint32_t)slots;
    ln.lists.n_cls = want_cls;
    ln.slot_cap = (uint32_t)slots;
    ln.q_cap = (uint32_t)qn;
    ln.capacity = cap;
    ln.pool_cls = n_cls;
    return RT_OK;
}

template <typename T>
static int upload(rt_scene* s, const T* host, size_t count, const T** dev) {
    *dev = nullptr;
    if (count == 0) return RT_OK;
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, count * sizeof(T)));
    s->allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice));
    s->info.device_bytes_total += count * sizeof(T);
    *dev = (const T*)d;
    return RT_OK;
}

static int context_init(rt_context* c);

extern "C" {

const char* rt_last_error(void) { return g_err; }
int rt_abi_version(void) { return RT_ABI_VERSION; }
uint32_t rt_tile_owner(uint32_t tx, uint32_t ty, uint32_t world) { return rtabi_tile_owner(tx, ty, world); }

int rt_context_create(const int* device_ids, int n_devices, rt_context** out) {
    if (!out) return fail(RT_ERR_INVALID_ARG, "rt_context_create: out is null");
    *out = nullptr;
    if (n_devices < 0 || n_devices > 64) return fail(RT_ERR_INVALID_ARG, "rt_context_create: n_devices out of range");
    if (n_devices > 0 && !device_ids) return fail(RT_ERR_INVALID_ARG, "rt_context_create: device_ids is null");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device: %s", hipGetErrorString(e));
    int dev = n_devices >= 1 ? device_ids[0] : 0;
    for (int i = 0; i < n_devices; i++)
        if (device_ids[i] < 0 || device_ids[i] >= count)
            return fail(RT_ERR_NO_DEVICE, "device %d out of range (%d devices)", device_ids[i], count);
    rt_context* c = new (std::nothrow) rt_context();
    if (!c) return fail(RT_ERR_OOM, "host allocation failed");
    c->device = dev;
    int rc = context_init(c);
    // further devices: one single-device context each (a device id may repeat: two contexts then share that GPU,
    // which is how the multi-device path is exercised on a one-GPU box)
    for (int i = 1; rc == RT_OK && i < n_devices; i++) {
        rt_context* p = nullptr;
        rc = rt_context_create(&device_ids[i], 1, &p);
        if (rc != RT_OK) break;
        c->peers.push_back(p);
        if (p->device != c->device) {  // direct peer-to-peer copies over xGMI where the platform allows them
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, c->device, p->device) == hipSuccess && can) {
                (void)hipSetDevice(c->device);
                hipError_t pe = hipDeviceEnablePeerAccess(p->device, 0);
                if (pe != hipSuccess) (void)hipGetLastError();  // already enabled / unsupported: hipMemcpyPeer still works
                (void)hipSetDevice(p->device);
                pe = hipDeviceEnablePeerAccess(c->device, 0);
                if (pe != hipSuccess) (void)hipGetLastError();
            }
        }
    }
    if (rc != RT_OK) {
        rt_context_destroy(c);  // tolerates partially initialised members; g_err keeps the cause
        return rc;
    }
    *out = c;
    return RT_OK;
}

static int context_init(rt_context* c) {
    const int dev = c->device;
    HIP_TRY(hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    if (const char* e = getenv("RT_TRACE_REFILL")) c->tune.refill_lanes = std::min(64, std::max(1, atoi(e)));
    if (const char* e = getenv("RT_TRACE_NODE_BIAS")) c->tune.node_bias = std::max(1, atoi(e));
    if (const char* e = getenv("RT_TRACE_RESERVE")) c->tune.reserve = std::max(64, atoi(e) & ~63);
    if (const char* e = getenv("RT_TAIL_PATHS")) c->tail_paths = (uint32_t)std::max(0, atoi(e));
    if (const char* e = getenv("RT_LANES")) c->n_lanes = std::min(kLanes, std::max(1, atoi(e)));
    for (int i = 0; i < kLanes; i++) {
        Lane& ln = c->lanes[i];
        HIP_TRY(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&ln.side, hipStreamNonBlocking));
        for (auto& cs : ln.cls_side) HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        HIP_TRY(hipHostMalloc((void**)&ln.mirror_h, sizeof(MirrorEntry) * kRing, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(ln.mirror_h, 0, sizeof(MirrorEntry) * kRing);
        HIP_TRY(hipHostGetDevicePointer((void**)&ln.mirror_d, ln.mirror_h, 0));
        HIP_TRY(hipMalloc((void**)&ln.ctl, sizeof(Ctl)));
        HIP_TRY(hipMalloc((void**)&ln.cls_tab, sizeof(uint32_t) * kMaxCls * 8192));
    }
    HIP_TRY(hipMalloc((void**)&c->stats, sizeof(DevStats) * kStatShards));
    HIP_TRY(hipMalloc((void**)&c->batch, sizeof(BatchCtl)));
    return RT_OK;
}

int rt_context_destroy(rt_context* c) {
    if (!c) return RT_OK;
    for (rt_context* p : c->peers) (void)rt_context_destroy(p);
    (void)hipSetDevice(c->device);
    if (c->pf_rgb) (void)hipFree(c->pf_rgb);
    if (c->pf_n) (void)hipFree(c->pf_n);
    if (c->pack) (void)hipFree(c->pack);
    if (c->stage) (void)hipFree(c->stage);
    if (c->stage_pix) (void)hipFree(c->stage_pix);
    if (c->ev_gather) (void)hipEventDestroy(c->ev_gather);
    for (auto ev : c->events) (void)hipEventDestroy(ev);
    for (int i = 0; i < kLanes; i++) {
        Lane& ln = c->lanes[i];
        for (auto ev : ln.events) (void)hipEventDestroy(ev);
        for (auto ev : ln.sync_events) (void)hipEventDestroy(ev);
        if (ln.side) (void)hipStreamDestroy(ln.side);
        for (auto cs : ln.cls_side) if (cs) (void)hipStreamDestroy(cs);
        if (ln.pool) (void)hipFree(ln.pool);
        if (ln.ctl) (void)hipFree(ln.ctl);
        if (ln.cls_tab) (void)hipFree(ln.cls_tab);
        if (ln.mirror_h) (void)hipHostFree(ln.mirror_h);
        if (ln.stream) (void)hipStreamDestroy(ln.stream);
    }
    if (c->stats) (void)hipFree(c->stats);
    if (c->batch) (void)hipFree(c->batch);
    if (c->lf) (void)hipFree(c->lf);
    if (c->pix_list) (void)hipFree(c->pix_list);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RT_OK;
}

int rt_scene_create(rt_context* ctx, rt_scene** out) {
    if (!ctx || !out) return fail(RT_ERR_INVALID_ARG, "rt_scene_create: null argument");
    rt_scene* s = new (std::nothrow) rt_scene();
    if (!s) return fail(RT_ERR_OOM, "host allocation failed");
    s->ctx = ctx;
    for (rt_context* p : ctx->peers) {  // device copies for the other devices; filled by rt_scene_commit
        rt_scene* r = new (std::nothrow) rt_scene();
        if (!r) {
            rt_scene_destroy(s);
            return fail(RT_ERR_OOM, "host allocation failed");
        }
        r->ctx = p;
        s->replicas.push_back(r);
    }
    *out = s;
    return RT_OK;
}

#define SCENE_MUTABLE(s)                                                      \
    if (!(s)) return fail(RT_ERR_INVALID_ARG, "scene is null");               \
    if ((s)->committed) return fail(RT_ERR_STATE, "scene is already committed")

int rt_scene_set_meshes(rt_scene* s, const rt_mesh* meshes, uint64_t count) {
    SCENE_MUTABLE(s);
    if (count && !meshes) return fail(RT_ERR_INVALID_ARG, "meshes is null");
    s->meshes.clear();
    for (uint64_t i = 0; i < count; i++) {
        const rt_mesh& m = meshes[i];
        if (!m.p || !m.ind || m.n_p == 0 || m.n_ind % 3 != 0)
            return fail(RT_ERR_INVALID_ARG, "mesh %llu: missing positions/indices or n_ind not a multiple of 3",
                        (unsigned long long)i);
        if ((m.n_n != 0 && m.n_n != m.n_p) || (m.n_uv != 0 && m.n_uv != m.n_p) || (m.n_n && !m.n) || (m.n_uv && !m.uv))
            return fail(RT_ERR_INVALID_ARG, "mesh %llu: normals/uvs must be absent or one per position",
                        (unsigned long long)i);
        for (uint64_t k = 0; k < m.n_ind; k++)
            if (m.ind[k] >= m.n_p)
                return fail(RT_ERR_INVALID_ARG, "mesh %llu: index %u out of range", (unsigned long long)i, m.ind[k]);
        rt_scene::HostMesh hm;
        hm.p.assign(m.p, m.p + m.n_p * 3);
        if (m.n_n) hm.n.assign(m.n, m.n + m.n_n * 3);
        if (m.n_uv) hm.uv.assign(m.uv, m.uv + m.n_uv * 2);
        hm.ind.assign(m.ind, m.ind + m.n_ind);
        s->meshes.push_back(std::move(hm));
    }
    return RT_OK;
}
int rt_scene_set_primitives(rt_scene* s, const rt_primitive* prims, uint64_t count) {
    SCENE_MUTABLE(s);
    if (count && !prims) return fail(RT_ERR_INVALID_ARG, "prims is null");
    // a leaf code is (first_slot * 8 + count - 1) | kLeafCodeOther (scene_dev.h): slots must stay below bit 27
    static_assert((((uint64_t)kMaxPrims - 1) * 8 + 7) < (uint64_t)kLeafCodeOther, "leaf code layout");
    if (count >= kMaxPrims) return fail(RT_ERR_UNSUPPORTED, "too many primitives (limit 2^27 - 1)");
    s->prims.assign(prims, prims + count);
    return RT_OK;
}
int rt_scene_set_transforms(rt_scene* s, const rt_xform* x, uint64_t count) {
    SCENE_MUTABLE(s);
    if (count && !x) return fail(RT_ERR_INVALID_ARG, "xforms is null");
    s->xforms.assign(x, x + count);
    return RT_OK;
}
int rt_scene_set_materials(rt_scene* s, const rt_material* m, uint64_t count) {
    SCENE_MUTABLE(s);
    if (count && !m) return fail(RT_ERR_INVALID_ARG, "materials is null");
    s->mats.assign(m, m + count);
    return RT_OK;
}
int rt_scene_set_textures(rt_scene* s, const rt_texture* t, uint64_t count) {
    SCENE_MUTABLE(s);
    if (count && !t) return fail(RT_ERR_INVALID_ARG, "textures is null");
    for (uint64_t i = 0; i < count; i++)
        if (t[i].kind == RT_TEX_HDR && (!t[i].rgbe || t[i].width == 0 || t[i].height == 0 || t[i].width > 32768 || t[i].height > 32768))
            return fail(RT_ERR_INVALID_ARG, "texture %llu: HDR texels missing or bad size", (unsigned long long)i);
    s->texs.assign(t, t + count);
    s->hdr.assign(count, {});
    for (uint64_t i = 0; i < count; i++) {
        if (t[i].kind != RT_TEX_HDR) continue;
        s->hdr[i].assign(t[i].rgbe, t[i].rgbe + (size_t)t[i].width * t[i].height * 4);
        s->texs[i].rgbe = s->hdr[i].data();
    }
    return RT_OK;
}
int rt_scene_set_lights(rt_scene* s, const rt_light* l, uint64_t count) {
    SCENE_MUTABLE(s);
    if (count && !l) return fail(RT_ERR_INVALID_ARG, "lights is null");
    if (count > 65535) return fail(RT_ERR_UNSUPPORTED, "more than 65535 lights");
    s->lights.assign(l, l + count);
    return RT_OK;
}

static int validate_scene(const rt_scene* s) {
    for (size_t i = 0; i < s->texs.size(); i++) {
        const rt_texture& t = s->texs[i];
        if (t.kind == RT_TEX_CHECKERED) {
            if (t.odd >= s->texs.size() || t.even >= s->texs.size())
                return fail(RT_ERR_INVALID_ARG, "texture %zu: checker child out of range", i);
        } else if (t.kind != RT_TEX_SOLID && t.kind != RT_TEX_HDR) {
            return fail(RT_ERR_UNSUPPORTED, "texture %zu: kind %u is outside the hot-path scope", i, t.kind);
        }
    }
    auto tex_ok = [&](uint32_t id) { return id < s->texs.size(); };
    for (size_t i = 0; i < s->mats.size(); i++) {
        const rt_material& m = s->mats[i];
        switch (m.kind) {
            case RT_MAT_MATTE:
                if (!tex_ok(m.tex[0])) return fail(RT_ERR_INVALID_ARG, "material %zu: texture out of range", i);
                if (m.f[0] != 0.0) return fail(RT_ERR_UNSUPPORTED, "material %zu: Oren-Nayar (sigma != 0) is out of scope", i);
                break;
            case RT_MAT_LIGHT: break;
            case RT_MAT_PLASTIC:
            case RT_MAT_GLASS:
                if (!tex_ok(m.tex[0]) || !tex_ok(m.tex[1])) return fail(RT_ERR_INVALID_ARG, "material %zu: texture out of range", i);
                break;
            case RT_MAT_METAL:
                if (!tex_ok(m.tex[0]) || !tex_ok(m.tex[1])) return fail(RT_ERR_INVALID_ARG, "material %zu: texture out of range", i);
                for (int k = 3; k <= 4; k++) {
                    uint32_t id = m.tex[k] == RT_NO_TEXTURE ? m.tex[2] : m.tex[k];
                    if (!tex_ok(id)) return fail(RT_ERR_INVALID_ARG, "material %zu: roughness texture out of range", i);
                }
                break;
            case RT_MAT_MIRROR:
                if (!tex_ok(m.tex[0])) return fail(RT_ERR_INVALID_ARG, "material %zu: texture out of range", i);
                break;
            default: return fail(RT_ERR_UNSUPPORTED, "material %zu: kind %u is outside the hot-path scope", i, m.kind);
        }
    }
    for (size_t i = 0; i < s->prims.size(); i++) {
        const rt_primitive& p = s->prims[i];
        if (p.kind > RT_PRIM_YZ_RECT) return fail(RT_ERR_INVALID_ARG, "primitive %zu: bad kind %u", i, p.kind);
        if (p.mat_index >= s->mats.size()) return fail(RT_ERR_INVALID_ARG, "primitive %zu: material out of range", i);
        if (p.light_index >= (int32_t)s->lights.size()) return fail(RT_ERR_INVALID_ARG, "primitive %zu: light out of range", i);
        if (p.kind == RT_PRIM_TRIANGLE) {
            if (p.mesh_index >= s->meshes.size()) return fail(RT_ERR_INVALID_ARG, "primitive %zu: mesh out of range", i);
            if ((size_t)p.tri_ind + 2 >= s->meshes[p.mesh_index].ind.size())
                return fail(RT_ERR_INVALID_ARG, "primitive %zu: tri_ind out of range", i);
        } else if (p.kind != RT_PRIM_SPHERE) {
            if (p.xform_index >= (int32_t)s->xforms.size()) return fail(RT_ERR_INVALID_ARG, "primitive %zu: transform out of range", i);
        }
    }
    int n_infinite = 0;
    for (size_t i = 0; i < s->lights.size(); i++) {
        const rt_light& l = s->lights[i];
        if (l.kind == RT_LIGHT_INFINITE) {
            if (++n_infinite > 1) return fail(RT_ERR_UNSUPPORTED, "light %zu: more than one infinite light", i);
            if (l.tex_index >= s->texs.size() || s->texs[l.tex_index].kind != RT_TEX_HDR)
                return fail(RT_ERR_INVALID_ARG, "light %zu: the environment map must be an RT_TEX_HDR texture", i);
            if (l.xform_index >= (int32_t)s->xforms.size()) return fail(RT_ERR_INVALID_ARG, "light %zu: transform out of range", i);
            continue;
        }
        if (l.kind != RT_LIGHT_DIFFUSE) return fail(RT_ERR_UNSUPPORTED, "light %zu: only Light::Diffuse and Light::Infinite are in scope", i);
        if (l.prim_index >= s->prims.size()) return fail(RT_ERR_INVALID_ARG, "light %zu: primitive out of range", i);
    }
    return RT_OK;
}

int rt_scene_commit(rt_scene* s) { return rt_scene_commit_ex(s, RT_COMMIT_HOST_SAH); }

// What the host builder produces, kept so that the replicas of a multi-device context upload the same arrays
// instead of building the tree again.
struct HostLeafData {
    bool valid = false;
    std::vector<DevNode> nodes;
    std::vector<uint32_t> leaf_prim;
    std::vector<double> leaf_tri, leaf_nrm;
    std::vector<LeafMeta> leaf_meta;
    uint32_t depth = 0;
    uint64_t n_tri = 0;
};
// scene_dev.h: leaf_trav -- the leaf slots re-laid for the traversal kernel, one 128-B line each (bit copies)
__global__ __launch_bounds__(256) void k_make_leaf_trav(const double* __restrict__ leaf_tri, const uint32_t* __restrict__ leaf_prim,
                                                       uint32_t n, double* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long* s = reinterpret_cast<const unsigned long long*>(leaf_tri) + (size_t)i * 9;
    unsigned long long* o = reinterpret_cast<unsigned long long*>(out) + (size_t)i * 16;
    const uint32_t e = leaf_prim[i];
    if (e & kLeafOther) {
        for (int k = 0; k < 6; k++) o[k] = s[k];
        for (int k = 6; k < 15; k++) o[k] = 0ull;
    } else {
        for (int c = 0; c < 3; c++)
            for (int v = 0; v < 3; v++) o[c * 3 + v] = s[v * 3 + c];
        for (int v = 0; v < 3; v++) {
            o[9 + v] = s[v * 3 + 0];
            o[12 + v] = s[v * 3 + 1];
        }
    }
    o[15] = (unsigned long long)e;
}

// scene_dev.h: the vertex class of every leaf slot goes into bits 27-30 of its leaf_prim word (tab: per material, the
// class of a mesh hit and of a sphere / rect hit)
__global__ __launch_bounds__(256) void k_set_leaf_cls(uint32_t* leaf_prim, const LeafMeta* __restrict__ leaf_meta,
                                                     const uint8_t* __restrict__ tab, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t e = leaf_prim[i] & (kIdxMask | kLeafOther);
    const uint32_t mat = leaf_meta[i].mat_flags & kMetaMatMask;
    const uint32_t c = tab[2u * mat + ((e & kLeafOther) ? 1u : 0u)];
    leaf_prim[i] = e | (c << kClsShift);
}

static int commit_to(rt_scene* s, rt_scene* t, uint32_t flags, HostLeafData& cache);

// (host BVH shared between the processes of one node, RT_BVH_CACHE: bvh_cache.cpp)

// The builder emits nodes depth-first.  Move the top of the tree -- the first `n_top` nodes met breadth-first from the
// root, which every ray walks -- to the front, in that order: they then share a few cache lines, and a kernel can hold
// them on chip (k_trace's RT_LDS_NODES experiment).  Pure renumbering: the tree and every box stay what they were.
static void top_of_tree_first(std::vector<DevNode>& nodes, uint32_t n_top) {
    const uint32_t n = (uint32_t)nodes.size();
    if (n < 2) return;
    std::vector<uint32_t> order;  // new index -> old index
    std::vector<uint32_t> new_of(n, 0xffffffffu);
    order.reserve(n);
    order.push_back(0);
    new_of[0] = 0;
    for (size_t q = 0; q < order.size() && order.size() < n_top; q++)
        for (int k = 0; k < 4; k++) {
            const int32_t c = nodes[order[q]].child[k];
            if (c >= 0 && new_of[c] == 0xffffffffu && order.size() < n_top) {
                new_of[c] = (uint32_t)order.size();
                order.push_back((uint32_t)c);
            }
        }
    for (uint32_t i = 0; i < n; i++)
        if (new_of[i] == 0xffffffffu) {
            new_of[i] = (uint32_t)order.size();
            order.push_back(i);
        }
    std::vector<DevNode> out(n);
    for (uint32_t i = 0; i < n; i++) {
        DevNode nd = nodes[order[i]];
        for (int k = 0; k < 4; k++)
            if (nd.child[k] >= 0) nd.child[k] = (int32_t)new_of[nd.child[k]];
        out[i] = nd;
    }
    nodes.swap(out);
}

int rt_scene_commit_ex(rt_scene* s, uint32_t flags) {
    SCENE_MUTABLE(s);
    if (flags & ~(uint32_t)RT_COMMIT_DEVICE_LBVH) return fail(RT_ERR_INVALID_ARG, "unknown commit flags 0x%x", flags);
    int rc = validate_scene(s);
    if (rc != RT_OK) return rc;
    HostLeafData cache;
    rc = commit_to(s, s, flags, cache);
    // multi-device context: the scene is replicated on every device, same arrays, same tree
    for (size_t i = 0; rc == RT_OK && i < s->replicas.size(); i++) rc = commit_to(s, s->replicas[i], flags, cache);
    if (rc != RT_OK) {
        // a copy that failed half way (e.g. out of memory on one peer) must not leave the scene renderable: every
        // device copy made so far is released and the scene stays mutable, so the caller may retry or destroy it
        auto release = [](rt_scene* t) {
            (void)hipSetDevice(t->ctx->device);
            for (void* p : t->allocs) (void)hipFree(p);
            t->allocs.clear();
            t->dev = DevScene{};
            t->committed = false;
        };
        release(s);
        for (rt_scene* r : s->replicas) release(r);
        return rc;
    }
    return RT_OK;
}

// Uploads the scene held (on the host) by `s` to the device of `t` (t == s for the primary copy).
static int commit_to(rt_scene* s, rt_scene* t, uint32_t flags, HostLeafData& cache) {
    int rc = RT_OK;
    HIP_TRY(hipSetDevice(t->ctx->device));
    const size_t np = s->prims.size();
    t->info = rt_scene_info{};
    DevScene& d = t->dev;
    d = DevScene{};
    // records the builders and the kernels read
    if ((rc = upload(t, s->prims.data(), s->prims.size(), &d.prims)) != RT_OK) return rc;
    std::vector<DevMesh> dm(s->meshes.size());
    uint32_t has_uv = 0;
    for (size_t i = 0; i < s->meshes.size(); i++) {
        auto& m = s->meshes[i];
        if ((rc = upload(t, m.p.data(), m.p.size(), &dm[i].p)) != RT_OK) return rc;
        if ((rc = upload(t, m.n.data(), m.n.size(), &dm[i].n)) != RT_OK) return rc;
        if ((rc = upload(t, m.uv.data(), m.uv.size(), &dm[i].uv)) != RT_OK) return rc;
        if ((rc = upload(t, m.ind.data(), m.ind.size(), &dm[i].ind)) != RT_OK) return rc;
        if (!m.uv.empty()) has_uv = 1;
    }
    if ((rc = upload(t, dm.data(), dm.size(), &d.meshes)) != RT_OK) return rc;
    if ((rc = upload(t, s->xforms.data(), s->xforms.size(), &d.xforms)) != RT_OK) return rc;
    {   // materials with their texture references resolved (scene_dev.h: DevMat)
        std::vector<DevMat> dmats(s->mats.size());
        for (size_t i = 0; i < dmats.size(); i++) {
            DevMat& dm = dmats[i];
            std::memset(&dm, 0, sizeof(dm));
            dm.m = s->mats[i];
            for (int k = 0; k < 5; k++) dm.tex[k] = dm.m.tex[k];
            if (dm.m.kind == RT_MAT_METAL)
                for (int k = 3; k <= 4; k++)
                    if (dm.tex[k] == RT_NO_TEXTURE) dm.tex[k] = dm.tex[2];
            for (int k = 0; k < 5; k++) {
                if (dm.tex[k] >= s->texs.size() || s->texs[dm.tex[k]].kind != RT_TEX_SOLID) continue;
                dm.solid_mask |= 1u << k;
                for (int c = 0; c < 3; c++) dm.col[k][c] = s->texs[dm.tex[k]].color[c];
            }
        }
        if ((rc = upload(t, dmats.data(), dmats.size(), &d.mats)) != RT_OK) return rc;
    }
    {   // textures: HDR texels go to HBM first, the records point at the device copies
        std::vector<rt_texture> dtex = s->texs;
        for (size_t i = 0; i < dtex.size(); i++) {
            if (dtex[i].kind != RT_TEX_HDR) {
                dtex[i].rgbe = nullptr;
                continue;
            }
            const uint8_t* dp = nullptr;
            if ((rc = upload(t, s->hdr[i].data(), s->hdr[i].size(), &dp)) != RT_OK) return rc;
            dtex[i].rgbe = dp;
        }
        if ((rc = upload(t, dtex.data(), dtex.size(), &d.texs)) != RT_OK) return rc;
    }
    if ((rc = upload(t, s->lights.data(), s->lights.size(), &d.lights)) != RT_OK) return rc;
    {   // scene_dev.h: light_prims
        std::vector<rt_primitive> lp(s->lights.size());
        std::memset(lp.data(), 0, lp.size() * sizeof(rt_primitive));
        for (size_t i = 0; i < lp.size(); i++)
            if (s->lights[i].kind == RT_LIGHT_DIFFUSE) lp[i] = s->prims[s->lights[i].prim_index];
        if ((rc = upload(t, lp.data(), lp.size(), &d.light_prims)) != RT_OK) return rc;
    }
    d.n_mats = (uint32_t)s->mats.size();
    d.n_texs = (uint32_t)s->texs.size();
    d.env.light = -1;
    int need = 0;  // shading.h: kFeat*
    auto mat_need = [](const rt_material& m) {
        if (m.kind == RT_MAT_PLASTIC) return kFeatMicro | kFeatTwo;
        if (m.kind == RT_MAT_METAL) return kFeatMicro;
        if (m.kind == RT_MAT_MIRROR) return kFeatSpec;
        if (m.kind == RT_MAT_GLASS) return (m.f[0] != 0.0 || m.f[1] != 0.0) ? (kFeatTrans | kFeatMicro | kFeatTwo) : kFeatSpec;
        return 0;
    };
    for (const rt_material& m : s->mats) need |= mat_need(m);
    for (size_t i = 0; i < s->lights.size(); i++) {
        if (s->lights[i].kind != RT_LIGHT_INFINITE) continue;
        // Light::make_infinite_light's Distribution2D (light.rs:608-638), rebuilt from the texels
        EnvDist ed;
        build_env_dist(s->texs[s->lights[i].tex_index], ed);
        if ((rc = upload(t, ed.img.data(), ed.img.size(), &d.env.img)) != RT_OK) return rc;
        if ((rc = upload(t, ed.cond_cdf.data(), ed.cond_cdf.size(), &d.env.cond_cdf)) != RT_OK) return rc;
        if ((rc = upload(t, ed.marg_func.data(), ed.marg_func.size(), &d.env.marg_func)) != RT_OK) return rc;
        if ((rc = upload(t, ed.marg_cdf.data(), ed.marg_cdf.size(), &d.env.marg_cdf)) != RT_OK) return rc;
        d.env.marg_int = ed.marg_int;
        d.env.nu = ed.nu;
        d.env.nv = ed.nv;
        d.env.light = (int32_t)i;
        need |= kFeatEnv;
    }
    t->shade_variant = kNumFeatVariants - 1;
    for (int v = kNumFeatVariants - 1; v >= 0; v--)
        if ((need & ~kFeatVariants[v]) == 0) t->shade_variant = v;
    int forced_variant = -1;
    if (const char* e = getenv("RT_SHADE_VARIANT")) {  // experiment: force a larger instance
        const int v = std::min(kNumFeatVariants - 1, std::max(0, atoi(e)));
        if ((need & ~kFeatVariants[v]) == 0) t->shade_variant = forced_variant = v;
    }
    // ---- vertex classes (scene_dev.h): {smallest kernel instance covering the material (and the environment light, a
    // scene-wide feature), kind of hit record}, one class per combination the scene's primitives make
    std::vector<uint8_t> cls_tab(2 * std::max<size_t>(1, s->mats.size()), 0);
    {
        auto variant_for = [&](int nd) {
            if (forced_variant >= 0) return forced_variant;
            int best = kNumFeatVariants - 1;
            for (int v = kNumFeatVariants - 1; v >= 0; v--)
                if ((nd & ~kFeatVariants[v]) == 0) best = v;
            return best;
        };
        const bool one_class = getenv("RT_ONE_CLASS") != nullptr;  // experiment: no split by material / kind of hit
        const int tri_kind = has_uv ? kKindAny : kKindMesh;
        for (int pass = 0; pass < 2; pass++) {
            // pass 1 (only when pass 0 needed more than 15 classes, or RT_ONE_CLASS): the scene-wide instance for everything
            t->n_cls = 1;
            std::fill(cls_tab.begin(), cls_tab.end(), (uint8_t)0);
            bool ok = true;
            for (const rt_primitive& p : s->prims) {
                const bool tri = p.kind == RT_PRIM_TRIANGLE;
                const int kind = (pass == 1 && one_class) ? kKindAny : (tri ? tri_kind : kKindOther);
                const int v = pass == 0 ? variant_for(mat_need(s->mats[p.mat_index]) | (need & kFeatEnv)) : t->shade_variant;
                uint8_t& slot = cls_tab[2 * (size_t)p.mat_index + (tri ? 0 : 1)];
                if (slot) continue;
                uint32_t k = 1;
                for (; k < t->n_cls; k++)
                    if (t->cls[k].variant == v && t->cls[k].kind == kind) break;
                if (k == t->n_cls) {
                    if (t->n_cls >= (uint32_t)kMaxCls) {
                        ok = false;
                        break;
                    }
                    t->cls[k] = ClsDesc{(uint8_t)v, (uint8_t)kind};
                    t->n_cls++;
                }
                slot = (uint8_t)k;
            }
            if (ok && !(pass == 0 && one_class)) break;
        }
        t->all_lambert = !getenv("RT_NO_LAMBERT_W2");
        for (uint32_t k = 1; k < t->n_cls; k++) t->all_lambert = t->all_lambert && t->cls[k].variant == 0;
    }
    uint64_t n_tri = 0;
    uint32_t depth = 0;
    bool any_normals = false;
    for (const auto& m : s->meshes) any_normals = any_normals || !m.n.empty();
    if (np >= kMetaMatMask || s->mats.size() >= kMetaMatMask) return fail(RT_ERR_UNSUPPORTED, "too many primitives or materials");
    const auto t0 = std::chrono::steady_clock::now();
    if (flags & RT_COMMIT_DEVICE_LBVH) {
        // next-row f3: the tree is built where the primitives already are (bvh_gpu.hip)
        DeviceBvh gb;
        char berr[400] = "";
        rc = build_bvh_device(t->ctx->stream, d.prims, d.meshes, (uint32_t)np, any_normals, &gb, berr, sizeof(berr));
        if (rc != RT_OK) return fail(rc, "%s", berr);
        for (void* p : {(void*)gb.nodes, (void*)gb.leaf_prim, (void*)gb.leaf_tri, (void*)gb.leaf_nrm, (void*)gb.leaf_meta})
            if (p) t->allocs.push_back(p);
        d.nodes = gb.nodes;
        d.leaf_prim = gb.leaf_prim;
        d.leaf_tri = gb.leaf_tri;
        d.leaf_nrm = gb.leaf_nrm;
        d.leaf_meta = gb.leaf_meta;
        d.n_nodes = gb.n_nodes;
        t->info.device_bytes_total += (uint64_t)gb.n_nodes * sizeof(DevNode) + np * (sizeof(uint32_t) + 9 * sizeof(double) + sizeof(LeafMeta) + (any_normals ? 9 * sizeof(double) : 0));
        depth = gb.depth;
        n_tri = gb.n_triangles;
        t->info.build_device_ms = gb.build_ms;
    } else {
        if (!cache.valid) {
        BvhOut bvh;
        t->info.build_from_cache = (uint32_t)build_bvh_shared(s->prims.data(), s->prims.size(), bvh);
        if (bvh.depth + 1 > (uint32_t)kMaxBvhDepth) return fail(RT_ERR_UNSUPPORTED, "BVH depth %u exceeds the traversal stack", bvh.depth);
        // leaf-ordered triangle vertices + ids
        std::vector<uint32_t>& leaf_prim = cache.leaf_prim;
        std::vector<double>& leaf_tri = cache.leaf_tri;
        std::vector<double>& leaf_nrm = cache.leaf_nrm;
        std::vector<LeafMeta>& leaf_meta = cache.leaf_meta;
        leaf_prim.assign(np, 0u);
        leaf_tri.assign(np * 9, 0.0);
        leaf_nrm.assign(any_normals ? np * 9 : 0, 0.0);
        leaf_meta.assign(np, LeafMeta{});
        std::atomic<uint64_t> n_tri_atomic{0};
        const unsigned fill_threads = np >= 65536 ? std::max(1u, std::min(16u, std::thread::hardware_concurrency())) : 1u;
        auto fill = [&](size_t i0, size_t i1) {
        uint64_t n_tri_host = 0;
        for (size_t i = i0; i < i1; i++) {
            const uint32_t id = bvh.order[i];
            const rt_primitive& p = s->prims[id];
            leaf_meta[i] = LeafMeta{(p.mat_index & kMetaMatMask) | (p.flip ? kMetaFlip : 0u), p.light_index};
            if (p.kind == RT_PRIM_TRIANGLE) {
                const auto& m = s->meshes[p.mesh_index];
                for (int v = 0; v < 3; v++) {
                    const uint32_t vi = m.ind[p.tri_ind + v];
                    for (int a = 0; a < 3; a++) leaf_tri[i * 9 + v * 3 + a] = m.p[3 * vi + a];
                    if (!m.n.empty())
                        for (int a = 0; a < 3; a++) leaf_nrm[i * 9 + v * 3 + a] = m.n[3 * vi + a];
                }
                if (!m.n.empty()) leaf_meta[i].mat_flags |= kMetaHasNormals;
                leaf_prim[i] = id;
                n_tri_host++;
            } else {
                // sphere / rect: parameters + {kind, transform index + 1} ride in the vertex slot (geom.h: leaf_step)
                for (int a = 0; a < 5; a++) leaf_tri[i * 9 + a] = p.v[a];
                const uint64_t meta = (uint64_t)(p.kind & 0xffu) | ((uint64_t)(uint32_t)(p.xform_index + 1) << 32);
                std::memcpy(&leaf_tri[i * 9 + 5], &meta, 8);
                leaf_prim[i] = id | kLeafOther;
            }
        }
        n_tri_atomic += n_tri_host;
        };
        {   // the leaf slots are independent: filled by slices on several threads
            std::vector<std::thread> th;
            for (unsigned k = 1; k < fill_threads; k++) th.emplace_back(fill, np * k / fill_threads, np * (k + 1) / fill_threads);
            fill(0, np / fill_threads);
            for (auto& t : th) t.join();
        }
        const uint64_t n_tri_host = n_tri_atomic.load();
        cache.nodes = std::move(bvh.nodes);
        top_of_tree_first(cache.nodes, 256);
        if (getenv("RT_DIAG"))
            fprintf(stderr, "[rt diag] commit: tree + leaf arrays on the host %.1f ms\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        cache.depth = bvh.depth;
        cache.n_tri = n_tri_host;
        cache.valid = true;
        }
        if ((rc = upload(t, cache.nodes.data(), cache.nodes.size(), &d.nodes)) != RT_OK) return rc;
        if ((rc = upload(t, cache.leaf_prim.data(), cache.leaf_prim.size(), &d.leaf_prim)) != RT_OK) return rc;
        if ((rc = upload(t, cache.leaf_tri.data(), cache.leaf_tri.size(), &d.leaf_tri)) != RT_OK) return rc;
        if ((rc = upload(t, cache.leaf_nrm.data(), cache.leaf_nrm.size(), &d.leaf_nrm)) != RT_OK) return rc;
        if ((rc = upload(t, cache.leaf_meta.data(), cache.leaf_meta.size(), &d.leaf_meta)) != RT_OK) return rc;
        d.n_nodes = (uint32_t)cache.nodes.size();
        depth = cache.depth;
        n_tri = cache.n_tri;
    }
    if (np > 0) {  // the vertex class of every leaf slot, into its leaf_prim word (both builders leave the bits clear)
        const uint8_t* d_tab = nullptr;
        if ((rc = upload(t, cls_tab.data(), cls_tab.size(), &d_tab)) != RT_OK) return rc;
        hipLaunchKernelGGL(k_set_leaf_cls, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, t->ctx->stream,
                           const_cast<uint32_t*>(d.leaf_prim), d.leaf_meta, d_tab, (uint32_t)np);
        HIP_TRY(hipGetLastError());
    }
    if (np > 0) {  // the traversal kernel's one-line-per-slot copy of the leaf slots (scene_dev.h: leaf_trav)
        double* trav = nullptr;
        HIP_TRY(hipMalloc((void**)&trav, np * 16 * sizeof(double)));
        t->allocs.push_back(trav);
        hipLaunchKernelGGL(k_make_leaf_trav, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, t->ctx->stream, d.leaf_tri,
                           d.leaf_prim, (uint32_t)np, trav);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(t->ctx->stream));
        d.leaf_trav = trav;
        t->info.device_bytes_total += np * 16 * sizeof(double);
    }
    t->info.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    t->info.build_flags = flags;
    d.n_prims = (uint32_t)np;
    d.simple_others = getenv("RT_NO_SIMPLE_OTHERS") ? 0u : 1u;
    for (const rt_primitive& p : s->prims)
        if (p.kind == RT_PRIM_SPHERE || (p.kind != RT_PRIM_TRIANGLE && p.xform_index >= 0)) d.simple_others = 0u;
    d.n_lights = (uint32_t)s->lights.size();
    d.mesh_has_uv = has_uv;
    t->info.n_prims = np;
    t->info.n_triangles = n_tri;
    t->info.n_others = np - n_tri;
    t->info.n_bvh_nodes = d.n_nodes;
    t->info.bvh_depth = depth;
    t->info.n_classes = t->n_cls;
    t->info.node_bytes = sizeof(DevNode);
    // bytes a primitive test requests from its leaf_trav line: the nine coordinates / v[5] + meta, + the index word
    t->info.tri_bytes = 9 * sizeof(double) + sizeof(uint32_t);
    t->info.other_bytes = 6 * sizeof(double) + sizeof(uint32_t);
    t->committed = true;
    return RT_OK;
}

int rt_scene_destroy(rt_scene* s) {
    if (!s) return RT_OK;
    for (rt_scene* r : s->replicas) (void)rt_scene_destroy(r);
    (void)hipSetDevice(s->ctx->device);
    for (void* p : s->allocs) (void)hipFree(p);
    delete s;
    return RT_OK;
}

int rt_scene_get_info(const rt_scene* s, rt_scene_info* out) {
    if (!s || !out) return fail(RT_ERR_INVALID_ARG, "null argument");
    if (!s->committed) return fail(RT_ERR_STATE, "scene is not committed");
    *out = s->info;
    return RT_OK;
}

// RT_PRECISION_F32: leaf-ordered vertices / normals as floats; a sphere / rect slot keeps its {kind, transform} word
// bit for bit in floats 5 and 6 (scene_dev.h)
__global__ __launch_bounds__(256) void k_leaf_to_f32(const double* __restrict__ tri, const double* __restrict__ nrm,
                                                    const uint32_t* __restrict__ leaf_prim, uint32_t n, float* tri32,
                                                    float* nrm32) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* t = tri + (size_t)i * 9;
    float* o = tri32 + (size_t)i * 9;
    if (leaf_prim[i] & kLeafOther) {
        for (int k = 0; k < 5; k++) o[k] = (float)t[k];
        const unsigned long long meta = (unsigned long long)__double_as_longlong(t[5]);
        o[5] = __uint_as_float((uint32_t)(meta & 0xffffffffull));
        o[6] = __uint_as_float((uint32_t)(meta >> 32));
        o[7] = o[8] = 0.0f;
    } else {
        for (int k = 0; k < 9; k++) o[k] = (float)t[k];
    }
    if (nrm32)
        for (int k = 0; k < 9; k++) nrm32[(size_t)i * 9 + k] = (float)nrm[(size_t)i * 9 + k];
}

// First fast-mode use of a scene: binary32 copies of the leaf-ordered vertices / normals (scene_dev.h).  Every entry
// that launches rtd32 traversal code calls this first -- those kernels read leaf_tri32 unconditionally.
static int ensure_f32_leaves(rt_scene* s, hipStream_t stream) {
    if (s->dev.leaf_tri32 || !s->dev.n_prims) return RT_OK;
    const size_t nflt = (size_t)s->dev.n_prims * 9;
    float *t32 = nullptr, *n32 = nullptr;
    HIP_TRY(hipMalloc((void**)&t32, nflt * sizeof(float)));
    s->allocs.push_back(t32);
    if (s->dev.leaf_nrm) {
        HIP_TRY(hipMalloc((void**)&n32, nflt * sizeof(float)));
        s->allocs.push_back(n32);
    }
    hipLaunchKernelGGL(k_leaf_to_f32, dim3((unsigned)((s->dev.n_prims + 255) / 256)), dim3(256), 0, stream, s->dev.leaf_tri,
                       s->dev.leaf_nrm, s->dev.leaf_prim, s->dev.n_prims, t32, n32);
    HIP_TRY(hipGetLastError());
    s->dev.leaf_tri32 = t32;
    s->dev.leaf_nrm32 = n32;
    s->info.device_bytes_total += nflt * sizeof(float) * (n32 ? 2 : 1);
    return RT_OK;
}

static uint32_t next_pow2(uint32_t v) {  // sampler.rs:633-642
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

static hipEvent_t get_event(std::vector<hipEvent_t>& pool, size_t i, bool timing = true) {
    while (pool.size() <= i) {
        hipEvent_t ev;
        if ((timing ? hipEventCreate(&ev) : hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return nullptr;
        pool.push_back(ev);
    }
    return pool[i];
}

struct RenderJob {
    rt_context* c;
    rt_scene* s;
    rt_camera cam;
    const rt_render_cfg* cfg;
    ChunkDesc batch;                 // the batch both lanes are working on
    unsigned long long batch_total;  // its camera samples
    uint32_t pool;                   // paths each lane keeps in flight
    std::atomic<bool> abort{false};
    std::atomic<bool> cancelled{false};  // rt_render_cfg.cancel was seen non-zero
    int trace_blocks;
    bool light_overlap = false;  // the light kernel runs on the lane's side stream, under the next traversal launch
    int cls_streams = 2;         // streams the class kernels of one bounce alternate between
    int shade_blocks[kMaxCls];  // resident blocks of each class kernel (persistent grids); [0]: the light kernel
    bool count_trav;
    bool f32;  // RT_PRECISION_F32: the binary32 kernel set
    bool f32_gen, f32_trace, f32_shade, f32_tail;  // (debug: RT_F32_MIX picks the kernels that use it)
};

static int lane_fail(Lane& ln, int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ln.err, sizeof(ln.err), fmt, ap);
    va_end(ap);
    ln.rc = code;
    return code;
}
#define LANE_TRY(expr)                                                                                      \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            job.abort.store(true);                                                                          \
            return lane_fail(ln, RT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
        }                                                                                                   \
    } while (0)

// The kernels exist twice: binary64 (namespace rtd, the parity mode) and binary32 (namespace rtd32, the same sources
// transformed, RT_PRECISION_F32).  Path records, film staging and scene records are the same arrays in both, so the
// launch schedule below does not care which set it drives.  kernels_api.h: the table the kernels' translation units fill.
using rtk::ShadeClsKernel;
using rtk::ShadeLightKernel;
using rtk::TailKernel;
using rtk::TraceKernel;
using rtk::GenKernel;
static ShadeClsKernel shade_cls_kernel(const ClsDesc& cd, bool f32, bool all_lambert) {
    if (all_lambert && cd.variant == 0) return rtk::kernel_table().shade_cls_w2[f32 ? 1 : 0][cd.kind];
    return rtk::kernel_table().shade_cls[f32 ? 1 : 0][cd.variant][cd.kind];
}
static ShadeLightKernel shade_light_kernel(bool env, bool f32) { return rtk::kernel_table().shade_light[f32 ? 1 : 0][env ? 1 : 0]; }
static TailKernel tail_kernel(int v, bool count, bool f32) { return rtk::kernel_table().tail[f32 ? 1 : 0][v][count ? 1 : 0]; }
static TraceKernel trace_kernel(bool count, bool simple, bool f32) { return rtk::kernel_table().trace[f32 ? 1 : 0][count ? 2 : (simple ? 0 : 1)]; }
static GenKernel generate_kernel(bool f32) { return rtk::kernel_table().generate[f32 ? 1 : 0]; }

// One lane's share of a batch: keep `pool` paths alive, topping up from the shared batch counter,
// until the batch is exhausted and this lane's paths have all retired.
static int run_lane(RenderJob& job, int lane_id) {
    rt_context* c = job.c;
    Lane& ln = c->lanes[lane_id];
    const rt_render_cfg* cfg = job.cfg;
    LANE_TRY(hipSetDevice(c->device));
    static const bool no_ev = getenv("RT_NO_TRACE_EVENTS") != nullptr;
    hipStream_t stream = ln.stream;
    // The light kernel of iteration `it` needs the lists of `it` and writes film staging only: with job.light_overlap it
    // runs on the side stream from the end of k_classify_scatter(it) on, beside the class kernels of `it` and k_trace(it+1)
    // (memory-bound, a tenth of the VALU busy, under two issue-bound kernels), and has to be through before the next
    // scatter overwrites the lists -- and before anything else that touches both pools (k_tail) or ends the lane.
    hipEvent_t light_done = nullptr;
    auto join_light = [&]() -> hipError_t {
        if (!light_done) return hipSuccess;
        const hipError_t e = hipStreamWaitEvent(stream, light_done, 0);
        light_done = nullptr;
        return e;
    };
    const uint32_t P = job.pool;
    const uint32_t gen_blocks = (P + 255) / 256;
    LANE_TRY(hipMemsetAsync(ln.ctl, 0, sizeof(Ctl), stream));
    const uint32_t seq0 = ln.seq;
    // enough for every sample to be started and for the deepest path to finish, with slack
    const unsigned long long max_iters =
        (job.batch_total / std::max<uint32_t>(P, 1u) + 2ull) * ((unsigned long long)cfg->max_depth + 3ull) + 64ull;
    uint32_t bound_active = P;
    bool exhausted_known = false;
    unsigned long long it = 0;
    // whichever way this function is left, the next render on this lane must not reuse a sequence number that a
    // kernel still queued here may publish
    struct SeqGuard {
        Lane& ln;
        uint32_t seq0;
        const unsigned long long& it;
        ~SeqGuard() { ln.seq = seq0 + (uint32_t)it + 8u; }
    } seq_guard{ln, seq0, it};
    auto poll_cancel = [&]() -> bool {
        if (cfg->cancel && *cfg->cancel != 0) {
            job.cancelled.store(true);
            job.abort.store(true);
        }
        return job.abort.load();
    };
    // How far the host runs ahead of the counters it sizes the grids from (they never grow): k_shade's grid for iteration
    // `it` covers the paths that were alive `lag` iterations earlier, and every block beyond the live ones costs ~2 ns of
    // dispatch.  RT_MIRROR_LAG=1 against the default 2: C4 +0.3 %, C3 +0.2 %, C2 -1.5 % (the launch latency shows on
    // small frames) -- profiles/r03_sweep_mirror_lag.txt.
    static const unsigned long long lag = [] {
        const char* e = getenv("RT_MIRROR_LAG");
        return (unsigned long long)(e ? std::min(4, std::max(1, atoi(e))) : 2);
    }();
    for (; it < max_iters; it++) {
        if (poll_cancel()) return RT_ERR_HIP;
        static const bool no_mirror = getenv("RT_NO_MIRROR") != nullptr;  // experiment: fixed iteration count
        if (no_mirror) {
            if (it > (unsigned long long)cfg->max_depth + 2) break;
        } else if (it >= lag) {
            // counters published when k_trace(it - lag) started; `lag` iterations stay queued behind it
            volatile MirrorEntry* me = &ln.mirror_h[(it - lag) % kRing];
            const uint32_t want_seq = seq0 + (uint32_t)(it - lag);
            const auto t0 = std::chrono::steady_clock::now();
            uint64_t spins = 0;
            // The wait is normally a few microseconds (the device runs two iterations ahead of this read): spin
            // briefly, then back off to short sleeps so that a long kernel does not pin a host core.
            while (__atomic_load_n(&me->seq, __ATOMIC_ACQUIRE) != want_seq) {
                if (++spins < 2048) {
                    __builtin_ia32_pause();
                    continue;
                }
                std::this_thread::sleep_for(std::chrono::microseconds(spins < 4096 ? 5 : 50));
                if (poll_cancel()) return RT_ERR_HIP;
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) {
                    job.abort.store(true);
                    return lane_fail(ln, RT_ERR_HIP, "device did not publish iteration %llu counters within 30 s", it - lag);
                }
            }
            const uint32_t live = me->n_active, rem = me->remaining;
            if (rem == 0) {
                // the batch has been handed out completely: from here on the lane only drains
                exhausted_known = true;
                bound_active = std::min(bound_active, live);
                if (live == 0) break;
                if (live <= c->tail_paths) {
                    // few paths left: one fused launch finishes them (k_tail) instead of ~2 launches per bounce
                    // that are each as slow as their single longest ray
                    LANE_TRY(join_light());
                    const uint32_t tail_blocks = std::min((live + 255) / 256, (uint32_t)c->num_cus * 2u);  // persistent waves
                    hipLaunchKernelGGL(tail_kernel(job.s->shade_variant, job.count_trav, job.f32_tail), dim3(tail_blocks), dim3(256), 0, stream, job.s->dev, ln.st[0],
                                       ln.st[1], ln.ctl, (uint32_t)it, cfg->max_depth, ln.queue[it & 1], ln.lists, c->lf, c->stats);
                    break;
                }
            }
        }
        // top up the pool with new camera samples, then trace and shade everything alive
        hipLaunchKernelGGL(rtk::kernel_table().plan, dim3(1), dim3(1), 0, stream, ln.ctl, c->batch, (uint32_t)it, P, job.batch_total, c->stats);
        if (!exhausted_known)
            hipLaunchKernelGGL(generate_kernel(job.f32_gen), dim3(gen_blocks), dim3(256), 0, stream, ln.st[it & 1], job.cam, job.batch,
                               c->pix_list, ln.queue[it & 1], ln.ctl);
        const uint32_t shade_want = std::max(1u, (bound_active + 255u) / 256u);
        const uint32_t tblocks = std::max(
            1u, (uint32_t)std::min<uint64_t>((3ull * bound_active + 255) / 256, (uint64_t)job.trace_blocks));
        hipEvent_t a = nullptr, b = nullptr;
        if (!no_ev) {
            a = get_event(ln.events, ln.ev_i++);
            b = get_event(ln.events, ln.ev_i++);
            if (!a || !b) {
                job.abort.store(true);
                return lane_fail(ln, RT_ERR_HIP, "hipEventCreate failed");
            }
            LANE_TRY(hipEventRecord(a, stream));
        }
        const uint32_t seq = seq0 + (uint32_t)it;
        hipLaunchKernelGGL(trace_kernel(job.count_trav, job.s->dev.simple_others != 0, job.f32_trace), dim3(tblocks), dim3(256), 0, stream,
                           job.s->dev, ln.st[it & 1], ln.queue[it & 1], ln.ctl, (uint32_t)it, c->stats, c->tune,
                           no_mirror ? nullptr : ln.mirror_d, seq, c->batch, job.batch_total, ln.hitw);
        if (!no_ev) {
            LANE_TRY(hipEventRecord(b, stream));
            ln.trace_ev.emplace_back(a, b);
        }
        ln.trace_launches++;
        // the traced paths to the lists of their vertex classes: a counting sort over the queue (count, scan, scatter)
        {
            const uint32_t cblocks = std::min((3u * bound_active + 255u) / 256u + 1u, std::min((uint32_t)c->num_cus * 8u, 8192u / 4u));
            hipLaunchKernelGGL(rtk::kernel_table().classify_count, dim3(cblocks), dim3(256), 0, stream, ln.queue[it & 1], ln.hitw, ln.ctl,
                               (uint32_t)it, ln.cls_tab);
            hipLaunchKernelGGL(rtk::kernel_table().classify_scan, dim3(1), dim3(512), 0, stream, ln.cls_tab, cblocks * 4u, ln.ctl, (uint32_t)it);
            LANE_TRY(join_light());  // (the light kernel of the previous iteration still reads the lists)
            hipLaunchKernelGGL(rtk::kernel_table().classify_scatter, dim3(cblocks), dim3(256), 0, stream, ln.queue[it & 1], ln.hitw,
                               ln.ctl, (uint32_t)it, ln.cls_tab, ln.lists);
        }
        hipEvent_t b2 = b;
        if (!no_ev) {  // (the classify launches run from event b, the end of k_trace, to this one)
            b2 = get_event(ln.events, ln.ev_i++);
            if (!b2) {
                job.abort.store(true);
                return lane_fail(ln, RT_ERR_HIP, "hipEventCreate failed");
            }
            LANE_TRY(hipEventRecord(b2, stream));
            ln.classify_ev.emplace_back(b, b2);
        }
        // the paths that end without a vertex (escaped, fold only): the light kernel -- on the side stream when overlapped
        auto launch_light = [&](hipStream_t on) -> int {
            hipEvent_t l0 = nullptr, l1 = nullptr;
            if (!no_ev) {
                l0 = get_event(ln.events, ln.ev_i++);
                l1 = get_event(ln.events, ln.ev_i++);
                if (!l0 || !l1) {
                    job.abort.store(true);
                    return lane_fail(ln, RT_ERR_HIP, "hipEventCreate failed");
                }
                LANE_TRY(hipEventRecord(l0, on));
            }
            hipLaunchKernelGGL(shade_light_kernel(job.s->dev.env.light >= 0, job.f32_shade), dim3(std::min(shade_want, (uint32_t)job.shade_blocks[0])), dim3(256),
                               0, on, job.s->dev, ln.st[it & 1], ln.ctl, (uint32_t)it, cfg->max_depth, ln.lists, c->lf);
            if (!no_ev) {
                LANE_TRY(hipEventRecord(l1, on));
                ln.light_ev.emplace_back(l0, l1);
            }
            return RT_OK;
        };
        if (job.light_overlap) {
            hipEvent_t lists_ready = get_event(ln.sync_events, ln.sync_i++, false);
            light_done = get_event(ln.sync_events, ln.sync_i++, false);
            if (!lists_ready || !light_done) {
                job.abort.store(true);
                return lane_fail(ln, RT_ERR_HIP, "hipEventCreate failed");
            }
            LANE_TRY(hipEventRecord(lists_ready, stream));
            LANE_TRY(hipStreamWaitEvent(ln.side, lists_ready, 0));
            if (int rc = launch_light(ln.side)) return rc;
            LANE_TRY(hipEventRecord(light_done, ln.side));
        }
        // one kernel per vertex class, the heaviest instances first.  Persistent grids: a class with few paths this bounce
        // costs a launch, not a grid of empty blocks.
        // The class kernels of one bounce are independent of each other (shared cursors and counters are atomics): they
        // alternate between the lane's stream and a second one, so that two of them are resident together -- a glass class
        // that waits for its records beside a Lambert class that issues f64 arithmetic.  C4 4725 -> 4825 Mrays/s, C2 3940 ->
        // 4020, hdr 2135 -> 2245, C3 (two classes of the same kind of work) +- 0; three streams: no better than one
        // (profiles/r04_exp_class_streams.txt).  RT_CLS_STREAMS=1: one after the other.
        const int cls_streams = job.cls_streams;
        hipEvent_t cls_fork = nullptr;
        bool cls_used[2] = {false, false};
        if (cls_streams > 1 && job.s->n_cls > 2) {
            cls_fork = get_event(ln.sync_events, ln.sync_i++, false);
            LANE_TRY(hipEventRecord(cls_fork, stream));
        }
        for (uint32_t k = 1; k < job.s->n_cls; k++) {
            hipStream_t on = stream;
            const int w = cls_fork ? (int)((k - 1) % (uint32_t)cls_streams) : 0;
            if (w > 0) {
                on = ln.cls_side[w - 1];
                if (!cls_used[w - 1]) LANE_TRY(hipStreamWaitEvent(on, cls_fork, 0));
                cls_used[w - 1] = true;
            }
            hipLaunchKernelGGL(shade_cls_kernel(job.s->cls[k], job.f32_shade, job.s->all_lambert), dim3(std::min(shade_want, (uint32_t)job.shade_blocks[k])), dim3(256), 0,
                               on, job.s->dev, ln.st[it & 1], ln.st[(it + 1) & 1], ln.ctl, (uint32_t)it, cfg->max_depth, ln.lists, k,
                               ln.queue[(it + 1) & 1], ln.q_cap, ln.slot_cap, c->lf, c->stats);
        }
        for (int w = 0; w < 2; w++)
            if (cls_used[w]) {
                hipEvent_t j = get_event(ln.sync_events, ln.sync_i++, false);
                LANE_TRY(hipEventRecord(j, ln.cls_side[w]));
                LANE_TRY(hipStreamWaitEvent(stream, j, 0));
            }
        if (!no_ev) {  // the class kernels run from event b2 (end of the classify launches) to this one
            hipEvent_t e = get_event(ln.events, ln.ev_i++);
            if (!e) {
                job.abort.store(true);
                return lane_fail(ln, RT_ERR_HIP, "hipEventCreate failed");
            }
            LANE_TRY(hipEventRecord(e, stream));
            ln.shade_ev.emplace_back(b2, e);
        }
        if (!job.light_overlap)
            if (int rc = launch_light(stream)) return rc;
        ln.shade_launches++;
    }
    if (it >= max_iters) {
        job.abort.store(true);
        return lane_fail(ln, RT_ERR_HIP, "lane %d did not drain within %llu iterations", lane_id, max_iters);
    }
    LANE_TRY(join_light());
    LANE_TRY(hipGetLastError());
    LANE_TRY(hipStreamSynchronize(stream));
    return RT_OK;
}

static size_t batch_max() {  // camera samples per batch (film staging: 24 B each)
    static const size_t v = [] {
        // 2^30 samples = 25.8 GB of staging when a frame is that large.  A batch larger than the pool (2^28 paths) is refilled
        // while it lasts and drains once: per C4 frame two drains instead of eight -- the iterations that end a batch run the
        // machine on a few million paths.  2^28 / 2^29 / 2^30: C4 4896 / 4932 / 4960 Mrays/s, C5 4967 / 4952 / 5050
        // (profiles/r04_sweep_batch.txt)
        size_t lg = 30;
        if (const char* e = getenv("RT_BATCH_LOG2")) lg = (size_t)std::min(31, std::max(16, atoi(e)));
        return (size_t)1 << lg;
    }();
    return v;
}
#define kBatchMax (batch_max())

// sub_rank / sub_world: device `sub_rank` of a several-device context takes every sub_world-th of the caller's tiles
static int render_impl(rt_context* c, rt_scene* s, const rt_camera* cam, const rt_render_cfg* cfg, double* d_rgb,
                       uint32_t* d_n, hipStream_t stream, rt_stats* stats, uint32_t sub_rank = 0, uint32_t sub_world = 1) {
    const uint32_t W = cfg->width, H = cfg->height;
    const uint32_t spp = next_pow2(cfg->spp);
    uint32_t x0 = cfg->x0, y0 = cfg->y0, x1 = cfg->x1, y1 = cfg->y1;
    if (x1 == 0 && y1 == 0) {
        x0 = 0; y0 = 0; x1 = W; y1 = H;
    }
    const uint32_t ts = cfg->tile_size ? cfg->tile_size : 16;
    const uint32_t world = cfg->tile_world ? cfg->tile_world : 1;
    const uint32_t rank = cfg->tile_rank;
    // owned pixels: the tiles rt_tile_owner() gives this rank, in row-major order; 16-wide rows inside a tile
    std::vector<uint32_t> pix;
    const uint32_t tw = (W + ts - 1) / ts, th = (H + ts - 1) / ts;
    uint32_t n_own = 0;
    for (uint32_t k = 0; k < tw * th; k++) {
        const uint32_t tx = k % tw, ty = k / tw;
        if (rtabi_tile_owner(tx, ty, world) != rank) continue;
        if (n_own++ % sub_world != sub_rank) continue;
        for (uint32_t y = 0; y < ts; y++)
            for (uint32_t x = 0; x < ts; x++) {
                const uint32_t px = tx * ts + x, py = ty * ts + y;
                if (px < x0 || px >= x1 || py < y0 || py >= y1) continue;
                pix.push_back(py * W + px);
            }
    }
    HIP_TRY(hipSetDevice(c->device));
    c->last_pix = pix;
    if (!(cfg->flags & RT_RENDER_ACCUMULATE)) {
        HIP_TRY(hipMemsetAsync(d_rgb, 0, sizeof(double) * 3 * (size_t)W * H, stream));
        HIP_TRY(hipMemsetAsync(d_n, 0, sizeof(uint32_t) * (size_t)W * H, stream));
    }
    // progressive pass: samples [s_first, s_end) of every pixel
    const uint32_t s_first = std::min(cfg->sample_first, spp);
    const uint32_t s_end = cfg->sample_count ? (uint32_t)std::min<uint64_t>((uint64_t)s_first + cfg->sample_count, spp) : spp;
    const uint32_t pass_spp = s_end - s_first;
    HIP_TRY(hipMemsetAsync(c->stats, 0, sizeof(DevStats) * kStatShards, stream));
    const size_t NP = pix.size();
    if (NP > 0) {
        // The device copy of the pixel list is refreshed whenever this rank owns pixels -- also for an empty progressive
        // pass (sample_first >= spp): a multi-device context's gather packs every peer's film through it (render_multi).
        if (c->pix_capacity < NP) {
            if (c->pix_list) HIP_TRY(hipFree(c->pix_list));
            c->pix_list = nullptr;
            c->pix_capacity = 0;
            HIP_TRY(hipMalloc((void**)&c->pix_list, NP * sizeof(uint32_t)));
            c->pix_capacity = NP;
        }
        // (pix is a local: the copy has to finish before it goes out of scope on the early-return paths)
        HIP_TRY(hipMemcpyAsync(c->pix_list, c->last_pix.data(), NP * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    }
    double kernel_ms = 0.0, trace_ms = 0.0, shade_ms = 0.0, classify_ms = 0.0, light_ms = 0.0;
    uint64_t trace_launches = 0, shade_launches = 0;
    if (NP > 0 && pass_spp > 0) {
        // batch shape: PB pixels x ns samples, at most kBatchMax camera samples (film staging size)
        uint32_t PB, ns;
        if (NP * (size_t)pass_spp <= kBatchMax) {
            PB = (uint32_t)NP;
            ns = pass_spp;
        } else if (NP <= kBatchMax) {
            PB = (uint32_t)NP;
            ns = 1;
            while ((size_t)PB * ns * 2 <= kBatchMax && ns * 2 <= pass_spp) ns *= 2;
        } else {
            PB = (uint32_t)kBatchMax;
            ns = 1;
        }
        const size_t batch_cap = (size_t)PB * ns;
        // pool per lane
        // (the film staging first: what the pool's default may take is what is left after it)
        if (c->lf_capacity < batch_cap) {
            if (c->lf) HIP_TRY(hipFree(c->lf));
            c->lf = nullptr;
            c->lf_capacity = 0;
            HIP_TRY(hipMalloc((void**)&c->lf, batch_cap * 3 * sizeof(double)));
            c->lf_capacity = batch_cap;
        }
        // default pool: 256 Mi paths or a whole batch if that is less (0.85 KB per path: 238 GB of the 288 GB).  Fewer, fuller launches: C4 3822 / 4077 / 4105 / 4206 Mrays/s at 16 / 64 / 128 / 256 Mi,
        // C3 3910 / 4090 / 4141 / 4185 (profiles/r03_sweep_pool.txt, r03_sweep_pool_big.txt).  A default that does not leave
        // kPoolHeadroom of the device's free memory to everybody else (the next scene commit, the fast mode's leaf copies, the
        // caller's own buffers, other ranks sharing the GPU), or that fails to allocate, is halved (down to 16 Mi) instead of
        // failing; a size the caller asked for is not.
        const bool pool_default = cfg->paths_in_flight == 0;
        uint32_t P = pool_default ? (1u << 28) : cfg->paths_in_flight;
        P = std::max<uint32_t>(64u, std::min<uint32_t>(P, 1u << 28));
        P = (P + 63u) & ~63u;
        int n_lanes = 1;
        for (;;) {
            n_lanes = (batch_cap > (size_t)P) ? c->n_lanes : 1;
            P = (uint32_t)std::min<size_t>(P, (batch_cap + 63) & ~(size_t)63);
            int rc = RT_OK;
            const uint32_t floor_paths = test_pool_oom_above() ? 64u : (1u << 24);
            if (pool_default && P > floor_paths && c->lanes[0].capacity < P) {
                constexpr size_t kPoolHeadroom = (size_t)12 << 30;
                size_t fr = 0, tot = 0, slots, qn, bytes;
                pool_layout(P, s->n_cls, slots, qn, bytes);
                size_t held = 0;  // (a smaller pool of an earlier render is released first)
                for (int i = 0; i < n_lanes; i++) held += c->lanes[i].pool_bytes;
                if (hipMemGetInfo(&fr, &tot) == hipSuccess && bytes * (size_t)n_lanes + kPoolHeadroom > fr + held) {
                    P >>= 1;
                    continue;
                }
            }
            for (int i = 0; i < n_lanes && rc == RT_OK; i++) rc = ensure_lane_capacity(c, c->lanes[i], P, s->n_cls);
            if (rc == RT_OK) break;
            if (rc != RT_ERR_OOM || !pool_default || P <= floor_paths) return rc;
            (void)hipGetLastError();
            P >>= 1;
        }
        RenderJob job;
        job.c = c;
        job.s = s;
        job.cam = *cam;
        job.cfg = cfg;
        job.pool = P;
        job.count_trav = (cfg->flags & RT_RENDER_COUNT_TRAVERSAL) != 0;
        // persistent grid = what is resident at once (more blocks would only queue behind them)
        job.f32 = cfg->precision == RT_PRECISION_F32;
        job.f32_gen = job.f32_trace = job.f32_shade = job.f32_tail = job.f32;
        if (job.f32) {
            const int frc = ensure_f32_leaves(s, stream);
            if (frc != RT_OK) return frc;
        }
        if (const char* e = getenv("RT_F32_MIX")) {  // debug: e.g. "t" = only the traversal kernel in f32
            job.f32_gen = job.f32 && strchr(e, 'g');
            job.f32_trace = job.f32 && strchr(e, 't');
            job.f32_shade = job.f32 && strchr(e, 's');
            job.f32_tail = job.f32 && strchr(e, 'l');
        }
        int occ = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, trace_kernel(job.count_trav, s->dev.simple_others != 0, job.f32), 256, 0));
        int per_cu = std::max(1, occ);
        // the light kernel beside the class kernels and the next traversal launch, one block per CU (128 VGPRs: a class
        // kernel wave or a traversal block fewer on that SIMD while it is there): C4 4624 -> 4692 Mrays/s, C3 4226 -> 4395,
        // C2 3809 -> 3929; with its full grid it only swaps places with the class kernels (profiles/r04_exp_light_overlap.txt)
        job.light_overlap = true;
        if (const char* e = getenv("RT_LIGHT_OVERLAP")) job.light_overlap = atoi(e) != 0;
        if (const char* e = getenv("RT_CLS_STREAMS")) job.cls_streams = std::min(3, std::max(1, atoi(e)));
        if (const char* e = getenv("RT_TRACE_BLOCKS_PER_CU")) per_cu = std::max(1, atoi(e));
        job.trace_blocks = c->num_cus * per_cu;
        for (uint32_t k = 0; k < s->n_cls; k++) {
            int so = 0;
            if (k == 0)
                HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&so, shade_light_kernel(s->dev.env.light >= 0, job.f32), 256, 0));
            else
                HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&so, shade_cls_kernel(s->cls[k], job.f32, s->all_lambert), 256, 0));
            job.shade_blocks[k] = c->num_cus * std::max(1, so);
            if (k == 0) {
                if (job.light_overlap) job.shade_blocks[0] = c->num_cus;
                if (const char* e = getenv("RT_LIGHT_BLOCKS_PER_CU")) job.shade_blocks[0] = c->num_cus * std::max(1, atoi(e));
            }
        }
        size_t ev_i = 0;
        hipEvent_t ev_begin = get_event(c->events, ev_i++), ev_end = get_event(c->events, ev_i++);
        hipEvent_t ev_ready = get_event(c->events, ev_i++);
        if (!ev_begin || !ev_end || !ev_ready) return fail(RT_ERR_HIP, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(ev_begin, stream));
        for (int i = 0; i < n_lanes; i++) {
            Lane& ln = c->lanes[i];
            ln.ev_i = 0;
            ln.sync_i = 0;
            ln.trace_ev.clear();
            ln.classify_ev.clear();
            ln.shade_ev.clear();
            ln.light_ev.clear();
            ln.trace_launches = 0;
            ln.shade_launches = 0;
        }
        for (size_t pb = 0; pb < NP; pb += PB) {
            const uint32_t npx = (uint32_t)std::min<size_t>(PB, NP - pb);
            for (uint32_t sb = s_first; sb < s_end; sb += ns) {
                ChunkDesc& ck = job.batch;
                ck.n_pixels = npx;
                ck.n_samples = std::min(ns, s_end - sb);
                ck.pixel_base = (uint32_t)pb;
                ck.sample_base = sb;
                ck.width = W;
                ck.height = H;
                ck.seed = cfg->seed;
                job.batch_total = (unsigned long long)npx * ck.n_samples;
                // the lanes start after everything queued on the caller's stream so far (film zeroing,
                // pixel list upload, the previous batch's resolve)
                HIP_TRY(hipMemsetAsync(c->batch, 0, sizeof(BatchCtl), stream));
                HIP_TRY(hipEventRecord(ev_ready, stream));
                for (int i = 0; i < n_lanes; i++) {
                    Lane& ln = c->lanes[i];
                    ln.rc = RT_OK;
                    ln.err[0] = 0;
                    HIP_TRY(hipStreamWaitEvent(ln.stream, ev_ready, 0));
                }
                std::vector<std::thread> workers;
                for (int i = 1; i < n_lanes; i++) workers.emplace_back([&job, i] { run_lane(job, i); });
                run_lane(job, 0);
                for (auto& t : workers) t.join();
                if (job.cancelled.load()) {
                    (void)hipDeviceSynchronize();  // the launches already queued run to completion
                    return fail(RT_ERR_CANCELLED, "rt_render: cancelled by the caller");
                }
                for (int i = 0; i < n_lanes; i++) {
                    Lane& ln = c->lanes[i];
                    if (ln.rc != RT_OK) {
                        (void)hipDeviceSynchronize();
                        return fail(ln.rc, "%s", ln.err);
                    }
                }
                // every lane has drained (run_lane synchronises its stream): add the batch to the film
                hipLaunchKernelGGL(rtk::kernel_table().resolve, dim3((npx + 255) / 256), dim3(256), 0, stream, c->lf,
                                   ck, c->pix_list, d_rgb, d_n);
                HIP_TRY(hipGetLastError());
            }
        }
        HIP_TRY(hipEventRecord(ev_end, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev_begin, ev_end));
        kernel_ms = ms;
        for (int i = 0; i < n_lanes; i++) {
            Lane& ln = c->lanes[i];
            for (auto& pr : ln.trace_ev) {
                HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
                trace_ms += ms;
            }
            for (auto& pr : ln.shade_ev) {
                HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
                shade_ms += ms;
            }
            for (auto& pr : ln.classify_ev) {
                HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
                classify_ms += ms;
            }
            for (auto& pr : ln.light_ev) {
                HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
                light_ms += ms;
            }
            trace_launches += ln.trace_launches;
            shade_launches += ln.shade_launches;
        }
    } else {
        HIP_TRY(hipStreamSynchronize(stream));
    }
    if (stats) {
        DevStats shards[kStatShards];
        HIP_TRY(hipMemcpy(shards, c->stats, sizeof(shards), hipMemcpyDeviceToHost));
        DevStats ds{};
        for (int i = 0; i < kStatShards; i++) {
            ds.paths += shards[i].paths; ds.r1 += shards[i].r1; ds.r2 += shards[i].r2; ds.r3 += shards[i].r3;
            ds.vertices += shards[i].vertices; ds.nodes += shards[i].nodes; ds.tris += shards[i].tris;
            ds.others += shards[i].others;
            ds.tail_rays += shards[i].tail_rays; ds.tail_nodes += shards[i].tail_nodes;
            ds.tail_tris += shards[i].tail_tris; ds.tail_others += shards[i].tail_others;
        }
        std::memset(stats, 0, sizeof(*stats));
        stats->paths = ds.paths;
        stats->rays_extension = ds.r1;
        stats->rays_shadow = ds.r2;
        stats->rays_probe = ds.r3;
        stats->vertices_shaded = ds.vertices;
        stats->nodes_fetched = ds.nodes;
        stats->tris_tested = ds.tris;
        stats->others_tested = ds.others;
        stats->kernel_ms = kernel_ms;
        stats->trace_ms = trace_ms;
        stats->trace_launches = trace_launches;
        stats->gather_ms = 0.0;
        stats->n_devices = 1;
        stats->shade_ms = shade_ms;
        stats->shade_launches = shade_launches;
        stats->classify_ms = classify_ms;
        stats->light_ms = light_ms;
        if (getenv("RT_DIAG")) {
            unsigned long long d[4] = {0, 0, 0, 0}, over64 = 0, over256 = 0;
            for (int i = 0; i < kStatShards; i++) {
                for (int q = 0; q < 4; q++) d[q] += shards[i].pad[8 + q];
                over64 += shards[i].pad[4];
                over256 += shards[i].pad[5];
            }
            fprintf(stderr, "[rt diag] node rounds %llu avg lanes %.1f | prim rounds %llu avg lanes %.1f\n", d[0],
                    d[0] ? (double)d[1] / d[0] : 0.0, d[2], d[2] ? (double)d[3] / d[2] : 0.0);
            // instrumented build only: short trace launches (in-kernel time, 10 ns ticks) and the longest traversals
            fprintf(stderr, "[rt diag] trace launches under 4096 rays: %llu, avg in-kernel us %.1f, avg rays %.1f | "
                            "max steps per ray %llu, rays over 64 steps %llu, over 256 steps %llu\n",
                    shards[0].pad[1], shards[0].pad[1] ? shards[0].pad[0] / 100.0 / shards[0].pad[1] : 0.0,
                    shards[0].pad[1] ? (double)shards[0].pad[2] / shards[0].pad[1] : 0.0, shards[0].pad[3], over64, over256);
        }
        stats->tail_rays = ds.tail_rays;
        stats->tail_nodes_fetched = ds.tail_nodes;
        stats->tail_tris_tested = ds.tail_tris;
        stats->tail_others_tested = ds.tail_others;
    }
    return RT_OK;
}

// ------------------------------------------------------------------ multi-device render
// Tile (tx, ty) belongs to rank rt_tile_owner(tx, ty, world) (include/rt_abi.h).  One caller: the context's N devices
// are the lattice's N ranks; a caller that is one of several keeps its own tiles and deals them to its devices in
// turn.  Pixels are independent and keyed
// by (seed, pixel, sample), so the film is bit-identical to a one-device render.  The only exchange is the final
// gather: each further device packs its own pixels ({r, g, b, n} as four doubles), copies them straight to the
// primary device (hipMemcpyPeerAsync: one xGMI link per peer, 1/N of the bytes a full-frame reduce would move) where
// a scatter kernel writes them into the caller's film.
__global__ __launch_bounds__(256) void k_film_pack(const double* __restrict__ rgb, const uint32_t* __restrict__ n,
                                                   const uint32_t* __restrict__ pix_list, uint32_t np, double* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np) return;
    const size_t p = pix_list[i];
    out[(size_t)i * 4 + 0] = rgb[p * 3 + 0];
    out[(size_t)i * 4 + 1] = rgb[p * 3 + 1];
    out[(size_t)i * 4 + 2] = rgb[p * 3 + 2];
    out[(size_t)i * 4 + 3] = (double)n[p];
}
__global__ __launch_bounds__(256) void k_film_unpack(const double* __restrict__ in, const uint32_t* __restrict__ pix_list,
                                                     uint32_t np, double* rgb, uint32_t* n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np) return;
    const size_t p = pix_list[i];
    rgb[p * 3 + 0] = in[(size_t)i * 4 + 0];
    rgb[p * 3 + 1] = in[(size_t)i * 4 + 1];
    rgb[p * 3 + 2] = in[(size_t)i * 4 + 2];
    n[p] = (uint32_t)in[(size_t)i * 4 + 3];
}

extern "C++" {
template <typename T>
static int ensure_dev_buffer(int device, T** buf, size_t* cap, size_t want) {
    if (*cap >= want) return RT_OK;
    HIP_TRY(hipSetDevice(device));
    if (*buf) HIP_TRY(hipFree(*buf));
    *buf = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc((void**)buf, want * sizeof(T)));
    *cap = want;
    return RT_OK;
}
}  // extern "C++"

static int render_multi(rt_context* c, rt_scene* s, const rt_camera* cam, const rt_render_cfg* cfg, double* d_rgb,
                        uint32_t* d_n, hipStream_t stream, rt_stats* stats) {
    const int N = 1 + (int)c->peers.size();
    const size_t npix = (size_t)cfg->width * cfg->height;
    const uint32_t world = cfg->tile_world ? cfg->tile_world : 1;
    if ((uint64_t)world * (uint64_t)N > 0xffffffffull) return fail(RT_ERR_INVALID_ARG, "rt_render: tile_world too large");
    // one caller (the usual case): the context's devices are the lattice's ranks; a caller that is itself one of
    // several (tile_world > 1) keeps its own tiles and deals them to its devices in turn
    std::vector<rt_render_cfg> cfgs((size_t)N, *cfg);
    const uint32_t sub_world = world > 1 ? (uint32_t)N : 1u;
    if (world == 1)
        for (int i = 0; i < N; i++) {
            cfgs[i].tile_world = (uint32_t)N;
            cfgs[i].tile_rank = (uint32_t)i;
        }
    int rc = RT_OK;
    if (s->replicas.size() != c->peers.size()) return fail(RT_ERR_STATE, "rt_render: scene was not created on this multi-device context");
    for (const rt_scene* r : s->replicas)
        if (!r->committed) return fail(RT_ERR_STATE, "rt_render: a device copy of the scene is not committed");
    for (int i = 1; i < N; i++) {
        rt_context* p = c->peers[i - 1];
        if (p->pf_cap < npix) {  // scratch film of the peer
            size_t cap_rgb = 0, cap_n = 0;
            if (p->pf_rgb) (void)hipFree(p->pf_rgb);
            if (p->pf_n) (void)hipFree(p->pf_n);
            p->pf_rgb = nullptr;
            p->pf_n = nullptr;
            p->pf_cap = 0;
            if ((rc = ensure_dev_buffer(p->device, &p->pf_rgb, &cap_rgb, npix * 3)) != RT_OK) return rc;
            if ((rc = ensure_dev_buffer(p->device, &p->pf_n, &cap_n, npix)) != RT_OK) return rc;
            p->pf_cap = npix;
        }
        if (cfg->flags & RT_RENDER_ACCUMULATE) {
            // progressive pass: every device continues from the caller's film, so that a pixel's running sum is
            // added to in the same order as on one device
            HIP_TRY(hipMemcpyPeerAsync(p->pf_rgb, p->device, d_rgb, c->device, npix * 3 * sizeof(double), stream));
            HIP_TRY(hipMemcpyPeerAsync(p->pf_n, p->device, d_n, c->device, npix * sizeof(uint32_t), stream));
        }
    }
    if (cfg->flags & RT_RENDER_ACCUMULATE) HIP_TRY(hipStreamSynchronize(stream));
    std::vector<rt_stats> st((size_t)N);
    std::vector<int> rcs((size_t)N, RT_OK);
    std::vector<std::string> errs((size_t)N);
    std::vector<std::thread> workers;
    for (int i = 1; i < N; i++)
        workers.emplace_back([&, i] {  // one host thread per device
            rt_context* p = c->peers[i - 1];
            rcs[i] = render_impl(p, s->replicas[i - 1], cam, &cfgs[i], p->pf_rgb, p->pf_n, p->stream, &st[i],
                                 sub_world > 1 ? (uint32_t)i : 0u, sub_world);
            if (rcs[i] != RT_OK) errs[i] = g_err;  // g_err is thread-local
        });
    rcs[0] = render_impl(c, s, cam, &cfgs[0], d_rgb, d_n, stream, &st[0], 0u, sub_world);
    if (rcs[0] != RT_OK) errs[0] = g_err;
    for (auto& t : workers) t.join();
    for (int i = 0; i < N; i++)
        if (rcs[i] != RT_OK) return fail(rcs[i], "device %d of the context: %s", i, errs[i].c_str());
    // ---- gather the peers' pixels into the caller's film
    const auto g0 = std::chrono::steady_clock::now();
    size_t total = 0;
    for (int i = 1; i < N; i++) total += c->peers[i - 1]->last_pix.size();
    if ((rc = ensure_dev_buffer(c->device, &c->stage, &c->stage_cap, total * 4)) != RT_OK) return rc;
    if ((rc = ensure_dev_buffer(c->device, &c->stage_pix, &c->stage_pix_cap, total)) != RT_OK) return rc;
    size_t off = 0;
    for (int i = 1; i < N; i++) {
        rt_context* p = c->peers[i - 1];
        const size_t np = p->last_pix.size();
        if (np == 0) continue;
        if ((rc = ensure_dev_buffer(p->device, &p->pack, &p->pack_cap, np * 4)) != RT_OK) return rc;
        HIP_TRY(hipSetDevice(p->device));
        if (!p->ev_gather) HIP_TRY(hipEventCreateWithFlags(&p->ev_gather, hipEventDisableTiming));
        // p->pix_list still holds this render's pixel list on the peer device
        hipLaunchKernelGGL(k_film_pack, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, p->stream, p->pf_rgb, p->pf_n,
                           p->pix_list, (uint32_t)np, p->pack);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyPeerAsync(c->stage + off * 4, c->device, p->pack, p->device, np * 4 * sizeof(double), p->stream));
        HIP_TRY(hipEventRecord(p->ev_gather, p->stream));
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamWaitEvent(stream, p->ev_gather, 0));
        HIP_TRY(hipMemcpyAsync(c->stage_pix + off, p->last_pix.data(), np * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_film_unpack, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, stream, c->stage + off * 4,
                           c->stage_pix + off, (uint32_t)np, d_rgb, d_n);
        HIP_TRY(hipGetLastError());
        off += np;
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(stream));
    const double gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g0).count();
    if (stats) {
        rt_stats t = st[0];
        for (int i = 1; i < N; i++) {
            t.paths += st[i].paths; t.rays_extension += st[i].rays_extension; t.rays_shadow += st[i].rays_shadow;
            t.rays_probe += st[i].rays_probe; t.vertices_shaded += st[i].vertices_shaded;
            t.nodes_fetched += st[i].nodes_fetched; t.tris_tested += st[i].tris_tested;
            t.others_tested += st[i].others_tested; t.trace_ms += st[i].trace_ms;
            t.trace_launches += st[i].trace_launches; t.tail_rays += st[i].tail_rays;
            t.shade_ms += st[i].shade_ms; t.shade_launches += st[i].shade_launches; t.classify_ms += st[i].classify_ms; t.light_ms += st[i].light_ms;
            t.tail_nodes_fetched += st[i].tail_nodes_fetched; t.tail_tris_tested += st[i].tail_tris_tested;
            t.tail_others_tested += st[i].tail_others_tested;
            t.kernel_ms = std::max(t.kernel_ms, st[i].kernel_ms);  // the devices run side by side
        }
        t.gather_ms = gather_ms;
        t.n_devices = (uint64_t)N;
        *stats = t;
    }
    return RT_OK;
}

static int render_any(rt_context* c, rt_scene* s, const rt_camera* cam, const rt_render_cfg* cfg, double* d_rgb,
                      uint32_t* d_n, hipStream_t stream, rt_stats* stats) {
    if (c->peers.empty()) return render_impl(c, s, cam, cfg, d_rgb, d_n, stream, stats);
    return render_multi(c, s, cam, cfg, d_rgb, d_n, stream, stats);
}

static int check_render_args(rt_context* c, rt_scene* s, const rt_camera* cam, const rt_render_cfg* cfg) {
    if (!c || !s || !cam || !cfg) return fail(RT_ERR_INVALID_ARG, "rt_render: null argument");
    if (s->ctx != c) return fail(RT_ERR_INVALID_ARG, "rt_render: scene belongs to another context");
    if (!s->committed) return fail(RT_ERR_STATE, "rt_render: scene is not committed");
    if (cfg->width == 0 || cfg->height == 0 || cfg->spp == 0) return fail(RT_ERR_INVALID_ARG, "rt_render: empty image or spp == 0");
    if ((uint64_t)cfg->width * cfg->height >= (1ull << 30)) return fail(RT_ERR_UNSUPPORTED, "rt_render: image too large");
    if (cfg->max_depth > 255) return fail(RT_ERR_UNSUPPORTED, "rt_render: max_depth > 255");
    if (cfg->precision != RT_PRECISION_F64 && cfg->precision != RT_PRECISION_F32)
        return fail(RT_ERR_UNSUPPORTED, "rt_render: unknown precision");
    if (!(cfg->x1 == 0 && cfg->y1 == 0) &&
        (cfg->x1 > cfg->width || cfg->y1 > cfg->height || cfg->x0 > cfg->x1 || cfg->y0 > cfg->y1))
        return fail(RT_ERR_INVALID_ARG, "rt_render: bad pixel window");
    const uint32_t world = cfg->tile_world ? cfg->tile_world : 1;
    if (cfg->tile_rank >= world) return fail(RT_ERR_INVALID_ARG, "rt_render: tile_rank >= tile_world");
    return RT_OK;
}

int rt_render_device(rt_context* c, rt_scene* s, const rt_camera* cam, const rt_render_cfg* cfg, double* d_rgb_sum,
                     uint32_t* d_n, void* hip_stream, rt_stats* stats) {
    int rc = check_render_args(c, s, cam, cfg);
    if (rc != RT_OK) return rc;
    if (!d_rgb_sum || !d_n) return fail(RT_ERR_INVALID_ARG, "rt_render_device: null film pointer");
    return render_any(c, s, cam, cfg, d_rgb_sum, d_n, (hipStream_t)hip_stream, stats);  // NULL = the null stream
}

int rt_render(rt_context* c, rt_scene* s, const rt_camera* cam, const rt_render_cfg* cfg, double* rgb_sum, uint32_t* n,
              rt_stats* stats) {
    int rc = check_render_args(c, s, cam, cfg);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    const size_t npix = (size_t)cfg->width * cfg->height;
    double* d_rgb = nullptr;
    uint32_t* d_n = nullptr;
    HIP_TRY(hipMalloc((void**)&d_rgb, npix * 3 * sizeof(double)));
    hipError_t e = hipMalloc((void**)&d_n, npix * sizeof(uint32_t));
    if (e != hipSuccess) {
        (void)hipFree(d_rgb);
        return fail(RT_ERR_OOM, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    if (cfg->flags & RT_RENDER_ACCUMULATE) {  // progressive pass: continue from the caller's film
        if (!rgb_sum || !n) {
            (void)hipFree(d_rgb);
            (void)hipFree(d_n);
            return fail(RT_ERR_INVALID_ARG, "RT_RENDER_ACCUMULATE needs both film pointers");
        }
        e = hipMemcpy(d_rgb, rgb_sum, npix * 3 * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_n, n, npix * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(d_rgb);
            (void)hipFree(d_n);
            return fail(RT_ERR_HIP, "film upload failed: %s", hipGetErrorString(e));
        }
    }
    rc = render_any(c, s, cam, cfg, d_rgb, d_n, c->stream, stats);
    if (rc == RT_OK && rgb_sum) {
        e = hipMemcpy(rgb_sum, d_rgb, npix * 3 * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RT_ERR_HIP, "film download failed: %s", hipGetErrorString(e));
    }
    if (rc == RT_OK && n) {
        e = hipMemcpy(n, d_n, npix * sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RT_ERR_HIP, "film download failed: %s", hipGetErrorString(e));
    }
    (void)hipFree(d_rgb);
    (void)hipFree(d_n);
    return rc;
}

// rt_intersect_batch_ex with RT_INTERSECT_WAVEFRONT: the caller's rays go through the RENDER's traversal kernel
// (k_trace: persistent waves, queue reservations, refill, while-while scheduling, deferred write-back) instead of the
// run-to-completion loop of k_intersect_batch -- a kernel-level parity entry for exactly the code the films depend on.
__global__ __launch_bounds__(256) void k_wf_setup(const rt_ray* __restrict__ rays, uint32_t n, PathState st, uint32_t* queue,
                                                  Ctl* ctl, int f32) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        ctl->n_rays[0] = n;
        ctl->n_active[0] = n;
        ctl->head[0] = 0;
        for (int x = 0; x < 8; x++) ctl->xhead[0][x][0] = 0;
    }
    if (i >= n) return;
    const rt_ray r = rays[i];
    // (the fast mode keeps binary32 values in the low half of each element of the ray arrays, kernels.hip: ld3 / st3)
    auto put = [&](double* a, double v) {
        reinterpret_cast<unsigned long long*>(a)[i] = f32 ? (unsigned long long)__float_as_uint((float)v) : (unsigned long long)__double_as_longlong(v);
    };
    put(st.ox, r.origin[0]); put(st.oy, r.origin[1]); put(st.oz, r.origin[2]);
    // the three ray kinds of the render read their direction from different fields: spread the rays over two of them
    // (an extension ray and a probe ray are both traced on [SMALL, inf) -- a shadow ray is not a free-form ray)
    if (i & 1u) {
        put(st.pdx, r.dir[0]); put(st.pdy, r.dir[1]); put(st.pdz, r.dir[2]);
        queue[i] = i | (kRayProbe << 30);
    } else {
        put(st.dx, r.dir[0]); put(st.dy, r.dir[1]); put(st.dz, r.dir[2]);
        queue[i] = i | (kRayExt << 30);
    }
    st.pr_prim[i] = -2;
}
__global__ __launch_bounds__(256) void k_wf_collect(DevScene sc, const rt_ray* __restrict__ rays, uint32_t n, PathState st,
                                                    const uint32_t* __restrict__ hitw, rt_hit* hits) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const rt_ray r = rays[i];
    // (the queue position of ray i is i: an extension ray's hit word is hitw[i] -- the leaf slot of a triangle, the
    // primitive index of a sphere / rect, 0 = miss; a probe ray's result is the primitive itself)
    int32_t prim;
    if (i & 1u) {
        prim = st.pr_prim[i];
    } else {
        const uint32_t hw = hitw[i];
        prim = hw == 0u ? -1 : (int32_t)((hw & kLeafOther) ? (hw & kIdxMask) : (sc.leaf_prim[hw & kIdxMask] & kIdxMask));
    }
    rt_hit h;
    h.prim = prim;
    h.t = kInf;
    h.reserved = 0;
    if (prim >= 0) {  // the winner's own test gives t (the same arithmetic as the traversal's)
        HitRec rec;
        if (prim_intersects(sc, prim, d3(r.origin[0], r.origin[1], r.origin[2]), d3(r.dir[0], r.dir[1], r.dir[2]), r.tmin, r.tmax, rec))
            h.t = rec.t;
    }
    hits[i] = h;
}

int rt_intersect_batch(rt_context* c, rt_scene* s, const rt_ray* rays, uint64_t n, rt_hit* hits) {
    return rt_intersect_batch_ex(c, s, rays, n, hits, 0u);
}

int rt_intersect_batch_ex(rt_context* c, rt_scene* s, const rt_ray* rays, uint64_t n, rt_hit* hits, uint32_t flags) {
    if (!c || !s || (n && (!rays || !hits))) return fail(RT_ERR_INVALID_ARG, "rt_intersect_batch: null argument");
    if (!s->committed) return fail(RT_ERR_STATE, "rt_intersect_batch: scene is not committed");
    if (flags & ~(uint32_t)(RT_INTERSECT_F32 | RT_INTERSECT_WAVEFRONT)) return fail(RT_ERR_INVALID_ARG, "rt_intersect_batch_ex: unknown flags 0x%x", flags);
    if (n == 0) return RT_OK;
    const bool f32 = (flags & RT_INTERSECT_F32) != 0, wavefront = (flags & RT_INTERSECT_WAVEFRONT) != 0;
    if (wavefront) {
        if (n > (1u << 26)) return fail(RT_ERR_UNSUPPORTED, "rt_intersect_batch_ex: more than 2^26 rays in wavefront mode");
        for (uint64_t i = 0; i < n; i++)
            if (rays[i].tmin != RT_SMALL || rays[i].tmax != RT_INFINITY)
                return fail(RT_ERR_UNSUPPORTED, "rt_intersect_batch_ex: wavefront mode traces on [RT_SMALL, RT_INFINITY) only (ray %llu)",
                            (unsigned long long)i);
    }
    HIP_TRY(hipSetDevice(c->device));
    if (f32) {
        const int frc = ensure_f32_leaves(s, c->stream);
        if (frc != RT_OK) return frc;
    }
    rt_ray* d_rays = nullptr;
    rt_hit* d_hits = nullptr;
    HIP_TRY(hipMalloc((void**)&d_rays, n * sizeof(rt_ray)));
    hipError_t e = hipMalloc((void**)&d_hits, n * sizeof(rt_hit));
    if (e != hipSuccess) {
        (void)hipFree(d_rays);
        return fail(RT_ERR_OOM, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    int rc = RT_OK;
    e = hipMemcpyAsync(d_rays, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && wavefront) {
        Lane& ln = c->lanes[0];
        rc = ensure_lane_capacity(c, ln, (uint32_t)((n + 63) & ~(uint64_t)63), 1u);
        if (rc == RT_OK) {
            const unsigned blocks = (unsigned)((n + 255) / 256);
            int occ = 0;
            const TraceKernel tk = trace_kernel(false, s->dev.simple_others != 0, f32);
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, tk, 256, 0);
            const unsigned tblocks = std::max(1u, std::min(blocks, (unsigned)(c->num_cus * std::max(1, occ))));
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_wf_setup, dim3(blocks), dim3(256), 0, c->stream, d_rays, (uint32_t)n, ln.st[0], ln.queue[0], ln.ctl, f32 ? 1 : 0);
                hipLaunchKernelGGL(tk, dim3(tblocks), dim3(256), 0, c->stream, s->dev, ln.st[0], ln.queue[0], ln.ctl, 0u, c->stats,
                                   c->tune, (MirrorEntry*)nullptr, 0u, c->batch, 0ull, ln.hitw);
                hipLaunchKernelGGL(k_wf_collect, dim3(blocks), dim3(256), 0, c->stream, s->dev, d_rays, (uint32_t)n, ln.st[0], ln.hitw, d_hits);
                e = hipGetLastError();
            }
        }
    } else if (e == hipSuccess) {
        hipLaunchKernelGGL(rtk::kernel_table().intersect[f32 ? 1 : 0], dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, s->dev, d_rays, n, d_hits);
        e = hipGetLastError();
    }
    if (rc == RT_OK && e == hipSuccess) e = hipMemcpyAsync(hits, d_hits, n * sizeof(rt_hit), hipMemcpyDeviceToHost, c->stream);
    if (rc == RT_OK && e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (rc == RT_OK && e != hipSuccess) rc = fail(RT_ERR_HIP, "rt_intersect_batch: %s", hipGetErrorString(e));
    (void)hipDeviceSynchronize();
    (void)hipFree(d_rays);
    (void)hipFree(d_hits);
    return rc;
}

int rt_resolve_rgb8(rt_context* c, const double* rgb_sum, const uint32_t* n, uint32_t width, uint32_t height,
                    uint8_t* rgb8) {
    if (!c || !rgb_sum || !n || !rgb8 || width == 0 || height == 0) return fail(RT_ERR_INVALID_ARG, "rt_resolve_rgb8: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t npix = (size_t)width * height;
    double* d_rgb = nullptr;
    uint32_t* d_n = nullptr;
    uint8_t* d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_rgb, npix * 3 * sizeof(double)));
    hipError_t e = hipMalloc((void**)&d_n, npix * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, npix * 3);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rgb, rgb_sum, npix * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_n, n, npix * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rtk::kernel_table().tonemap, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, c->stream, d_rgb, d_n, (uint64_t)npix, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(rgb8, d_out, npix * 3, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_rgb);
    if (d_n) (void)hipFree(d_n);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "rt_resolve_rgb8: %s", hipGetErrorString(e));
    return RT_OK;
}

}  // extern "C"

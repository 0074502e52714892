// kernels.hip -- the wavefront path-tracing kernels for gfx950 (wave64).
//
// A "batch" = up to kBatchMax camera samples of one block of pixels (the whole render when it
// fits).  A lane keeps up to `paths_in_flight` paths alive and runs, per iteration:
//   k_plan + k_generate  top the lane's pool up with the batch's next camera samples (streaming
//            regeneration: every launch stays full until the batch's single final drain)
//   k_trace  persistent waves pull 64-ray batches from the ray queue (one atomic per
//            wave) and run the closest-hit traversal for the three root call sites
//            of SURVEY.md 3.2: R1 extension (integrator.rs:388), R2 shadow
//            (hittable.rs:25-39, Q13: closest hit, compared by prim index), R3 MIS
//            probe (integrator.rs:615).  A finished extension ray leaves its hit word (primitive + the VERTEX CLASS
//            of that primitive, scene_dev.h) at its queue position.
//   k_classify_count / _scan / _scatter  a counting sort of the queue by vertex class (escaped / mesh hit or sphere-rect
//            hit of a material group): one list of 8-B entries per class, in queue order, no atomics.
//   k_shade_light + one k_shade_cls per class  one lane per path of the class: folds the previous
//            vertex's direct-light terms using the R2/R3 results, rebuilds the winning hit's record,
//            applies the emitted-light rule, builds the BSDF, samples one light + MIS
//            (integrator.rs:530-659), samples the continuation, Russian roulette
//            (integrator.rs:375-445), and writes survivors and their rays to the next pool / queue.
//            Every wave of a class kernel runs ONE class over the whole launch (round 4; rounds 1-3 dealt
//            a block's 256 consecutive slots to its waves by class).  The host (abi.hip: run_lane) runs the class
//            kernels of a bounce two at a time on two streams and the light kernel on a third, beside them and the
//            next k_trace launch; waves take runs of 64-path groups from a cursor per list.
// Path state (scene_dev.h): ray arrays for the traversal kernel + one 256-B record per slot for the shading kernels, two
// pools that ping-pong per bounce: a class kernel reads record `slot` of X[it&1] and writes the survivor to a freshly
// allocated slot of X[(it+1)&1].  Records move as whole 128-B lines, eight lanes per line, through LDS.
// Waves are independent: output slots, queue entries and list entries come from per-wave CHUNKS of the
// shared counters (one scalar atomic per chunk, no block barrier anywhere); the unused end of a wave's last
// chunk is filled with null entries.  A retired path drops its radiance into lfinal[orig].
// The R2/R3 results never influence control flow or RNG draws of the path, only
// additions into L, which is why they can be traced one iteration late.
// Film: k_resolve sums each pixel's samples in sample order in f64 -- the order of
// util::increment_color (util.rs:208-232) -- so images are bit-reproducible and
// independent of how tiles are split over GPUs.
#include "shading.h"

#ifndef RT_XCD_QUEUE
#define RT_XCD_QUEUE 1  // the ray queue in eight parts, one per XCD (k_trace); 0 = one head for all waves
#endif

namespace rtd {

// Ctl, BatchCtl, MirrorEntry, ChunkDesc, TraceTune, Lists, kRing: scene_dev.h (shared with the f32 kernels)

// ---- path records (scene_dev.h: PathState).  Everything is moved as 8-byte words / aligned 16-byte pairs; the fast
// mode keeps binary32 values in the low half of a word (no v_cvt per field and bounce; the launch schedule and the
// pools are shared with the parity mode).
typedef unsigned long long rt_w;
typedef rt_w rt_w2 __attribute__((ext_vector_type(2)));
#ifdef RT_F32
RTD double w2r(rt_w w) { return __uint_as_float((uint32_t)w); }
RTD rt_w r2w(double v) { return (rt_w)__float_as_uint(v); }
#else
RTD double w2r(rt_w w) { return __longlong_as_double((long long)w); }
RTD rt_w r2w(double v) { return (rt_w)__double_as_longlong(v); }
#endif
// the ray arrays (one f64-sized element per slot; fast mode: the low half)
// The ray / light-term words of a pool are fifteen arrays of ONE slab, equally spaced, in the order of scene_dev.h: PathState
// (abi.hip: ensure_lane_capacity checks it): word k of slot i is ox[k * stride + i] -- one base and one stride in SGPRs
// instead of up to fifteen pointers per pool (the class kernels take two pools).
constexpr int kRayO = 0, kRayD = 3, kRaySp = 6, kRayPd = 9, kRayFa = 12;
RTD D3 ldr(const PathState& st, int k, uint32_t i) {
    const size_t rs = (size_t)(st.oy - st.ox);
    const rt_w* b = reinterpret_cast<const rt_w*>(st.ox) + (size_t)k * rs + i;
    return d3(w2r(b[0]), w2r(b[rs]), w2r(b[2 * rs]));
}
RTD void str(const PathState& st, int k, uint32_t i, D3 v) {
    const size_t rs = (size_t)(st.oy - st.ox);
    rt_w* b = reinterpret_cast<rt_w*>(st.ox) + (size_t)k * rs + i;
    b[0] = r2w(v.x);
    b[rs] = r2w(v.y);
    b[2 * rs] = r2w(v.z);
}
RTD rt_w* rec_words(const PathState& st, uint32_t slot) { return reinterpret_cast<rt_w*>(st.rec + (size_t)slot * kRecBytes); }
// a vec3 at word W: one aligned pair and one word, whichever way round W's parity puts them
template <int W>
RTD D3 ld3w(const rt_w* r) {
    if (W & 1) {
        const rt_w a = r[W];
        const rt_w2 b = *reinterpret_cast<const rt_w2*>(r + W + 1);
        return d3(w2r(a), w2r(b.x), w2r(b.y));
    }
    const rt_w2 a = *reinterpret_cast<const rt_w2*>(r + W);
    const rt_w b = r[W + 2];
    return d3(w2r(a.x), w2r(a.y), w2r(b));
}
template <int W>
RTD void st3w(rt_w* r, D3 v) {
    rt_w2 q;
    if (W & 1) {
        r[W] = r2w(v.x);
        q.x = r2w(v.y);
        q.y = r2w(v.z);
        *reinterpret_cast<rt_w2*>(r + W + 1) = q;
    } else {
        q.x = r2w(v.x);
        q.y = r2w(v.y);
        *reinterpret_cast<rt_w2*>(r + W) = q;
        r[W + 2] = r2w(v.z);
    }
}
RTD void st_meta(rt_w* r, uint64_t rng, uint32_t orig, uint32_t flags) {
    rt_w2 q;
    q.x = rng;
    q.y = (rt_w)orig | ((rt_w)flags << 32);
    *reinterpret_cast<rt_w2*>(r + kWRng) = q;
}
// film staging: the radiance of a retired path, three consecutive doubles at its (pixel, sample) slot (one partly written
// HBM atom per path; rounds 1-3 had three arrays, whose writes came in path order -- the class lists scramble it)
RTD void film_put(f64_t* lf, uint32_t og, D3 v) {
    f64_t* q = lf + (size_t)og * 3;
    q[0] = v.x;
    q[1] = v.y;
    q[2] = v.z;
}
// w8-15: beta, L and the zeroed result words
RTD void st_beta_l(rt_w* r, D3 beta, D3 l) {
    rt_w2 q;
    q.x = r2w(beta.x); q.y = r2w(beta.y);
    *reinterpret_cast<rt_w2*>(r + 8) = q;
    q.x = r2w(beta.z); q.y = r2w(l.x);
    *reinterpret_cast<rt_w2*>(r + 10) = q;
    q.x = r2w(l.y); q.y = r2w(l.z);
    *reinterpret_cast<rt_w2*>(r + 12) = q;
    q.x = 0ull; q.y = 0ull;
    *reinterpret_cast<rt_w2*>(r + 14) = q;
}
// ---- records move as WHOLE LINES, eight lanes per 128-B line.  A lane that touches its own record with eight 16-B
// loads or stores makes eight L2 requests for one line, and the chip serves 50-75 G requests a second whatever their
// size (tools/ubench_partial_write.hip: own-line 128-B writes 9.5 G lines/s, eight lanes per line 37.9 G lines/s).  So a
// wave's 64 lines go through LDS: every lane deposits (or collects) its line there, pitch 9 pairs = conflict-free, and
// the wave moves the lines between LDS and HBM together.  All 64 lanes must call stage_flush / stage_fetch (wave-uniform
// control flow); a slot of kNullEntry at a rank = no line there.
constexpr int kStagePitch = 9;                      // pairs per staged line (8 + 1 pad)
constexpr int kStageWave = 64 * kStagePitch + 16;   // pairs per wave: 64 lines + their slots (64 x 4 B)
RTD uint32_t* stage_slots(rt_w2* s) { return reinterpret_cast<uint32_t*>(s + 64 * kStagePitch); }
RTD void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// line 0 of a record: o, d, rng, {orig, flags}, beta, L, two spare words
RTD void stage_line0(rt_w2* s, uint32_t rank, uint32_t slot, D3 o, D3 d, uint64_t rng, uint32_t orig, uint32_t flags, D3 beta, D3 l) {
    rt_w2* q = s + rank * kStagePitch;
    rt_w2 v;
    v.x = r2w(o.x); v.y = r2w(o.y); q[0] = v;
    v.x = r2w(o.z); v.y = r2w(d.x); q[1] = v;
    v.x = r2w(d.y); v.y = r2w(d.z); q[2] = v;
    v.x = rng; v.y = (rt_w)orig | ((rt_w)flags << 32); q[3] = v;
    v.x = r2w(beta.x); v.y = r2w(beta.y); q[4] = v;
    v.x = r2w(beta.z); v.y = r2w(l.x); q[5] = v;
    v.x = r2w(l.y); v.y = r2w(l.z); q[6] = v;
    v.x = 0ull; v.y = 0ull; q[7] = v;
    stage_slots(s)[rank] = slot;
}
// line 1: the pending direct-light terms A, Q, K
RTD void stage_line1(rt_w2* s, uint32_t rank, uint32_t slot, D3 a, D3 q3, D3 k) {
    rt_w2* q = s + rank * kStagePitch;
    rt_w2 v;
    v.x = r2w(a.x); v.y = r2w(a.y); q[0] = v;
    v.x = r2w(a.z); v.y = r2w(q3.x); q[1] = v;
    v.x = r2w(q3.y); v.y = r2w(q3.z); q[2] = v;
    v.x = r2w(k.x); v.y = r2w(k.y); q[3] = v;
    v.x = r2w(k.z); v.y = 0ull; q[4] = v;
    v.x = 0ull; q[5] = v; q[6] = v; q[7] = v;
    stage_slots(s)[rank] = slot;
}
// lines 0 .. cnt-1 of the wave's staging area go to line `line` of their records (eight rounds of eight lines, written out
// so that the LDS reads of all rounds are in flight together -- as a loop every round waited for its own two LDS reads)
RTD void stage_flush(rt_w2* s, const PathState& st, uint32_t cnt, uint32_t line) {
    wave_sync_lds();
    const uint32_t lane = threadIdx.x & 63u, part = lane & 7u, sub = lane >> 3;
    const uint32_t* slots = stage_slots(s);
    char* base = st.rec + (size_t)(line * 8u + part) * 16u;
    const rt_w2* row = s + sub * kStagePitch + part;
#define RT_FLUSH_LD(k) const uint32_t s##k = (k) * 8u + sub < cnt ? slots[(k) * 8 + sub] : kNullEntry; const rt_w2 v##k = row[(k) * 8 * kStagePitch];
    RT_FLUSH_LD(0) RT_FLUSH_LD(1) RT_FLUSH_LD(2) RT_FLUSH_LD(3) RT_FLUSH_LD(4) RT_FLUSH_LD(5) RT_FLUSH_LD(6) RT_FLUSH_LD(7)
#undef RT_FLUSH_LD
#define RT_FLUSH_ST(k) if (s##k != kNullEntry) *reinterpret_cast<rt_w2*>(base + (size_t)s##k * kRecBytes) = v##k;
    RT_FLUSH_ST(0) RT_FLUSH_ST(1) RT_FLUSH_ST(2) RT_FLUSH_ST(3) RT_FLUSH_ST(4) RT_FLUSH_ST(5) RT_FLUSH_ST(6) RT_FLUSH_ST(7)
#undef RT_FLUSH_ST
    wave_sync_lds();  // (the area is reused)
}
// the reverse: line `line` of record `slot` (kNullEntry: none) of every lane arrives at that lane's rank = lane
RTD void stage_fetch(rt_w2* s, const PathState& st, uint32_t slot, uint32_t line) {
    const uint32_t lane = threadIdx.x & 63u, part = lane & 7u, sub = lane >> 3;
    uint32_t* slots = stage_slots(s);
    slots[lane] = slot;
    wave_sync_lds();
    // eight rounds of eight lines; the loads of all rounds are in flight together.  (Written out with scalars: the same
    // code with small arrays and unrolled loops made the register allocator spill 1500 values in the class kernels.)
    const char* base = st.rec + (size_t)(line * 8u + part) * 16u;
    const uint32_t s0 = slots[sub], s1 = slots[8 + sub], s2 = slots[16 + sub], s3 = slots[24 + sub];
    const uint32_t s4 = slots[32 + sub], s5 = slots[40 + sub], s6 = slots[48 + sub], s7 = slots[56 + sub];
    rt_w2 z;
    z.x = 0ull;
    z.y = 0ull;
#define RT_FETCH(v, sj) const rt_w2 v = sj != kNullEntry ? *reinterpret_cast<const rt_w2*>(base + (size_t)sj * kRecBytes) : z;
    RT_FETCH(v0, s0) RT_FETCH(v1, s1) RT_FETCH(v2, s2) RT_FETCH(v3, s3) RT_FETCH(v4, s4) RT_FETCH(v5, s5) RT_FETCH(v6, s6) RT_FETCH(v7, s7)
#undef RT_FETCH
    rt_w2* row = s + sub * kStagePitch + part;
    row[0 * 8 * kStagePitch] = v0;
    row[1 * 8 * kStagePitch] = v1;
    row[2 * 8 * kStagePitch] = v2;
    row[3 * 8 * kStagePitch] = v3;
    row[4 * 8 * kStagePitch] = v4;
    row[5 * 8 * kStagePitch] = v5;
    row[6 * 8 * kStagePitch] = v6;
    row[7 * 8 * kStagePitch] = v7;
    wave_sync_lds();
}
// a record's pairs in registers (the fields a kernel wants are fetched up front, independent of each other)
struct RecRegs {
    rt_w2 p[16];
};
template <int W>
RTD double recw(const RecRegs& R) { return w2r((W & 1) ? R.p[W >> 1].y : R.p[W >> 1].x); }
template <int W>
RTD D3 rec3(const RecRegs& R) { return d3(recw<W>(R), recw<W + 1>(R), recw<W + 2>(R)); }

// Wave-uniform fetch-and-add on the SCALAR memory path (s_atomic_add, gfx9 family incl. gfx950; checked against
// vector atomics on the same word by tools/experiments/satomic_test.hip).  Its return travels through lgkmcnt, so
// the wave does not wait for its vector stores in flight (a vector atomic's return shares vmcnt with them and comes
// back in order behind them).  Executes once per wave whatever EXEC is: call it from wave-uniform control flow only.
RTD uint32_t wave_atomic_add(uint32_t* p, uint32_t v) {
    uint32_t ret = __builtin_amdgcn_readfirstlane(v);
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(ret) : "s"(p) : "memory");
    return ret;
}
// ---- per-wave chunks of a shared counter.  Same-address atomics saturate near 88 per microsecond on this chip
// (profiles/r03_exp_atomic_load.txt: rounds 1-3 ran k_shade's two list counters at 77 % of that), so a wave takes
// `chunk` consecutive indices per atomic and hands them out itself; a request that does not fit the rest of the chunk is
// SPLIT over the old and a new chunk, so only the end of a wave's LAST chunk stays unused (the caller fills it with
// null entries).  chunk = 0: exact requests, one atomic each (small launches: no unused ends at all).
struct Cursor {
    uint32_t cur, end;
};
// entries per atomic, sized so that the unused ends of all waves together stay below n / 16
RTD uint32_t pick_chunk(uint32_t n, uint32_t n_waves) {
    const uint32_t c = n / (n_waves * 16u);
    if (c < 64u) return 0u;
    return c >= 512u ? 512u : (c >= 256u ? 256u : (c >= 128u ? 128u : 64u));
}
// wave-uniform: c, chunk, cnt (1..64 <= chunk); the lanes that take part pass rank = 0 .. cnt-1
RTD uint32_t cursor_take(Cursor& c, uint32_t* counter, uint32_t chunk, uint32_t cnt, uint32_t rank) {
    if (chunk == 0u) return wave_atomic_add(counter, cnt) + rank;
    const uint32_t rem = c.end - c.cur;
    uint32_t idx = c.cur + rank;
    if (cnt > rem) {
        const uint32_t base = wave_atomic_add(counter, chunk);
        if (rank >= rem) idx = base + (rank - rem);
        c.cur = base + (cnt - rem);
        c.end = base + chunk;
    } else {
        c.cur += cnt;
    }
    return idx;
}
RTD void cursor_pad(const Cursor& c, uint32_t* arr, uint32_t cap) {
    for (uint32_t i = c.cur + (threadIdx.x & 63u); i < c.end; i += 64u)
        if (i < cap) arr[i] = kNullEntry;
}
// ---- a wave's share of a list: runs of `span` consecutive 64-entry groups, taken from a cursor until the list is used up.
// Not a fixed span per wave: the class kernels of a bounce and the light kernel run side by side (abi.hip: run_lane), so a
// launch's blocks are not all resident from its start -- a block that starts late must find less left, not its full share.
// The cursor is word kGroupCursorWord of the list counter's own 128-B line (scene_dev.h: Ctl::cls_count), cleared by k_plan.
RTD uint32_t group_span(uint32_t n_groups, uint32_t n_waves) {
    const uint32_t s = n_groups / (n_waves * 8u);
    return s < 1u ? 1u : (s > 32u ? 32u : s);
}
// statistics go to kStatShards cache lines that the host sums
RTD DevStats* stat_shard(DevStats* stats) { return stats + (blockIdx.x & (kStatShards - 1)); }

// One camera sample of the batch: integrator.rs:357-366 + sampler.rs:606-613 + geometry.rs:177-190 (+ util.rs:105-113).
// g = its index in the batch = its film staging slot.
RTD void camera_sample(const rt_camera& cam, const ChunkDesc& ck, const uint32_t* __restrict__ pix_list, uint32_t g, D3& o, D3& d,
                       uint64_t& rng_out) {
    const uint32_t s_local = g / ck.n_pixels, p_local = g - s_local * ck.n_pixels;
    const uint32_t pix = pix_list[ck.pixel_base + p_local];
    const uint32_t px = pix % ck.width, py = pix / ck.width;
    uint64_t rng = rng_init(ck.seed, (uint64_t)pix, (uint64_t)(ck.sample_base + s_local));
    const double ox = rng_next(rng), oy = rng_next(rng);
    (void)rng_next(rng);  // time
    (void)rng_next(rng);  // lens.x
    (void)rng_next(rng);  // lens.y
    const double fx = (double)px + ox, fy = (double)py + oy;
    const double u = fx / (double)ck.width, v = fy / (double)ck.height;
    double dx, dy;
    for (;;) {  // rand_in_disk
        dx = rng_next(rng);
        dy = rng_next(rng);
        if (dx * dx + dy * dy < 1.0) break;
    }
    const D3 in_disk = d3(dx, dy, 0.0) * cam.lens_radius;
    const D3 cu = d3(cam.u[0], cam.u[1], cam.u[2]), cv = d3(cam.v[0], cam.v[1], cam.v[2]);
    const D3 offset = cu * in_disk.x + cv * in_disk.y;
    const D3 origin = d3(cam.origin[0], cam.origin[1], cam.origin[2]);
    const D3 ulc = d3(cam.upper_left_corner[0], cam.upper_left_corner[1], cam.upper_left_corner[2]);
    const D3 ho = d3(cam.horizontal_offset[0], cam.horizontal_offset[1], cam.horizontal_offset[2]);
    const D3 vo = d3(cam.vertical_offset[0], cam.vertical_offset[1], cam.vertical_offset[2]);
    const D3 to = ulc + ho * u - vo * v;
    const D3 dir = to - origin;
    (void)rng_next(rng);  // rand_range(t0, t1)
    o = origin + offset;
    d = dir - offset;
    rng_out = rng;
}
// ------------------------------------------------------------------ generate
// (RT_KERNELS_CORE: the kernels that are not templates are compiled by ONE translation unit per precision, tu/tu_core.hip)
#if defined(RT_KERNELS_CORE) && !defined(RT_F32)  // (precision-independent: compiled once, in the f64 namespace)
// k_plan (one thread): how many camera samples this lane starts now = free pool slots, limited by
// what the batch still holds; reserves them from the shared batch counter and appends them to the
// pool / ray queue of iteration `it`.  Also clears the ring entries of iteration it+2.
__global__ void k_plan(Ctl* ctl, BatchCtl* batch, uint32_t it, uint32_t pool, unsigned long long batch_total,
                       DevStats* stats) {
    const uint32_t r = it % kRing;
    const uint32_t live = ctl->n_active[r];
    unsigned long long want = pool > live ? pool - live : 0u;
    unsigned long long first = 0;
    if (want) {
        first = atomicAdd(&batch->next, want);
        if (first >= batch_total)
            want = 0;
        else if (first + want > batch_total)
            want = batch_total - first;
    }
    ctl->gen_count = (uint32_t)want;
    ctl->gen_first = (uint32_t)first;
    ctl->gen_slot = live;
    ctl->gen_q = ctl->n_rays[r];
    ctl->gen_ring[it & 3u][0] = (uint32_t)first;
    ctl->gen_ring[it & 3u][1] = live;
    ctl->n_active[r] = live + (uint32_t)want;
    ctl->n_rays[r] += (uint32_t)want;
    const uint32_t z = (it + 2) % kRing;
    ctl->n_active[z] = 0;
    ctl->n_rays[z] = 0;
    ctl->head[z] = 0;
    for (int x = 0; x < 8; x++) ctl->xhead[(it + 2) & 3u][x][0] = 0;
    for (int c = 0; c < kMaxCls; c++) ctl->cls_count[(it + 2) & 3u][c][0] = ctl->cls_count[(it + 2) & 3u][c][kGroupCursorWord] = 0;
    ctl->fold_count[(it + 2) & 3u][0] = 0;
    if (want) {
        atomicAdd(&stats->paths, want);
        atomicAdd(&stats->r1, want);
    }
}

#endif

#ifdef RT_KERNELS_CORE
// the batch's next camera samples: their rays, where the traversal kernel reads them, and their queue entries
__global__ __launch_bounds__(256) void k_generate(PathState st, rt_camera cam, ChunkDesc ck,
                                                  const uint32_t* __restrict__ pix_list, uint32_t* queue,
                                                  const Ctl* ctl) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ctl->gen_count) return;
    const uint32_t g = ctl->gen_first + idx;  // path index inside the batch = film staging slot
    const uint32_t slot = ctl->gen_slot + idx;
    D3 o, d;
    uint64_t rng;
    camera_sample(cam, ck, pix_list, g, o, d, rng);
    str(st, kRayO, slot, o);
    str(st, kRayD, slot, d);
    st.rng0[slot] = rng;
    queue[ctl->gen_q + idx] = slot | (kRayExt << 30);
}
#endif  // RT_KERNELS_CORE

// --------------------------------------------------------------------- trace
// Persistent waves, while-while traversal, dynamic ray replacement.
//   * A lane whose ray is finished does not wait for the slowest ray of its batch: whenever
//     tune.refill_lanes or more lanes of the wave are idle (or all are), the idle lanes take the next
//     queue entries (__ballot/__popcll prefix) while the others keep their state.  Entries come from a
//     per-wave reservation of tune.reserve entries, so the single queue head sees one atomic per
//     reservation, not per refill (same-address atomics saturate near 88/us).  A finished ray's result is written
//     in those refill rounds, not when it finishes: loads and stores share vmcnt and return in order, so a store
//     in flight would delay the fetch every traversal round waits for.
//   * Every wave step is EITHER a node step OR a single-primitive step, whichever more lanes are
//     waiting for (majority scheduling): the wave never runs the primitive code for the sake of a few
//     lanes while the rest are walking the tree, and vice versa.
//   * Results: a shadow / probe ray writes the primitive it found at its path's slot (sh_prim / pr_prim); an extension
//     ray writes its hit word (scene_dev.h) at its QUEUE position (hitw[]: neighbouring lanes, neighbouring words), from
//     where k_classify deals the paths to the lists of their vertex classes.  (Appending to the class lists here -- a
//     cursor per class, a ballot per class and refill round -- was measured: +10 % on every launch of this kernel, camera
//     rays included; so was reading the rays out of the 256-B records, one 16-B request per lane and load: +5-10 %.)
// Every wave leaves the loop once the queue is exhausted and its own lanes are done.
template <bool COUNT, bool SIMPLE>
__global__ __launch_bounds__(256, RT_TRACE_WAVES) void k_trace(DevScene sc, PathState st, const uint32_t* __restrict__ queue,
                                               Ctl* ctl, uint32_t it_abs, DevStats* stats, TraceTune tune,
                                               MirrorEntry* mirror, uint32_t seq, const BatchCtl* batch,
                                               unsigned long long batch_total, uint32_t* hitw) {
    const uint32_t it = it_abs % kRing;
    const uint32_t n = ctl->n_rays[it];
    const unsigned long long t_start = COUNT ? wall_clock64() : 0ull;
    if (blockIdx.x == 0 && threadIdx.x == 0 && mirror) {
        const unsigned long long nx = batch->next;
        const unsigned long long rem = nx >= batch_total ? 0ull : batch_total - nx;
        mirror[it].n_active = ctl->n_active[it];
        mirror[it].n_rays = n;
        mirror[it].remaining = rem > 0xffffffffull ? 0xffffffffu : (uint32_t)rem;
        __hip_atomic_store(&mirror[it].seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // (the wave's number made wave-uniform for the compiler too: what is indexed / looped with it then lives in SGPRs)
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t n_waves = gridDim.x * 4u;
    if (sc.n_nodes == 0) {  // a scene without primitives (an environment only): every query misses
        for (uint32_t i0 = blockIdx.x * 256u + wave * 64u; i0 < n; i0 += gridDim.x * 256u) {
            const uint32_t i = i0 + lane;
            const uint32_t e = i < n ? queue[i] : kNullEntry;
            const uint32_t slot = e & kSlotMask, kind = e >> 30;
            const bool ext = kind == kRayExt;
            if (ext) hitw[i] = 0u;
            if (kind == kRayShadow) st.sh_prim[slot] = -1;
            if (kind == kRayProbe) st.pr_prim[slot] = -1;
        }
        return;
    }
    // a block only joins the work-pulling loop if the queue can give it at least one batch:
    // tail iterations with a handful of rays then cost a launch, not a grid of atomics
    if (blockIdx.x * 256u >= n) return;
    TravCount tc{0, 0, 0};
    __shared__ int2 lds_stack[kLdsStack * 256];
    RT_TRAV_STACK(ts, lds_stack)
#if RT_LDS_NODES > 0
    // BASELINE north_star: "BVH-node tiles staged in LDS" -- the first RT_LDS_NODES nodes, the top of the tree, which
    // rt_scene_commit lays out breadth-first, are copied into LDS once per block; node_step reads them with ds_read
    // (geom.h), the rest of the tree with global loads
    __shared__ DevNode lds_nodes[RT_LDS_NODES];
    {
        const uint32_t n_top = sc.n_nodes < (uint32_t)RT_LDS_NODES ? sc.n_nodes : (uint32_t)RT_LDS_NODES;
        const float4* src = reinterpret_cast<const float4*>(sc.nodes);
        float4* dst = reinterpret_cast<float4*>(lds_nodes);
        for (uint32_t i = threadIdx.x; i < n_top * 8u; i += 256u) dst[i] = src[i];
        __syncthreads();
        ts.top_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_nodes;  // C-style: the address-space cast
        ts.n_top = n_top;
    }
#endif
    Trav tv;
    tv.done = true;
    tv.cur = 0;
    bool has_ray = false;
    bool wb = false;  // this lane's finished ray still has to write its result (done in refill rounds)
    uint32_t ray_steps0 = 0;
    uint32_t diag_rounds[4] = {0, 0, 0, 0};  // wave-uniform diagnostics (instrumented build only)
    uint32_t diag_max_steps = 0, diag_over[2] = {0, 0};  // per lane: longest traversal, rays over 64 / 256 steps
    bool exhausted = false;  // wave-uniform: the queue and this wave's reservation have no more entries
    uint32_t slot_kind = 0, q_idx = 0;  // the lane's queue entry and its position
    uint32_t res_next = 0, res_end = 0;  // wave-uniform: [res_next, res_end) is reserved for this wave
    uint32_t res_base = 0, q_lo = 0, q_hi = 0;  // start of the reservation; its queue entries, two per lane
    // One atomic on the queue head hands a wave `reserve` entries.  128 is the measured optimum: 64 costs 30 % of the
    // kernel's time (twice the atomic -> queue -> ray-data chains); larger reservations served through a sliding
    // 128-entry window were 5-10 % slower (the extra code in the refill path, and neighbouring queue entries are
    // neighbouring paths: small reservations let concurrent waves share their nodes in L2).
    // Small queues: shrink the reservation so that the tail still spreads over the waves.
    uint32_t reserve = (uint32_t)tune.reserve;
    if (reserve > 128u) reserve = 128u;  // two cached entries per lane
    while (reserve > 64u && (uint64_t)reserve * n_waves * 4u > n) reserve >>= 1;
#if RT_XCD_QUEUE
    uint32_t xq_first, xq_tries = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xq_first));
    xq_first &= 7u;
    const uint32_t xq_len = ((n + 7u) / 8u + 127u) & ~127u;  // part length, a multiple of the reservation
#endif
    for (;;) {
        const unsigned long long idle = __ballot(!has_ray);
        const int n_idle = __popcll(idle);
        // Results are written in refill rounds only: a store in flight delays every later `s_waitcnt vmcnt`
        // (loads and stores share the counter), and some lane finishes in almost every round.
        if (n_idle >= tune.refill_lanes || exhausted) {
            const uint32_t slot = slot_kind & kSlotMask, kind = slot_kind >> 30;
            const bool wb_ext = wb && kind == kRayExt;
            if (wb_ext) hitw[q_idx] = tv.best_prim < 0 ? 0u : hit_word(tv.best_prim, tv.best_slot);
            if (wb && kind == kRayShadow) st.sh_prim[slot] = tv.best_prim;
            if (wb && kind == kRayProbe) st.pr_prim[slot] = tv.best_prim;
            wb = false;
        }
        if (!exhausted && (n_idle >= tune.refill_lanes)) {
            if (res_next >= res_end) {
                // (scalar atomic: its return does not wait behind the result stores issued just above; 2 % of the kernel)
#if RT_XCD_QUEUE
                // The queue in eight contiguous parts with a head each, one per XCD: a wave drains the part of its own
                // XCD first, then the following ones.  Neighbouring queue entries are neighbouring paths, so an XCD's
                // L2 sees one moving window of the scene instead of all eight (and a head is shared by 1/8 of the
                // waves): k_trace -1.5 % (C4) / -3.4 % (C3) / -5.3 % (C2).
                uint32_t base = n;
                while (xq_tries < 8u) {
                    const uint32_t part = (xq_first + xq_tries) & 7u;
                    const uint32_t lo = part * xq_len, hi = lo + xq_len < n ? lo + xq_len : n;
                    if (lo < n) {
                        const uint32_t off = wave_atomic_add(&ctl->xhead[it & 3u][part][0], reserve);
                        if (off < hi - lo) {
                            base = lo + off;
                            res_end = base + reserve < hi ? base + reserve : hi;
                            break;
                        }
                    }
                    xq_tries++;
                }
                res_next = base;
                if (base >= n) {
                    res_end = res_next;
                    exhausted = true;
                }
#else
                const uint32_t base = wave_atomic_add(&ctl->head[it], reserve);
                res_next = base;
                res_end = base + reserve < n ? base + reserve : n;
                if (base >= n) {
                    res_end = res_next;
                    exhausted = true;
                }
#endif
                // the reservation's queue entries are fetched once, two per lane (reserve <= 128), and handed out
                // with cross-lane reads: a refill then waits for the ray data only, not for queue -> ray data
                res_base = base;
                q_lo = base + lane < res_end ? queue[base + lane] : kNullEntry;
                q_hi = base + 64u + lane < res_end ? queue[base + 64u + lane] : kNullEntry;
            }
            if (!exhausted) {
                const uint32_t base = res_next;
                const uint32_t take = (uint32_t)n_idle < res_end - res_next ? (uint32_t)n_idle : res_end - res_next;
                res_next += take;
                const uint32_t my = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                const uint32_t idx = base + my;
                const uint32_t rel = idx - res_base;  // < 128 for the lanes that take an entry
                const uint32_t e_lo = __shfl(q_lo, (int)(rel & 63u), 64), e_hi = __shfl(q_hi, (int)(rel & 63u), 64);
                const uint32_t e = rel < 64u ? e_lo : e_hi;
                // (an entry of kind kRayNone is the unused end of a shading wave's queue chunk: nothing to trace)
                if (!has_ray && my < take && (e >> 30) != kRayNone) {
                    const uint32_t slot = e & kSlotMask, kind = e >> 30;
                    // (the ray arrays through one base and one stride, ldr: the kernel had 8-10 SGPRs spilled, now none, and its
                    // general instance 39 spilled VGPRs instead of 46)
                    D3 o = ldr(st, kRayO, slot);
                    D3 d = ldr(st, kind == kRayExt ? kRayD : (kind == kRayShadow ? kRaySp : kRayPd), slot);
                    double tmin = kSmall;
                    if (kind == kRayShadow) {  // Visibility::unoccluded, hittable.rs:25-32: d = target - o
                        d = d - o;
                        o = o + d * kSmall;
                        tmin = 0.0;
                    }
                    trav_init(tv, sc, o, d, tmin, kInf);
                    slot_kind = e;
                    q_idx = idx;
                    has_ray = true;
                    if (COUNT) ray_steps0 = tc.nodes + tc.tris + tc.others;
                }
            }
        }
        if (__ballot(has_ray) == 0ull) {
            if (exhausted) break;
            continue;  // all lanes idle: the refill above ran and either found rays or set `exhausted`
        }
        // majority scheduling: run the kind of step (node or primitive) that more lanes are waiting for;
        // the minority waits, and becomes the majority as the others change phase.
        const bool at_node = has_ray && !tv.done && tv.cur >= 0;
        const bool at_leaf = has_ray && !tv.done && tv.cur < 0;
        const unsigned long long mn = __ballot(at_node), ml = __ballot(at_leaf);
        const int cn = __popcll(mn), cl = __popcll(ml);
        if (cn * tune.node_bias >= cl * 4 && cn > 0) {
            if (COUNT) {  // diagnostic: lanes busy per node round
                diag_rounds[0]++;
                diag_rounds[1] += (uint32_t)cn;
            }
            if (at_node) node_step<COUNT>(tv, sc, ts, &tc);
        } else if (cl > 0) {
            if (COUNT) {  // diagnostic: lanes busy per primitive round
                diag_rounds[2]++;
                diag_rounds[3] += (uint32_t)cl;
            }
            if (at_leaf) leaf_step<COUNT, SIMPLE>(tv, sc, ts, &tc);
        }
        if (has_ray && tv.done) {
            if (COUNT) {  // diagnostic: longest traversal, and how many rays needed more than 64 / 256 steps
                const uint32_t steps = tc.nodes + tc.tris + tc.others - ray_steps0;
                diag_max_steps = steps > diag_max_steps ? steps : diag_max_steps;
                diag_over[0] += steps > 64u;
                diag_over[1] += steps > 256u;
            }
            wb = true;
            has_ray = false;
        }
    }
    if (COUNT) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && n < 4096u) {  // diagnostic: in-kernel time of tail launches (10 ns ticks)
            atomicAdd(&stats->pad[0], wall_clock64() - t_start);
            atomicAdd(&stats->pad[1], 1ull);
            atomicAdd(&stats->pad[2], (unsigned long long)n);
        }
        DevStats* sh = stat_shard(stats);
        // (per lane and launch, not per ray: a same-address atomic per finished ray made this build 8x slower)
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)diag_max_steps, o, 64);
            diag_max_steps = other > diag_max_steps ? other : diag_max_steps;
        }
        if (lane == 0) atomicMax(&stats->pad[3], (unsigned long long)diag_max_steps);
        if (diag_over[0]) atomicAdd(&sh->pad[4], (unsigned long long)diag_over[0]);
        if (diag_over[1]) atomicAdd(&sh->pad[5], (unsigned long long)diag_over[1]);
        if (lane == 0)
            for (int q = 0; q < 4; q++) atomicAdd(&sh->pad[8 + q], (unsigned long long)diag_rounds[q]);
        atomicAdd(&sh->nodes, (unsigned long long)tc.nodes);
        atomicAdd(&sh->tris, (unsigned long long)tc.tris);
        atomicAdd(&sh->others, (unsigned long long)tc.others);
    }
}

#ifdef RT_KERNELS_CORE
// rt_intersect_batch: the same traversal on caller rays
__global__ __launch_bounds__(256) void k_intersect_batch(DevScene sc, const rt_ray* __restrict__ rays, uint64_t n,
                                                         rt_hit* hits) {
    __shared__ int2 lds_stack[kLdsStack * 256];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const rt_ray r = rays[i];
    TravCount tc{0, 0, 0};
    RT_TRAV_STACK(ts, lds_stack)
    double t;
    const int32_t prim = closest_hit<true>(sc, d3(r.origin[0], r.origin[1], r.origin[2]),
                                           d3(r.dir[0], r.dir[1], r.dir[2]), r.tmin, r.tmax, t, ts, &tc);
    rt_hit h;
    h.t = prim >= 0 ? t : kInf;
    h.prim = prim;
    // diagnostic: what the traversal of this ray cost (saturating bytes: nodes | triangles << 8 | spheres/rects << 16)
    h.reserved = (tc.nodes > 255u ? 255u : tc.nodes) | ((tc.tris > 255u ? 255u : tc.tris) << 8) |
                 ((tc.others > 255u ? 255u : tc.others) << 16);
    hits[i] = h;
}
#endif  // RT_KERNELS_CORE

// ------------------------------------------------------------------ classify
// Deals the paths whose extension ray k_trace has just traced to the lists of their vertex classes (scene_dev.h): a
// streaming counting sort over the queue and the hit words, in three small launches -- count (every wave: the classes of
// its contiguous span of the queue), scan (one block: where every wave's part of every list starts), scatter (the
// entries).  No atomic (a first version appended through per-wave chunks of the class counters: 150 k same-address
// atomics per launch, 112 ms of a C4 frame -- the pass ran at the atomic ceiling, not at its 32 B per path), no unused
// entries, and the lists keep the queue's order, i.e. the order of the paths: what an XCD traces and shades stays one
// moving window of the scene.  A list entry carries everything the traversal found for the path -- hit word, and the
// shadow / probe result of a path with pending light terms -- so the shading kernels read no result array.
constexpr uint32_t kClassifyWavesMax = 8192;  // waves of the count / scatter launches (the scan block handles this many)
RTD uint32_t classify_span(uint32_t n, uint32_t n_waves, uint32_t wave_g, uint32_t& g_end) {
    const uint32_t n_groups = (n + 63u) / 64u;
    const uint32_t per_wave = (n_groups + n_waves - 1u) / n_waves;
    const uint32_t g_first = wave_g * per_wave < n_groups ? wave_g * per_wave : n_groups;
    g_end = g_first + per_wave < n_groups ? g_first + per_wave : n_groups;
    return g_first;
}
template <int DUMMY>
__global__ __launch_bounds__(256) void k_classify_count(const uint32_t* __restrict__ queue, const uint32_t* __restrict__ hitw, const Ctl* ctl,
                                                        uint32_t it_abs, uint32_t* __restrict__ counts /* [kMaxCls][n_waves] */) {
    const uint32_t n = ctl->n_rays[it_abs % kRing];
    const uint32_t lane = threadIdx.x & 63u, n_waves = gridDim.x * 4u;
    const uint32_t wave_g = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t g_end;
    const uint32_t g_first = classify_span(n, n_waves, wave_g, g_end);
    uint32_t cnt[kMaxCls];
#pragma unroll
    for (int c = 0; c < kMaxCls; c++) cnt[c] = 0u;
    for (uint32_t g0 = g_first; g0 < g_end; g0 += 4u) {
        uint32_t e4[4], hw4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = (g0 + u) * 64u + lane;
            const bool in = g0 + u < g_end && i < n;
            e4[u] = in ? queue[i] : kNullEntry;
            hw4[u] = in ? hitw[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t cls = (e4[u] >> 30) == kRayExt ? (hw4[u] >> kClsShift & (uint32_t)(kMaxCls - 1)) : (uint32_t)kMaxCls;
#pragma unroll
            for (int c = 0; c < kMaxCls; c++) cnt[c] += (uint32_t)__popcll(__ballot(cls == (uint32_t)c));
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < kMaxCls; c++) counts[(uint32_t)c * n_waves + wave_g] = cnt[c];
    }
}
// one block of 8 waves, a wave per class: exclusive prefix of the class's counts over the waves (in place), total -> list
// length.  64 counts per step, prefix inside the step by six shuffles, the running total carried along -- no block barrier
// (a first version scanned through LDS with 160 of them: 69 us per launch, 2 % of a C2 frame).
template <int DUMMY>
__global__ __launch_bounds__(512) void k_classify_scan(uint32_t* counts, uint32_t n_waves, Ctl* ctl, uint32_t it_abs) {
    static_assert(kMaxCls == 8, "one wave per class");
    __shared__ uint32_t s_total[kMaxCls];
    const uint32_t c = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t* row = counts + c * n_waves;
    uint32_t run = 0;
    for (uint32_t i0 = 0; i0 < n_waves; i0 += 512u) {  // eight steps' loads in flight together (a step alone is one load latency)
        uint32_t v0, v1, v2, v3, v4, v5, v6, v7;
#define RT_SCAN_LD(k) v##k = i0 + (k) * 64u + lane < n_waves ? row[i0 + (k) * 64u + lane] : 0u;
        RT_SCAN_LD(0) RT_SCAN_LD(1) RT_SCAN_LD(2) RT_SCAN_LD(3) RT_SCAN_LD(4) RT_SCAN_LD(5) RT_SCAN_LD(6) RT_SCAN_LD(7)
#undef RT_SCAN_LD
#define RT_SCAN_STEP(k)                                                      \
    {                                                                        \
        uint32_t incl = v##k;                                                \
        for (int d = 1; d < 64; d <<= 1) {                                   \
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);       \
            if ((int)lane >= d) incl += up;                                  \
        }                                                                    \
        const uint32_t i = i0 + (k) * 64u + lane;                            \
        if (i < n_waves) row[i] = run + incl - v##k;                         \
        run += (uint32_t)__shfl((int)incl, 63, 64);                          \
    }
        RT_SCAN_STEP(0) RT_SCAN_STEP(1) RT_SCAN_STEP(2) RT_SCAN_STEP(3) RT_SCAN_STEP(4) RT_SCAN_STEP(5) RT_SCAN_STEP(6) RT_SCAN_STEP(7)
#undef RT_SCAN_STEP
    }
    if (lane == 0) {
        ctl->cls_count[it_abs & 3u][c][0] = run;
        s_total[c] = run;
    }
    __syncthreads();
    if (lane == 0) {  // the lists share one arena, back to back in class order
        uint32_t base = 0;
        for (uint32_t k = 0; k < c; k++) base += s_total[k];
        ctl->cls_base[it_abs & 3u][c] = base;
    }
}
template <int DUMMY>
__global__ __launch_bounds__(256) void k_classify_scatter(const uint32_t* __restrict__ queue, const uint32_t* __restrict__ hitw,
                                                          const Ctl* ctl, uint32_t it_abs, const uint32_t* __restrict__ offsets, Lists lists) {
    const uint32_t n = ctl->n_rays[it_abs % kRing];
    const uint32_t lane = threadIdx.x & 63u, n_waves = gridDim.x * 4u;
    const uint32_t wave_g = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t g_end;
    const uint32_t g_first = classify_span(n, n_waves, wave_g, g_end);
    if (g_first >= g_end) return;
    uint32_t at[kMaxCls];
#pragma unroll
    for (int c = 0; c < kMaxCls; c++)
        at[c] = (uint32_t)c < lists.n_cls ? ctl->cls_base[it_abs & 3u][c] + offsets[(uint32_t)c * n_waves + wave_g] : 0u;
    // the camera samples k_generate appended to this queue: entries [gen_q, gen_q + gen_count), sample gen_first + k
    const uint32_t fresh_lo = ctl->gen_q, fresh_n = ctl->gen_count;
    const unsigned long long below = (1ull << lane) - 1ull;
    // four groups per round: their loads are in flight together
    for (uint32_t g0 = g_first; g0 < g_end; g0 += 4u) {
        uint32_t e4[4], hw4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = (g0 + u) * 64u + lane;
            const bool in = g0 + u < g_end && i < n;
            e4[u] = in ? queue[i] : kNullEntry;
            hw4[u] = in ? hitw[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t e = e4[u], hw = hw4[u];
            const bool ext = (e >> 30) == kRayExt;
            const uint32_t cls = ext ? (hw >> kClsShift & (uint32_t)(kMaxCls - 1)) : (uint32_t)kMaxCls;
            const uint32_t qi = (g0 + u) * 64u + lane;
            const bool fresh = qi - fresh_lo < fresh_n;  // (unsigned: also false below fresh_lo)
            const uint32_t slot_word = (e & kSlotMask) | ((e & kQPending) ? kEntPending : 0u) | (fresh ? kEntFresh : 0u);
#pragma unroll
            for (int c = 0; c < kMaxCls; c++) {
                const bool mine = cls == (uint32_t)c;
                const unsigned long long m = __ballot(mine);
                const uint32_t to = at[c] + (uint32_t)__popcll(m & below);
                if (mine && to < lists.cap) lists.ent[to] = ListEnt{slot_word, hw};
                at[c] += (uint32_t)__popcll(m);
            }
        }
    }
}

// --------------------------------------------------------------------- shade
// The per-vertex work in two halves:
//   shade_a: fold the previous vertex's direct light (estimate_direct's two additions, using the R2/R3
//            results), rebuild the record of the vertex the extension ray found, emitted-light rule.
//   shade_b: BSDF, one-light NEE + MIS (integrator.rs:530-634), continuation sample, Russian roulette
//            (integrator.rs:421-442); returns the survivor's state (ShadeRes), its rays go to the ray arrays.
// A class kernel runs them for one bounce of every path of its class in three phases per 64-path group: records in
// (whole lines through the wave's LDS staging area: rec_fetch), compute (no barrier inside), records out (rec_store).
// k_tail loops them per lane until the path retires and touches its records directly.  KIND (scene_dev.h: kKind*) says
// what kind of hit the caller's paths have, i.e. which record code is compiled in.
struct ShadeA {
    D3 L, o, d, beta;
    HitRec rec;
    uint32_t fl, bounces;
    uint32_t orig;  // film staging slot of the path
    uint64_t rng;
    bool live, spec, will_shade;
};

// A path's record in registers.  rec_fetch: every lane of a wave calls it (class kernels) -- line 0 of all valid lanes'
// records, and line 1 (the pending direct-light terms) of those whose list entry and flags say they carry any, arrive as
// whole lines through the staging area.  rec_load: the same by the lane itself (k_tail).
RTD bool rec_wants_fold(const RecRegs& R, bool valid, bool pending) {
    return valid && pending && ((uint32_t)(R.p[3].y >> 32) & (kHasShadow | kHasProbe)) != 0u;
}
RTD bool rec_wants_line1(const RecRegs& R, bool valid, bool pending) {
    return valid && pending && ((uint32_t)(R.p[3].y >> 32) & kLine1) != 0u;
}
// a pending light sample alone (no kLine1): its complete contribution sits in the fa* words of the slot; it takes the place
// of A in the registers (words kWA .. kWA+2), where shade_a adds it as it is
RTD void rec_put_fa(RecRegs& R, D3 fa) {
    R.p[8].x = r2w(fa.x); R.p[8].y = r2w(fa.y);
    R.p[9].x = r2w(fa.z);
}
// line 0 of a camera sample that has not been shaded yet (scene_dev.h: kEntFresh): ray and RNG state where k_generate
// left them, beta = 1, L = 0, no flags
template <bool RAY>
RTD void rec_fresh(const PathState& in, uint32_t slot, uint32_t g, RecRegs& R) {
    // (RAY = false: an escaped camera sample without an environment to see needs nothing but its film slot)
    D3 o = black(), d = black();
    uint64_t rng = 0;
    if (RAY) {
        o = ldr(in, kRayO, slot);
        d = ldr(in, kRayD, slot);
        rng = in.rng0[slot];
    }
    R.p[0].x = r2w(o.x); R.p[0].y = r2w(o.y);
    R.p[1].x = r2w(o.z); R.p[1].y = r2w(d.x);
    R.p[2].x = r2w(d.y); R.p[2].y = r2w(d.z);
    R.p[3].x = rng; R.p[3].y = (rt_w)g;
    R.p[4].x = r2w(1.0); R.p[4].y = r2w(1.0);
    R.p[5].x = r2w(1.0); R.p[5].y = r2w(0.0);
    R.p[6].x = r2w(0.0); R.p[6].y = r2w(0.0);
}
template <bool RAY>
RTD void rec_fetch(rt_w2* stage, const PathState& in, uint32_t slot, bool valid, bool pending, bool fresh, uint32_t fresh_g, RecRegs& R) {
    const uint32_t lane = threadIdx.x & 63u;
    const bool stored = valid && !fresh;
    // (issued before the lines, on the list entry's word alone: in flight together with them)
    D3 fa = black();
    if (stored && pending) fa = ldr(in, kRayFa, slot);
    if (__ballot(stored)) {
        stage_fetch(stage, in, stored ? slot : kNullEntry, 0u);
#pragma unroll
        for (int k = 0; k < 7; k++) R.p[k] = stage[lane * kStagePitch + k];
    }
    if (valid && fresh) rec_fresh<RAY>(in, slot, fresh_g, R);
    const bool fold = rec_wants_line1(R, valid, pending);
    if (__ballot(fold)) {
        wave_sync_lds();  // (every lane has read its line 0)
        stage_fetch(stage, in, fold ? slot : kNullEntry, 1u);
#pragma unroll
        for (int k = 0; k < 5; k++) R.p[8 + k] = stage[lane * kStagePitch + k];
    }
    if (stored && pending && !fold) rec_put_fa(R, fa);
    wave_sync_lds();  // (the area is free again)
}
RTD void rec_load(const PathState& in, uint32_t slot, bool pending, RecRegs& R) {
    const rt_w2* rp = reinterpret_cast<const rt_w2*>(rec_words(in, slot));
#pragma unroll
    for (int k = 0; k < 7; k++) R.p[k] = rp[k];
    if (rec_wants_line1(R, true, pending)) {
#pragma unroll
        for (int k = 0; k < 5; k++) R.p[8 + k] = rp[8 + k];
    } else if (rec_wants_fold(R, true, pending)) {
        rec_put_fa(R, ldr(in, kRayFa, slot));
    }
}

// `R` = the path's record, `hit` = the extension ray's hit word (scene_dev.h), `some` = it hit something (both ignored for
// a fold-only path), `sh`, `pp` = results of the path's shadow / probe ray, `pending` = it may carry pending light terms.
template <int FEAT, int KIND>
RTD void shade_a(const DevScene& sc, const PathState& in, const RecRegs& R, uint32_t slot, uint32_t hit, bool some, int32_t sh, int32_t pp,
                 bool pending, bool valid, uint32_t max_depth, ShadeA& a) {
    a.rng = R.p[3].x;
    a.orig = (uint32_t)R.p[3].y;
    const uint32_t fl = valid ? (uint32_t)(R.p[3].y >> 32) : 0u;
    a.fl = fl;
    a.live = valid;
    a.L = black();
    a.o = black();
    const bool fold = rec_wants_fold(R, valid, pending);
    if (a.live) {
        a.L = rec3<kWL>(R);
        a.o = rec3<kWO>(R);
        // ---- fold the previous vertex's direct lighting
        if (fold && !(fl & kLine1)) {
            // only a light sample is pending, and K was finite: (A * n_lights) (*) K came ready-made (shade_b); an
            // occluded one adds (0 * n_lights) (*) K = +0, as the general arm below would
            const rt_light& lt = sc.lights[fl >> kLightShift];
            const bool infinite = ((FEAT & kFeatEnv) != 0) && lt.kind == RT_LIGHT_INFINITE;
            const bool seen = infinite ? sh < 0 : sh == (int32_t)lt.prim_index;
            a.L = a.L + (seen ? rec3<kWA>(R) : black());
        } else if (fold) {
            const uint32_t light_idx = fl >> kLightShift;
            const rt_light& lt = sc.lights[light_idx];
            const bool infinite = ((FEAT & kFeatEnv) != 0) && lt.kind == RT_LIGHT_INFINITE;
            D3 ld = black();
            if (fl & kHasShadow) {
                // Visibility::unoccluded(infinite): an area light must be the closest hit, the environment needs a miss
                if (infinite ? sh < 0 : sh == (int32_t)lt.prim_index) ld = ld + rec3<kWA>(R);
            }
            if (fl & kHasProbe) {
                const D3 q_in = rec3<kWQ>(R);
                if (infinite) {
                    // integrator.rs:617-630: an escaped probe sees light.le(ray), already folded into q by shade_b
                    if (pp < 0) ld = ld + q_in;
                } else if (pp >= 0) {
                    const int32_t li = sc.prims[pp].light_index;
                    if (li >= 0 && (uint32_t)li == light_idx) {
                        const D3 pd = ldr(in, kRayPd, slot);
                        const rt_primitive& lpr = sc.prims[pp];
                        if (lpr.kind >= RT_PRIM_XY_RECT && lpr.xform_index < 0) {
                            // axis-aligned rect emitter: the record's normal faces the ray (set_front), so
                            // dot(n, -wi) = |wi's component along the rect's axis| and Light::l sees the colour
                            // unless that component is zero -- no need to build the record
                            double t = 0.0, ra = 0.0, rb = 0.0;
                            D3 to = black(), td = black();
                            if (rect_core(sc, lpr, a.o, pd, kSmall, kInf, t, ra, rb, to, td)) {
                                const rt_light& lt2 = sc.lights[li];
                                const bool lit = lt2.two_sided || absd(rect_axis_comp(lpr.kind, pd)) > 0.0;
                                if (lit && !is_black(d3(lt2.color[0], lt2.color[1], lt2.color[2]))) ld = ld + q_in;
                            }
                        } else {
                            HitRec nh{};
                            if (prim_intersects(sc, pp, a.o, pd, kSmall, kInf, nh)) {
                                const D3 col = light_l(sc.lights[li], nh.n, -pd);  // new_record.le(-wi)
                                if (!is_black(col)) ld = ld + q_in;
                            }
                        }
                    }
                }
            }
            a.L = a.L + cmul(ld * (double)sc.n_lights, rec3<kWK>(R));
        }
    }
    // ---- the vertex found by the extension ray
    const bool active = a.live && !(fl & kFoldOnly);
    a.d = black();
    a.beta = black();
    bool is_some = false;
    a.bounces = fl & kBounceMask;
    a.spec = (fl & kSpecular) != 0;
    if (active) {
        a.d = rec3<kWD>(R);
        a.beta = rec3<kWBeta>(R);
        is_some = KIND != kKindNone && some;
        if (is_some) is_some = hit_record<KIND>(sc, hit, a.o, a.d, kSmall, kInf, a.rec);
        if (a.bounces == 0 || a.spec) {  // integrator.rs:396-411 (Q18)
            if (is_some) {
                const int32_t li = a.rec.light;
                if (li >= 0) a.L = a.L + cmul(light_l(sc.lights[li], a.rec.n, -a.d), a.beta);
            } else if (((FEAT & kFeatEnv) != 0) && sc.env.light >= 0) {
                // escaped: every light adds le(ray), black for all but the infinite one (light.rs:499-512)
                a.L = a.L + cmul(infinite_le(sc, sc.lights[sc.env.light], a.d), a.beta);
            }
        }
    }
    a.will_shade = active && is_some && a.bounces < max_depth;
}

// what a shaded vertex leaves: the survivor's record fields (line 0: o = rec.p, d, rng, flags, beta, L; line 1 when it
// has pending terms: A, Q, K = the beta it arrived with) and which rays it started
struct ShadeRes {
    D3 d, beta, pa, pq;
    uint64_t rng;
    uint32_t flags;
    bool emit_ext, emit_sh, emit_pr, keep;
    bool line1;  // the pending terms go to line 1 (A, Q, K); a lone light sample with a finite K went to the fa* words
};

// Precondition: a.will_shade.  `os` = slot of `out` reserved for this vertex: its rays (origin, extension direction,
// shadow target, probe direction) are written to the ray arrays here, the record by the caller.
template <int FEAT>
RTD ShadeRes shade_b(const DevScene& sc, const PathState& out, uint32_t os, const ShadeA& a) {
    const HitRec& rec = a.rec;
    D3 beta = a.beta;
    uint32_t bounces = a.bounces;
    bool spec = a.spec;
    uint64_t rng = a.rng;
    // (locals are initialised even where every path that reads them assigns them first: inside the class kernels' loop an
    // undefined value is a register that stays allocated around the whole loop -- k_shade_cls<0, mesh> spilled 72 VGPRs
    // with `ShadeA a;` and 7 with `ShadeA a{};`)
    Bsdf bsdf{};
    compute_scattering<FEAT>(sc, rec, bsdf);
    bool has_sh = false, has_pr = false;
    uint32_t light_num = 0;
    D3 pa = black(), pq = black();
    // ---- uniform_sample_one_light / estimate_direct (integrator.rs:530-634)
    if (sc.n_lights > 0) {
        const double pick = rng_next(rng);
        light_num = (uint32_t)(pick * (double)sc.n_lights);
        if (light_num > sc.n_lights - 1) light_num = sc.n_lights - 1;
        const double ul0 = rng_next(rng), ul1 = rng_next(rng);
        const double us0 = rng_next(rng), us1 = rng_next(rng);
        const rt_light& lt = sc.lights[light_num];
        const bool infinite = ((FEAT & kFeatEnv) != 0) && lt.kind == RT_LIGHT_INFINITE;
        const rt_primitive& lp = sc.light_prims[light_num];  // = prims[lt.prim_index] (zeros for the infinite light, unused)
        const uint32_t nsf = RT_BSDF_ALL - RT_BSDF_SPECULAR;
        const D3 ltcolor = d3(lt.color[0], lt.color[1], lt.color[2]);
        D3 sp = black(), sn = black();
        double light_pdf = 0.0;
        D3 wi = black(), color = black();
        if (infinite) {
            infinite_sample_li(sc, lt, rec.p, ul0, ul1, wi, light_pdf, color, sp);
        } else {
            sample_area(sc, lp, ul0, ul1, sp, sn, light_pdf);  // Primitive::sample (Q9)
            const D3 wi_raw = sp - rec.p;
            if (norm2(wi_raw) == 0.0) {
                light_pdf = 0.0;
            } else {
                const D3 wn = normalize(wi_raw);
                light_pdf = light_pdf * norm2(rec.p - sp) / absd(dot(sn, -wn));
            }
            if (light_pdf == 0.0 || norm2(rec.p - sp) == 0.0) {
                light_pdf = 0.0;
                wi = black();
                color = ltcolor;
            } else {
                wi = normalize(sp - rec.p);
                color = light_l(lt, sn, -wi);
            }
        }
        if (light_pdf > 0.0 && !is_black(color)) {
            const D3 f = bsdf_f<FEAT>(bsdf, rec.wo, wi, nsf) * absd(dot(wi, rec.sh_n));
            const double scattering_pdf = bsdf_pdf<FEAT>(bsdf, rec.wo, wi, nsf);
            if (!is_black(f)) {
                has_sh = true;
                const double weight = power_heuristic(1, light_pdf, 1, scattering_pdf);
                pa = cmul(f, color) * (weight / light_pdf);
                str(out, kRaySp, os, sp);
            }
        }
        {
            D3 f2, wi2;
            double spdf;
            uint32_t sampled;
            bsdf_sample_f<FEAT>(bsdf, rec.wo, us0, us1, nsf, rng, f2, wi2, spdf, sampled);
            f2 = f2 * absd(dot(wi2, rec.sh_n));
            if (!is_black(f2) && spdf > 0.0) {
                double weight = 1.0;
                bool go = true;
                if ((sampled & RT_BSDF_SPECULAR) == 0) {
                    const double lpdf = infinite ? infinite_pdf_li(sc, lt, wi2) : prim_pdf(sc, lp, rec.p, wi2);  // Light::pdf_li
                    if (lpdf == 0.0)
                        go = false;
                    else
                        weight = power_heuristic(1, spdf, 1, lpdf);
                }
                if (go) {
                    has_pr = true;
                    // the radiance an escaped probe would see is a function of its direction only: fold it in now
                    const D3 pcol = infinite ? infinite_le(sc, lt, wi2) : ltcolor;
                    pq = is_black(pcol) ? black() : cmul(f2, pcol) * (weight / spdf);
                    str(out, kRayPd, os, wi2);
                }
            }
        }
    }
    // ---- continuation (integrator.rs:421-442)
    const D3 wo = -a.d;
    const double u0 = rng_next(rng), u1 = rng_next(rng);
    D3 f, wi;
    double pdf;
    uint32_t sflags;
    bsdf_sample_f<FEAT>(bsdf, wo, u0, u1, RT_BSDF_ALL, rng, f, wi, pdf, sflags);
    bool cont = !(is_black(f) || pdf == 0.0);
    if (cont) {
        beta = cmul(beta, f) * (absd(dot(wi, rec.sh_n)) / pdf);
        spec = (sflags & RT_BSDF_SPECULAR) != 0;
        if (bounces > 3) {
            const double q = rmax(0.05, 1.0 - rmax(beta.x, rmax(beta.y, beta.z)));
            if (rng_next(rng) < q)
                cont = false;
            else
                beta = beta * (1.0 / (1.0 - q));
        }
        bounces = bounces + 1;
    }
    ShadeRes r;
    r.emit_ext = cont;
    r.emit_sh = has_sh;
    r.emit_pr = has_pr;
    r.keep = cont || has_sh || has_pr;
    // (K = a.beta.  x - x == 0 for every finite x: with an infinite or NaN K the product 0 * K of an occluded sample is not 0,
    // and the general arm of the fold, which forms it, has to run)
    const bool k_finite = a.beta.x - a.beta.x == 0.0 && a.beta.y - a.beta.y == 0.0 && a.beta.z - a.beta.z == 0.0;
    r.line1 = has_pr || (has_sh && !k_finite);
    r.flags = (bounces & kBounceMask) | (spec ? kSpecular : 0u) | (cont ? 0u : kFoldOnly) | (has_sh ? kHasShadow : 0u) |
              (has_pr ? kHasProbe : 0u) | (r.line1 ? kLine1 : 0u) | (light_num << kLightShift);
    r.d = wi;
    r.beta = beta;
    r.pa = pa;
    r.pq = pq;
    r.rng = rng;
    // origin = hit point (spawn_ray, Q4); a fold-only path's d / beta are never read
    if (r.keep) {
        str(out, kRayO, os, rec.p);
        if (cont) str(out, kRayD, os, wi);
        if (has_sh && !r.line1) str(out, kRayFa, os, cmul((black() + pa) * (double)sc.n_lights, a.beta));
    }
    return r;
}
// the survivor's record, written by the lane itself (k_tail)
RTD void rec_store_direct(const PathState& out, uint32_t os, const ShadeA& a, const ShadeRes& r) {
    if (!r.keep) return;
    rt_w* ow = rec_words(out, os);
    st3w<kWO>(ow, a.rec.p);
    st3w<kWD>(ow, r.d);
    st_meta(ow, r.rng, a.orig, r.flags);
    st_beta_l(ow, r.beta, a.L);
    if (r.line1) {
        st3w<kWA>(ow, r.pa);
        st3w<kWQ>(ow, r.pq);
        st3w<kWK>(ow, a.beta);
    }
}

#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 2
#endif
#ifndef RT_SHADE3_MAXFEAT
#define RT_SHADE3_MAXFEAT 3  // instances up to this feature mask are compiled for 3 waves/SIMD (measured: the
                             // two-lobe and row-f4 instances spill too much to gain from it)
#endif
#undef RT_SHADE_BOUND
#ifndef RT_SHADE_FEAT0_WAVES
#define RT_SHADE_FEAT0_WAVES 3  // (experiment) the Lambert-only instances at 2 waves/SIMD: no spills, a third fewer waves
#endif
#define RT_SHADE_BOUND_RULE(F) ((F) == 0 ? RT_SHADE_FEAT0_WAVES : (feat_three_waves(F) && RT_SHADE3_MAXFEAT >= 3 ? 3 : RT_SHADE_WAVES))
#ifdef RT_F32
#ifndef RT_F32_SHADE_WAVES
#define RT_F32_SHADE_WAVES 4  // the binary32 single-lobe instances need 127-138 VGPRs: 4 waves/SIMD with a few spills
                              // (measured: C4 k_shade 628 -> 553 ms, C3 90.6 -> 78.6; at 5 waves 791 / 91.8)
#endif
#define RT_SHADE_BOUND(F) (feat_three_waves(F) ? RT_F32_SHADE_WAVES : RT_SHADE_BOUND_RULE(F))
#else
#define RT_SHADE_BOUND(F) RT_SHADE_BOUND_RULE(F)
#endif

// One vertex class of one bounce (scene_dev.h): the paths of list `cls`, which k_classify filled with the extension rays
// that hit a primitive of that class.  FEAT = the shading features the class's materials need (so a class of glass does
// not carry the microfacet code, nor a Lambertian floor the glass code), KIND = mesh slot / sphere-rect / generic record.
// Persistent, barrier-free: a wave takes a contiguous span of the list in 64-entry groups; output slots, queue entries
// (a chunk per ray kind, so a traversal wave's reservation is mostly one kind) and fold-list entries come from
// per-wave chunks of the shared counters; records move as whole lines through the wave's staging area in LDS.
// WAVES: waves per SIMD the instance is compiled for.  3 for the single-lobe instances (168 VGPRs, 35-110 of them spilled)
// -- except that a scene whose classes are ALL Lambertian runs <0, KIND, 2>: no spills, and C2's two classes gain 12 %
// where C4's Lambertian floor loses 4 % (profiles/r04_exp_feat0_waves.txt).
template <int FEAT, int KIND, int WAVES = RT_SHADE_BOUND(FEAT)>
__global__ __launch_bounds__(256, WAVES) void k_shade_cls(DevScene sc, PathState in, PathState out, Ctl* ctl, uint32_t it_abs,
                                               uint32_t max_depth, Lists lists, uint32_t cls, uint32_t* queue_out, uint32_t q_cap,
                                               uint32_t slot_cap, f64_t* lf, DevStats* stats) {
    const uint32_t itn = (it_abs + 1) % kRing;
    const uint32_t n = ctl->cls_count[it_abs & 3u][cls][0];
    const uint32_t n_groups = (n + 63u) / 64u;
    const uint32_t lane = threadIdx.x & 63u, n_waves = gridDim.x * (blockDim.x >> 6);
    if (n_groups == 0u) return;
    // runs of consecutive groups per wave (group_span): its queue chunks then hold neighbouring paths in list (= queue) order
    const uint32_t span = group_span(n_groups, n_waves);
    uint32_t* const c_groups = &ctl->cls_count[it_abs & 3u][cls][kGroupCursorWord];
    const uint32_t chunk = pick_chunk(n, n_waves < n_groups ? n_waves : n_groups);
    const ListEnt* ent = lists.ent + ctl->cls_base[it_abs & 3u][cls];
    const uint32_t gen_first = ctl->gen_ring[it_abs & 3u][0], gen_slot = ctl->gen_ring[it_abs & 3u][1];  // (camera samples: film slot)
    uint32_t* fold_out = lists.fold[(it_abs + 1u) & 1u];
    uint32_t* const c_slots = &ctl->n_active[itn];
    uint32_t* const c_rays = &ctl->n_rays[itn];
    uint32_t* const c_fold = &ctl->fold_count[(it_abs + 1u) & 3u][0];
    __shared__ rt_w2 s_stage[4][kStageWave];
    rt_w2* const stage = s_stage[threadIdx.x >> 6];
    Cursor cs{0u, 0u}, cq0{0u, 0u}, cq1{0u, 0u}, cq2{0u, 0u}, cf{0u, 0u};
    uint32_t n_r1 = 0, n_r2 = 0, n_r3 = 0, n_v = 0;  // wave-uniform
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t g = 0u, g_end = 0u;  // (one loop, so that the body exists once: the next run is taken when this one is used up)
    for (;; g++) {
        if (g >= g_end) {
            g = wave_atomic_add(c_groups, span);
            if (g >= n_groups) break;
            g_end = g + span < n_groups ? g + span : n_groups;
        }
        const uint32_t i = g * 64u + lane;
        ListEnt e{kNullEntry, 0u};
        if (i < n) e = ent[i];
        const bool valid = e.slot != kNullEntry;
        // ---- records in
        const bool pending = valid && (e.slot & kEntPending), fresh = valid && (e.slot & kEntFresh);
        // (the shadow / probe results of a path with pending terms: near-consecutive slots, in flight with the record)
        const int32_t e_sh = pending ? in.sh_prim[e.slot & kSlotMask] : -1, e_pr = pending ? in.pr_prim[e.slot & kSlotMask] : -1;
        RecRegs R{};
        rec_fetch<true>(stage, in, e.slot & kSlotMask, valid, pending, fresh, gen_first + ((e.slot & kSlotMask) - gen_slot), R);
        // ---- compute
        ShadeA a{};
        shade_a<FEAT, KIND>(sc, in, R, e.slot & kSlotMask, e.hit, true, e_sh, e_pr, pending, valid, max_depth, a);
        const unsigned long long m = __ballot(a.will_shade);
        const uint32_t cnt = (uint32_t)__popcll(m), rank = (uint32_t)__popcll(m & below);
        uint32_t os = 0;
        if (m) {
            os = cursor_take(cs, c_slots, chunk, cnt, rank);
            n_v += cnt;
        }
        ShadeRes r{};
        if (a.will_shade && os < slot_cap) r = shade_b<FEAT>(sc, out, os, a);
        // ---- records out: line 0 of every survivor, line 1 of those with pending terms
        if (m) {
            if (a.will_shade) {
                if (r.keep)
                    stage_line0(stage, rank, os, a.rec.p, r.d, r.rng, a.orig, r.flags, r.beta, a.L);
                else
                    stage_slots(stage)[rank] = kNullEntry;
            }
            stage_flush(stage, out, cnt, 0u);
            if (__ballot(r.line1)) {
                if (a.will_shade) {
                    if (r.line1)
                        stage_line1(stage, rank, os, r.pa, r.pq, a.beta);
                    else
                        stage_slots(stage)[rank] = kNullEntry;
                }
                stage_flush(stage, out, cnt, 1u);
            }
        }
        if (a.live && !r.keep) film_put(lf, a.orig, a.L);  // retired: its radiance goes to the film staging slot of (pixel, sample)
        // ---- rays of the next bounce, and the paths that only have light terms to fold
        const unsigned long long me = __ballot(r.emit_ext), ms = __ballot(r.emit_sh), mp = __ballot(r.emit_pr);
        const unsigned long long mf = __ballot(r.keep && !r.emit_ext);
        if (me) {
            const uint32_t at = cursor_take(cq0, c_rays, chunk, (uint32_t)__popcll(me), (uint32_t)__popcll(me & below));
            if (r.emit_ext && at < q_cap) queue_out[at] = os | ((r.emit_sh || r.emit_pr) ? kQPending : 0u) | (kRayExt << 30);
            n_r1 += (uint32_t)__popcll(me);
        }
        if (ms) {
            const uint32_t at = cursor_take(cq1, c_rays, chunk, (uint32_t)__popcll(ms), (uint32_t)__popcll(ms & below));
            if (r.emit_sh && at < q_cap) queue_out[at] = os | (kRayShadow << 30);
            n_r2 += (uint32_t)__popcll(ms);
        }
        if (mp) {
            const uint32_t at = cursor_take(cq2, c_rays, chunk, (uint32_t)__popcll(mp), (uint32_t)__popcll(mp & below));
            if (r.emit_pr && at < q_cap) queue_out[at] = os | (kRayProbe << 30);
            n_r3 += (uint32_t)__popcll(mp);
        }
        if (mf) {
            const uint32_t at = cursor_take(cf, c_fold, chunk, (uint32_t)__popcll(mf), (uint32_t)__popcll(mf & below));
            if (r.keep && !r.emit_ext && at < lists.cap) fold_out[at] = os;
        }
    }
    cursor_pad(cq0, queue_out, q_cap);
    cursor_pad(cq1, queue_out, q_cap);
    cursor_pad(cq2, queue_out, q_cap);
    cursor_pad(cf, fold_out, lists.cap);
    if (lane == 0) {
        DevStats* sh = stat_shard(stats);
        if (n_r1) atomicAdd(&sh->r1, (unsigned long long)n_r1);
        if (n_r2) atomicAdd(&sh->r2, (unsigned long long)n_r2);
        if (n_r3) atomicAdd(&sh->r3, (unsigned long long)n_r3);
        if (n_v) atomicAdd(&sh->vertices, (unsigned long long)n_v);
    }
}

// The paths that end this bounce without a vertex to shade: list 0 (the extension ray escaped) and the fold list (no
// extension ray, only pending light terms; their shadow / probe results are read at their slots).  Fold, emitted light
// of the environment, film staging -- a few dozen registers, one whole line of the record for most of them.
template <int FEAT>
__global__ __launch_bounds__(256, 4) void k_shade_light(DevScene sc, PathState in, Ctl* ctl, uint32_t it_abs, uint32_t max_depth,
                                                     Lists lists, f64_t* lf) {
    const uint32_t n0 = ctl->cls_count[it_abs & 3u][0][0], n1 = ctl->fold_count[it_abs & 3u][0];
    const uint32_t g0 = (n0 + 63u) / 64u, n_groups = g0 + (n1 + 63u) / 64u;
    const uint32_t lane = threadIdx.x & 63u, n_waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t* fold_in = lists.fold[it_abs & 1u];
    const uint32_t gen_first = ctl->gen_ring[it_abs & 3u][0], gen_slot = ctl->gen_ring[it_abs & 3u][1];
    __shared__ rt_w2 s_stage[4][kStageWave];
    rt_w2* const stage = s_stage[threadIdx.x >> 6];
    if (n_groups == 0u) return;
    const uint32_t span = group_span(n_groups, n_waves);
    uint32_t* const c_groups = &ctl->cls_count[it_abs & 3u][0][kGroupCursorWord];
    uint32_t g = 0u, g_end = 0u;
    for (;; g++) {
        if (g >= g_end) {
            g = wave_atomic_add(c_groups, span);
            if (g >= n_groups) break;
            g_end = g + span < n_groups ? g + span : n_groups;
        }
        uint32_t slot = kNullEntry;
        int32_t sh = -1, pr = -1;
        bool pending = false, fresh = false;
        if (g < g0) {
            const uint32_t i = g * 64u + lane;
            if (i < n0) {
                const ListEnt e = lists.ent[i];
                slot = e.slot;
                pending = slot != kNullEntry && (slot & kEntPending) != 0u;
                fresh = slot != kNullEntry && (slot & kEntFresh) != 0u;
                if (pending) {
                    sh = in.sh_prim[slot & kSlotMask];
                    pr = in.pr_prim[slot & kSlotMask];
                }
            }
        } else {
            const uint32_t i = (g - g0) * 64u + lane;
            if (i < n1) slot = fold_in[i];
            if (slot != kNullEntry) {  // a fold-only path is nothing but pending terms
                pending = true;
                sh = in.sh_prim[slot];
                pr = in.pr_prim[slot];
            }
        }
        const bool valid = slot != kNullEntry;
        RecRegs R{};
        rec_fetch<(FEAT & kFeatEnv) != 0>(stage, in, slot & kSlotMask, valid, pending, fresh, gen_first + ((slot & kSlotMask) - gen_slot), R);
        ShadeA a{};
        shade_a<FEAT, kKindNone>(sc, in, R, slot & kSlotMask, 0u, false, sh, pr, pending, valid, max_depth, a);
        if (a.live) film_put(lf, a.orig, a.L);
    }
}

// ---------------------------------------------------------------------- tail
// Once a lane's batch is exhausted and few paths are left, per-bounce launches are bound by the single
// longest ray of each launch (a few hundred dependent fetches), times the remaining bounces.  k_tail
// finishes those paths in ONE launch with the same closest_hit and the same shade_a / shade_b, ping-ponging a
// path's record between the two pools (same slot) until it retires.  Same arithmetic, same counters, no queues.
// Its paths are the queue's extension entries plus the fold list of the iteration it replaces.
// Persistent waves with (a) path replacement: a lane whose path has retired takes the next unfinished path
// (one atomic per wave and refill), so a wave does not sit on a register allocation for the sake of its one
// longest path; and (b) pooled rays: the pending rays of the wave's live paths (up to three each: shadow, probe,
// extension) are listed in LDS and dealt to ALL 64 lanes, so a path's three rays are traced side by side.
// The two pools alternate per bounce, wave-uniformly: new paths are taken on even passes only.
// COUNT: the instrumented build (RT_RENDER_COUNT_TRAVERSAL) also counts the tail's node / primitive tests; its rays
// are always counted (rt_stats.tail_*), so that the traversal kernel's own share is known exactly.
template <int FEAT, bool COUNT>
__global__ __launch_bounds__(256) void k_tail(DevScene sc, PathState buf0, PathState buf1, Ctl* ctl,
                                             uint32_t it_abs, uint32_t max_depth, const uint32_t* __restrict__ queue,
                                             Lists lists, f64_t* lf, DevStats* stats) {
    __shared__ int2 lds_stack[kLdsStack * 256];
    __shared__ uint32_t s_job[4][192];  // per wave: slot of the path | ray kind << 30
    __shared__ int2 s_res[4][192];      // per wave: {prim, hit word} found for job j
    const uint32_t ring = it_abs % kRing;
    const uint32_t n_q = ctl->n_rays[ring], n_total = n_q + ctl->fold_count[it_abs & 3u][0];
    const uint32_t* fold_in = lists.fold[it_abs & 1u];
    uint32_t* next_path = &ctl->head[ring];  // zero at launch: no k_trace runs in the tail iteration (k_plan cleared it)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t slot = 0;
    bool alive = false;
    bool list_done = false;  // wave-uniform
    RT_TRAV_STACK(ts, lds_stack)
    TravCount tc{0, 0, 0};
    unsigned long long n_r1 = 0, n_r2 = 0, n_r3 = 0, n_v = 0;
    uint32_t n_traced = 0;  // wave-uniform: rays this wave traced
    const int cur0 = (int)(it_abs & 1u);
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t pass = 0;; pass++) {
        const int cur = cur0 ^ (int)(pass & 1u);
        const PathState& in = cur ? buf1 : buf0;
        const PathState& out = cur ? buf0 : buf1;
        if (!(pass & 1u) && !list_done) {
            const unsigned long long md = __ballot(!alive);
            const uint32_t want = (uint32_t)__popcll(md);
            if (want) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(next_path, want);
                base = __shfl(base, 0, 64);
                const uint32_t idx = base + (uint32_t)__popcll(md & below);
                if (!alive && idx < n_total) {
                    // a path = an extension entry of the queue, or an entry of the fold list (shadow / probe entries and
                    // the null ends of chunks are skipped: the lane asks again on the next even pass)
                    uint32_t s = kNullEntry;
                    if (idx < n_q) {
                        const uint32_t e = queue[idx];
                        if ((e >> 30) == kRayExt) s = e & kSlotMask;
                    } else {
                        s = fold_in[idx - n_q];
                    }
                    if (s != kNullEntry) {
                        slot = s;
                        alive = true;
                    }
                }
                if (base + want >= n_total) list_done = true;
            }
        }
        // the state a path's owner lane wrote in the previous pass is read by other lanes of the wave below
        if (pass) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        uint32_t fl = 0;
        if (alive) fl = (uint32_t)(rec_words(in, slot)[kWMeta] >> 32);
        if (__ballot(alive) == 0ull) {
            if (list_done) break;
            continue;  // (nothing alive: the next even pass refills)
        }
        const bool has_sh = alive && (fl & kHasShadow), has_pr = alive && (fl & kHasProbe);
        const bool has_ex = alive && !(fl & kFoldOnly);
        const unsigned long long msh = __ballot(has_sh), mpr = __ballot(has_pr), mex = __ballot(has_ex);
        const uint32_t nsh = (uint32_t)__popcll(msh), npr = (uint32_t)__popcll(mpr), nex = (uint32_t)__popcll(mex);
        const uint32_t jsh = (uint32_t)__popcll(msh & below), jpr = nsh + (uint32_t)__popcll(mpr & below);
        const uint32_t jex = nsh + npr + (uint32_t)__popcll(mex & below);
        if (has_sh) s_job[wave][jsh] = slot | (kRayShadow << 30);
        if (has_pr) s_job[wave][jpr] = slot | (kRayProbe << 30);
        if (has_ex) s_job[wave][jex] = slot | (kRayExt << 30);
        wave_sync_lds();
        const uint32_t n_jobs = nsh + npr + nex;
        n_traced += n_jobs;
        for (uint32_t j = lane; j < n_jobs; j += 64u) {
            const uint32_t job = s_job[wave][j];
            const uint32_t js = job & kSlotMask, kind = job >> 30;
            const D3 o = ldr(in, kRayO, js);
            double t;
            uint32_t hs = 0;
            int32_t prim;
            if (kind == kRayShadow) {  // Visibility::unoccluded, hittable.rs:25-32
                const D3 d = ldr(in, kRaySp, js) - o;
                prim = closest_hit<COUNT>(sc, o + d * kSmall, d, 0.0, kInf, t, ts, &tc);
            } else if (kind == kRayProbe) {
                prim = closest_hit<COUNT>(sc, o, ldr(in, kRayPd, js), kSmall, kInf, t, ts, &tc);
            } else {
                prim = closest_hit<COUNT>(sc, o, ldr(in, kRayD, js), kSmall, kInf, t, ts, &tc, &hs);
            }
            s_res[wave][j] = make_int2(prim, prim < 0 ? 0 : (int)hit_word(prim, hs));
        }
        wave_sync_lds();
        if (alive) {
            const int32_t sh = has_sh ? s_res[wave][jsh].x : -1, pr = has_pr ? s_res[wave][jpr].x : -1;
            int2 hr = make_int2(-1, 0);
            if (has_ex) hr = s_res[wave][jex];
            RecRegs R{};
            rec_load(in, slot, true, R);
            ShadeA a{};
            shade_a<FEAT, kKindAny>(sc, in, R, slot, (uint32_t)hr.y, hr.x >= 0, sh, pr, true, true, max_depth, a);
            ShadeRes r{};
            if (a.will_shade) {
                r = shade_b<FEAT>(sc, out, slot, a);
                rec_store_direct(out, slot, a, r);
                n_v++;
            }
            n_r1 += r.emit_ext ? 1u : 0u;
            n_r2 += r.emit_sh ? 1u : 0u;
            n_r3 += r.emit_pr ? 1u : 0u;
            if (!r.keep) {
                film_put(lf, a.orig, a.L);
                alive = false;
            }
        }
    }
    DevStats* sh = stat_shard(stats);
    if (n_r1) atomicAdd(&sh->r1, n_r1);
    if (n_r2) atomicAdd(&sh->r2, n_r2);
    if (n_r3) atomicAdd(&sh->r3, n_r3);
    if (n_v) atomicAdd(&sh->vertices, n_v);
    if (lane == 0 && n_traced) atomicAdd(&sh->tail_rays, (unsigned long long)n_traced);
    if (COUNT) {
        atomicAdd(&sh->nodes, (unsigned long long)tc.nodes);
        atomicAdd(&sh->tris, (unsigned long long)tc.tris);
        atomicAdd(&sh->others, (unsigned long long)tc.others);
        atomicAdd(&sh->tail_nodes, (unsigned long long)tc.nodes);
        atomicAdd(&sh->tail_tris, (unsigned long long)tc.tris);
        atomicAdd(&sh->tail_others, (unsigned long long)tc.others);
    }
}

#if defined(RT_KERNELS_CORE) && !defined(RT_F32)  // the film is f64 in both modes: compiled once
// ------------------------------------------------------------------- resolve
// util::increment_color order: each pixel's samples are added one by one, in sample order.
__global__ __launch_bounds__(256) void k_resolve(const double* __restrict__ lf, ChunkDesc ck,
                                                 const uint32_t* __restrict__ pix_list, double* rgb_sum, uint32_t* n) {
    const uint32_t p_local = blockIdx.x * blockDim.x + threadIdx.x;
    if (p_local >= ck.n_pixels) return;
    const uint32_t pix = pix_list[ck.pixel_base + p_local];
    double r = rgb_sum[(size_t)pix * 3 + 0], g = rgb_sum[(size_t)pix * 3 + 1], b = rgb_sum[(size_t)pix * 3 + 2];
    for (uint32_t s = 0; s < ck.n_samples; s++) {
        const double* q = lf + (size_t)(s * ck.n_pixels + p_local) * 3;  // kernels.hip: film_put
        r += q[0];
        g += q[1];
        b += q[2];
    }
    rgb_sum[(size_t)pix * 3 + 0] = r;
    rgb_sum[(size_t)pix * 3 + 1] = g;
    rgb_sum[(size_t)pix * 3 + 2] = b;
    n[pix] += ck.n_samples;
}

// util.rs:400-408, 441-471: film -> ACES approx -> gamma 2.2 -> 8 bit (next-row f1)
__global__ __launch_bounds__(256) void k_tonemap(const double* __restrict__ rgb_sum, const uint32_t* __restrict__ n,
                                                 uint64_t npix, uint8_t* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const double scale = 1.0 / (double)n[i];
    for (int c = 0; c < 3; c++) {
        double x = rgb_sum[i * 3 + c] * scale;
        x = x * 0.6;
        x = clampd((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14), 0.0, 1.0);
        double v = dm_pow(x, 1.0 / 2.2) * 256.0;
        v = __builtin_round(v);
        out[i * 3 + c] = (uint8_t)(v != v ? 0.0 : (v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v)));
    }
}

#endif

}  // namespace rtd

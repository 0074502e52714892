// kernels.hip -- the wavefront path-tracing kernels for gfx950 (wave64).
//
// A "batch" = up to kBatchMax camera samples of one block of pixels (the whole render when it
// fits).  Each of the two lanes keeps up to `paths_in_flight` paths alive and runs, per iteration:
//   k_plan + k_generate  top the lane's pool up with the batch's next camera samples (streaming
//            regeneration: every launch stays full until the batch's single final drain)
//   k_trace  persistent waves pull 64-ray batches from the ray queue (one atomic per
//            wave) and run the closest-hit traversal for the three root call sites
//            of SURVEY.md 3.2: R1 extension (integrator.rs:388), R2 shadow
//            (hittable.rs:25-39, Q13: closest hit, compared by prim index), R3 MIS
//            probe (integrator.rs:615).  Output: one prim index per ray.
//   k_shade  one lane per live path: folds the previous vertex's direct-light terms
//            using the R2/R3 results, rebuilds the winning hit's record, applies the
//            emitted-light rule, builds the BSDF, samples one light + MIS
//            (integrator.rs:530-659), samples the continuation, Russian roulette
//            (integrator.rs:375-445), then compacts survivors and their rays into the
//            next queues with __ballot/__popcll prefix sums (one atomic per wave).
// Path state is PHYSICALLY compacted every bounce: k_shade reads slot i of buffer X[it&1] and
// writes survivors to consecutive slots of X[(it+1)&1], so every state access of every kernel
// is a dense, coalesced SoA stream however few paths survive (a sparse in-place SoA costs a
// whole 64-B HBM atom per 8-B field).  A retired path drops its radiance into lfinal[orig].
// The R2/R3 results never influence control flow or RNG draws of the path, only
// additions into L, which is why they can be traced one iteration late.
// Film: k_resolve sums each pixel's samples in sample order in f64 -- the order of
// util::increment_color (util.rs:208-232) -- so images are bit-reproducible and
// independent of how tiles are split over GPUs.
#include "shading.h"

#ifndef RT_SHADE_BLOCK
#define RT_SHADE_BLOCK 256  // threads per k_shade block = the paths dealt among its waves by class.  Measured (profiles/r03_exp_shade_block.txt,
                            // C4 / C3 k_shade ms): 64: 923 / 123, 128: 644 / 95.4, 192: 613 / 93.6, 256: 616 / 95.0, 384: 759 / 115 -- and a dealing window of 2-8
                            // blocks (tools/experiments/r03_deal_window.patch): 787-1026 / 114-146, the gathered state loads cost more than the
                            // purer waves save
#endif
#ifndef RT_SHADE_LAST_WAVE
#define RT_SHADE_LAST_WAVE 1  // k_shade (dealing instances): no barrier before the queue reservation, the block's last wave writes it
#endif
#ifndef RT_QUEUE_BY_KIND
#define RT_QUEUE_BY_KIND 1  // k_shade queues a block's rays kind by kind (0: wave by wave)
#endif
#ifndef RT_XCD_QUEUE
#define RT_XCD_QUEUE 1  // the ray queue in eight parts, one per XCD (k_trace); 0 = one head for all waves
#endif

namespace rtd {

// Ctl, BatchCtl, MirrorEntry, ChunkDesc, TraceTune, kRing: scene_dev.h (shared with the f32 kernels)

#ifdef RT_F32
// Fast mode: the path state keeps its f64-sized slots (the launch schedule and the pools are shared with the parity
// mode) but holds binary32 values in the low half of each slot -- no v_cvt_f32_f64 / v_cvt_f64_f32 per field and bounce.
RTD D3 ld3(const f64_t* x, const f64_t* y, const f64_t* z, uint32_t i) {
    return d3(reinterpret_cast<const float*>(x)[2u * i], reinterpret_cast<const float*>(y)[2u * i],  // RT_KEEP_F64
              reinterpret_cast<const float*>(z)[2u * i]);                                            // RT_KEEP_F64
}
RTD void st3(f64_t* x, f64_t* y, f64_t* z, uint32_t i, D3 v) {
    reinterpret_cast<float*>(x)[2u * i] = v.x;  // RT_KEEP_F64
    reinterpret_cast<float*>(y)[2u * i] = v.y;  // RT_KEEP_F64
    reinterpret_cast<float*>(z)[2u * i] = v.z;  // RT_KEEP_F64
}
#else
RTD D3 ld3(const f64_t* x, const f64_t* y, const f64_t* z, uint32_t i) { return d3(x[i], y[i], z[i]); }
RTD void st3(f64_t* x, f64_t* y, f64_t* z, uint32_t i, D3 v) {
    x[i] = v.x;
    y[i] = v.y;
    z[i] = v.z;
}
#endif

// Wave-uniform fetch-and-add on the SCALAR memory path (s_atomic_add, gfx9 family incl. gfx950; checked against
// vector atomics on the same word by tools/experiments/satomic_test.hip).  Its return travels through lgkmcnt, so
// the wave does not wait for its vector stores in flight (a vector atomic's return shares vmcnt with them and comes
// back in order behind them).  Executes once per wave whatever EXEC is: call it from wave-uniform control flow only.
// Not for the per-block counters of k_shade: there the scalar path is slower (C2 +8 % frame time) -- many more
// atomics per launch on one word, and their return was not what the block waits for.
RTD uint32_t wave_atomic_add(uint32_t* p, uint32_t v) {
    uint32_t ret = __builtin_amdgcn_readfirstlane(v);
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(ret) : "s"(p) : "memory");
    return ret;
}
// wave-level compaction: lanes with pred append `value` to out[]; one atomic per wave
RTD void wave_append(bool pred, uint32_t value, uint32_t* out, uint32_t* counter) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader, 64);
    if (pred) out[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = value;
}
// wave-level allocation: lanes with pred get consecutive indices from *counter (one atomic per wave)
RTD uint32_t wave_alloc(bool pred, uint32_t* counter) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return 0u;
    const uint32_t lane = threadIdx.x & 63u;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader, 64);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}
RTD void wave_count(bool pred, unsigned long long* counter) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    if ((int)lane == __ffsll((long long)mask) - 1) atomicAdd(counter, (unsigned long long)__popcll(mask));
}
// Same-address atomics saturate near 88 per microsecond on this chip, so the hot counters are
// (a) aggregated per 256-thread block through LDS (one atomic per block) and (b) for the statistics,
// spread over kStatShards cache lines that the host sums.
RTD DevStats* stat_shard(DevStats* stats) { return stats + (blockIdx.x & (kStatShards - 1)); }

// ------------------------------------------------------------------ generate
#ifndef RT_F32  // (precision-independent: compiled once, in the f64 namespace)
// k_plan (one thread): how many camera samples this lane starts now = free pool slots, limited by
// what the batch still holds; reserves them from the shared batch counter and appends them to the
// path list / ray queue of iteration `it`.  Also clears the ring entries of iteration it+2.
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
    ctl->n_active[r] = live + (uint32_t)want;
    ctl->n_rays[r] += (uint32_t)want;
    const uint32_t z = (it + 2) % kRing;
    ctl->n_active[z] = 0;
    ctl->n_rays[z] = 0;
    ctl->head[z] = 0;
    for (int x = 0; x < 8; x++) ctl->xhead[(it + 2) & 3u][x][0] = 0;
    if (want) {
        atomicAdd(&stats->paths, want);
        atomicAdd(&stats->r1, want);
    }
}

#endif

// integrator.rs:357-366 + sampler.rs:606-613 + geometry.rs:177-190 (+ util.rs:105-113)
__global__ __launch_bounds__(256) void k_generate(PathState st, rt_camera cam, ChunkDesc ck,
                                                  const uint32_t* __restrict__ pix_list, uint32_t* queue,
                                                  const Ctl* ctl) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ctl->gen_count) return;
    const uint32_t g = ctl->gen_first + idx;  // path index inside the batch = film staging slot
    const uint32_t slot = ctl->gen_slot + idx;
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
    st3(st.ox, st.oy, st.oz, slot, origin + offset);
    st3(st.dx, st.dy, st.dz, slot, dir - offset);
    // (beta = 1 and L = 0 are not written: kFresh tells shade_a -- 48 of this kernel's 116 B per sample, and the kernel
    // is bound by its HBM writes)
    st.rng[slot] = rng;
    st.flags[slot] = kFresh;
    st.orig[slot] = g;
    queue[ctl->gen_q + idx] = slot | (kRayExt << 30);
}

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
// Every wave leaves the loop once the queue is exhausted and its own lanes are done.
template <bool COUNT, bool SIMPLE>
__global__ __launch_bounds__(256, RT_TRACE_WAVES) void k_trace(DevScene sc, PathState st, const uint32_t* __restrict__ queue,
                                               Ctl* ctl, uint32_t it_abs, DevStats* stats, TraceTune tune,
                                               MirrorEntry* mirror, uint32_t seq, const BatchCtl* batch,
                                               unsigned long long batch_total) {
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
    // a block only joins the work-pulling loop if the queue can give it at least one batch:
    // tail iterations with a handful of rays then cost a launch, not a grid of atomics
    if (sc.n_nodes == 0) {  // a scene without primitives (an environment only): every query misses
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
            const uint32_t e = queue[i];
            const uint32_t slot = e & kSlotMask, kind = e >> 30;
            if (kind == kRayExt)
                st.hit_prim[slot] = -1;
            else if (kind == kRayShadow)
                st.sh_prim[slot] = -1;
            else
                st.pr_prim[slot] = -1;
        }
        return;
    }
    if (blockIdx.x * 256u >= n) return;
    const uint32_t lane = threadIdx.x & 63u;
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
    uint32_t slot_kind = 0;
    uint32_t res_next = 0, res_end = 0;  // wave-uniform: [res_next, res_end) is reserved for this wave
    uint32_t res_base = 0, q_lo = 0, q_hi = 0;  // start of the reservation; its queue entries, two per lane
    // One atomic on the queue head hands a wave `reserve` entries.  128 is the measured optimum: 64 costs 30 % of the
    // kernel's time (twice the atomic -> queue -> ray-data chains); larger reservations served through a sliding
    // 128-entry window were 5-10 % slower (the extra code in the refill path, and neighbouring queue entries are
    // neighbouring paths: small reservations let concurrent waves share their nodes in L2).
    // Small queues: shrink the reservation so that the tail still spreads over the waves.
    const uint32_t n_waves = gridDim.x * 4u;
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
            if (wb) {
                const uint32_t slot = slot_kind & kSlotMask, kind = slot_kind >> 30;
                if (kind == kRayExt) {
                    st.hit_prim[slot] = tv.best_prim;
                    st.hit_slot[slot] = tv.best_slot;
                } else if (kind == kRayShadow)
                    st.sh_prim[slot] = tv.best_prim;
                else
                    st.pr_prim[slot] = tv.best_prim;
                wb = false;
            }
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
                q_lo = base + lane < res_end ? queue[base + lane] : 0u;
                q_hi = base + 64u + lane < res_end ? queue[base + 64u + lane] : 0u;
            }
            if (!exhausted) {
                const uint32_t base = res_next;
                const uint32_t take = (uint32_t)n_idle < res_end - res_next ? (uint32_t)n_idle : res_end - res_next;
                res_next += take;
                const uint32_t my = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                const uint32_t idx = base + my;
                const uint32_t rel = idx - res_base;  // < 128 for the lanes that take an entry
                const uint32_t e_lo = __shfl(q_lo, (int)(rel & 63u), 64), e_hi = __shfl(q_hi, (int)(rel & 63u), 64);
                if (!has_ray && my < take) {
                    const uint32_t e = rel < 64u ? e_lo : e_hi;
                    const uint32_t slot = e & kSlotMask, kind = e >> 30;
                    D3 o = ld3(st.ox, st.oy, st.oz, slot);
                    D3 d;
                    double tmin = kSmall;
                    if (kind == kRayExt) {
                        d = ld3(st.dx, st.dy, st.dz, slot);
                    } else if (kind == kRayShadow) {  // Visibility::unoccluded, hittable.rs:25-32
                        d = ld3(st.spx, st.spy, st.spz, slot) - o;
                        o = o + d * kSmall;
                        tmin = 0.0;
                    } else {
                        d = ld3(st.pdx, st.pdy, st.pdz, slot);
                    }
                    trav_init(tv, sc, o, d, tmin, kInf);
                    slot_kind = e;
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

// --------------------------------------------------------------------- shade
// The per-vertex work, split where k_shade needs a block barrier (output-slot allocation):
//   shade_a: fold the previous vertex's direct light (estimate_direct's two additions, using the R2/R3
//            results), rebuild the record of the vertex the extension ray found, emitted-light rule.
//   shade_b: BSDF, one-light NEE + MIS (integrator.rs:530-634), continuation sample, Russian roulette
//            (integrator.rs:421-442); writes the survivor's state to slot `os` of `out`.
// k_shade runs them for one bounce of every path; k_tail loops them per lane until the path retires.
// Diagnostic build (-DRT_SHADE_PROF via tools/variants.sh, selected with RT_AMD_LIB): wave clock per section of
// the shading code, printed by rt_render.  Note that time at the block barriers (sections "alloc", "queue") is
// the wait for the slowest wave of the block, not issue time.
#ifdef RT_SHADE_PROF
__device__ unsigned long long g_shade_prof[16];
struct ShadeProf {
    unsigned long long acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last = 0;
};
#define RT_PROF_DECL ShadeProf prof_; prof_.last = __builtin_readcyclecounter();
#define RT_PROF_ARG , ShadeProf& prof_
#define RT_PROF_PASS , prof_
#define RT_PROF(k)                                                 \
    {                                                              \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        prof_.acc[k] += now_ - prof_.last;                         \
        prof_.last = now_;                                         \
    }
#define RT_PROF_FLUSH                                                                       \
    for (int k_ = 0; k_ < 10; k_++) {                                                       \
        unsigned long long v_ = prof_.acc[k_];                                              \
        for (int o_ = 32; o_ > 0; o_ >>= 1) {                                               \
            const unsigned long long w_ = __shfl_xor(v_, o_, 64);                           \
            v_ = w_ > v_ ? w_ : v_;                                                         \
        }                                                                                   \
        if ((threadIdx.x & 63u) == 0 && v_) atomicAdd(&g_shade_prof[k_], v_);               \
    }
#else
#define RT_PROF_DECL
#define RT_PROF_ARG
#define RT_PROF_PASS
#define RT_PROF(k)
#define RT_PROF_FLUSH
#endif

struct ShadeA {
    D3 L, o, d, beta;
    HitRec rec;
    uint32_t fl, bounces;
    uint32_t orig;  // film staging slot of the path
    uint64_t rng;
    bool live, spec, will_shade;
};

template <int FEAT>
RTD void shade_a(const DevScene& sc, const PathState& in, uint32_t slot, bool valid, uint32_t max_depth, ShadeA& a RT_PROF_ARG) {
    // Every field of the slot this function may need is fetched up front, unconditionally: the loads are
    // independent and coalesced, so they cost one memory latency instead of one per branch level below.
    const uint32_t ls = valid ? slot : 0u;  // (slot 0 always exists: the block would have exited otherwise)
    const uint32_t fl_raw = in.flags[ls];
    const D3 l_in = ld3(in.lx, in.ly, in.lz, ls), o_in = ld3(in.ox, in.oy, in.oz, ls);
    const D3 d_in = ld3(in.dx, in.dy, in.dz, ls), beta_in = ld3(in.bx, in.by, in.bz, ls);
    const int32_t hp_in = in.hit_prim[ls];
    const uint32_t hs_in = in.hit_slot[ls];
#ifdef RT_SHADE_COND_NEE
    // (experiment) the pending-light block (80 of the slot's 244 bytes) only for slots that carry pending terms:
    // a fresh camera sample and a path whose last vertex found no light to sample have none
    int32_t sh_in = -1, pp_in = -1;
    D3 a_in = black(), q_in = black(), k_in = black();
    if (valid && (fl_raw & (kHasShadow | kHasProbe)) && !(fl_raw & kDead)) {
        sh_in = in.sh_prim[ls];
        pp_in = in.pr_prim[ls];
        a_in = ld3(in.ax, in.ay, in.az, ls);
        q_in = ld3(in.qx, in.qy, in.qz, ls);
        k_in = ld3(in.kx, in.ky, in.kz, ls);
    }
#else
    const int32_t sh_in = in.sh_prim[ls], pp_in = in.pr_prim[ls];
    const D3 a_in = ld3(in.ax, in.ay, in.az, ls), q_in = ld3(in.qx, in.qy, in.qz, ls), k_in = ld3(in.kx, in.ky, in.kz, ls);
#endif
    a.rng = in.rng[ls];
    a.orig = in.orig[ls];
    a.fl = valid ? fl_raw : kDead;
    const uint32_t fl = a.fl;
    a.live = valid && !(fl & kDead);
    a.L = black();
    a.o = black();
    if (a.live) {
        a.L = (fl & kFresh) ? black() : l_in;
        a.o = o_in;
        // ---- fold the previous vertex's direct lighting
        if (fl & (kHasShadow | kHasProbe)) {
            const uint32_t light_idx = fl >> kLightShift;
            const rt_light& lt = sc.lights[light_idx];
            const bool infinite = ((FEAT & kFeatEnv) != 0) && lt.kind == RT_LIGHT_INFINITE;
            D3 ld = black();
            if (fl & kHasShadow) {
                // Visibility::unoccluded(infinite): an area light must be the closest hit, the environment needs a miss
                const int32_t sh = sh_in;
                if (infinite ? sh < 0 : sh == (int32_t)lt.prim_index) ld = ld + a_in;
            }
            if (fl & kHasProbe) {
                const int32_t pp = pp_in;
                if (infinite) {
                    // integrator.rs:617-630: an escaped probe sees light.le(ray), already folded into q by shade_b
                    if (pp < 0) ld = ld + q_in;
                } else if (pp >= 0) {
                    const int32_t li = sc.prims[pp].light_index;
                    if (li >= 0 && (uint32_t)li == light_idx) {
                        const D3 pd = ld3(in.pdx, in.pdy, in.pdz, slot);
                        const rt_primitive& lpr = sc.prims[pp];
                        if (lpr.kind >= RT_PRIM_XY_RECT && lpr.xform_index < 0) {
                            // axis-aligned rect emitter: the record's normal faces the ray (set_front), so
                            // dot(n, -wi) = |wi's component along the rect's axis| and Light::l sees the colour
                            // unless that component is zero -- no need to build the record
                            double t, ra, rb;
                            D3 to, td;
                            if (rect_core(sc, lpr, a.o, pd, kSmall, kInf, t, ra, rb, to, td)) {
                                const rt_light& lt2 = sc.lights[li];
                                const bool lit = lt2.two_sided || absd(rect_axis_comp(lpr.kind, pd)) > 0.0;
                                if (lit && !is_black(d3(lt2.color[0], lt2.color[1], lt2.color[2]))) ld = ld + q_in;
                            }
                        } else {
                            HitRec nh;
                            if (prim_intersects(sc, pp, a.o, pd, kSmall, kInf, nh)) {
                                const D3 col = light_l(sc.lights[li], nh.n, -pd);  // new_record.le(-wi)
                                if (!is_black(col)) ld = ld + q_in;
                            }
                        }
                    }
                }
            }
            a.L = a.L + cmul(ld * (double)sc.n_lights, k_in);
        }
    }
    RT_PROF(0)
    // ---- the vertex found by the extension ray
    const bool active = a.live && !(fl & kFoldOnly);
    a.d = black();
    a.beta = black();
    bool is_some = false;
    a.bounces = fl & kBounceMask;
    a.spec = (fl & kSpecular) != 0;
    if (active) {
        const int32_t hp = hp_in;
        a.d = d_in;
        a.beta = (fl & kFresh) ? white() : beta_in;
        is_some = hp >= 0;
        if (is_some) is_some = hit_record(sc, hp, hs_in, a.o, a.d, kSmall, kInf, a.rec);
        RT_PROF(1)
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
    RT_PROF(2)
}

struct ShadeOut {
    bool emit_ext, emit_sh, emit_pr, keep;
};

// Precondition: a.will_shade.  `os` = slot of `out` reserved for this vertex.
template <int FEAT>
RTD ShadeOut shade_b(const DevScene& sc, const PathState& in, const PathState& out, uint32_t slot, uint32_t os,
                     ShadeA& a RT_PROF_ARG) {
    const HitRec& rec = a.rec;
    D3 beta = a.beta;
    uint32_t bounces = a.bounces;
    bool spec = a.spec;
    uint64_t rng = a.rng;
    Bsdf bsdf;
    compute_scattering<FEAT>(sc, rec, bsdf);
    RT_PROF(3)
    bool has_sh = false, has_pr = false;
    uint32_t light_num = 0;
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
        D3 sp, sn;
        double light_pdf;
        D3 wi, color;
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
                st3(out.ax, out.ay, out.az, os, cmul(f, color) * (weight / light_pdf));
                st3(out.spx, out.spy, out.spz, os, sp);
            }
        }
        RT_PROF(4)
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
                    st3(out.qx, out.qy, out.qz, os, is_black(pcol) ? black() : cmul(f2, pcol) * (weight / spdf));
                    st3(out.pdx, out.pdy, out.pdz, os, wi2);
                }
            }
        }
        if (has_sh || has_pr) st3(out.kx, out.ky, out.kz, os, beta);
        RT_PROF(5)
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
    RT_PROF(6)
    ShadeOut r;
    r.emit_ext = cont;
    r.emit_sh = has_sh;
    r.emit_pr = has_pr;
    r.keep = cont || has_sh || has_pr;
    if (r.keep) {
        st3(out.ox, out.oy, out.oz, os, rec.p);  // spawn_ray: origin = hit point (Q4)
        st3(out.lx, out.ly, out.lz, os, a.L);
        if (cont) {
            st3(out.dx, out.dy, out.dz, os, wi);
            st3(out.bx, out.by, out.bz, os, beta);
            out.rng[os] = rng;
        }
        out.orig[os] = a.orig;
        out.flags[os] = (bounces & kBounceMask) | (spec ? kSpecular : 0u) | (cont ? 0u : kFoldOnly) |
                        (has_sh ? kHasShadow : 0u) | (has_pr ? kHasProbe : 0u) | (light_num << kLightShift);
    } else {
        out.flags[os] = kDead;
    }
    RT_PROF(7)
    return r;
}

#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 2
#endif
#ifndef RT_SHADE4_MAXFEAT
#define RT_SHADE4_MAXFEAT (-1)  // (experiment) instances up to this mask compiled for 4 waves/SIMD
#endif
#ifndef RT_SHADE3_MAXFEAT
#define RT_SHADE3_MAXFEAT 3  // instances up to this feature mask are compiled for 3 waves/SIMD (measured: the
                             // two-lobe and row-f4 instances spill too much to gain from it)
#endif
#ifndef RT_SORT_CLASSES
#define RT_SORT_CLASSES 5
#endif
#ifndef RT_SORT_FEAT0
#define RT_SORT_FEAT0 0  // (experiment) the Lambert-only instance deals its paths too
#endif
#undef RT_SHADE_BOUND
#define RT_SHADE_BOUND_RULE(F) ((F) <= RT_SHADE4_MAXFEAT ? 4 : (feat_three_waves(F) && RT_SHADE3_MAXFEAT >= 3 ? 3 : RT_SHADE_WAVES))
#ifdef RT_F32
#ifndef RT_F32_SHADE_WAVES
#define RT_F32_SHADE_WAVES 4  // the binary32 single-lobe instances need 127-138 VGPRs: 4 waves/SIMD with a few spills
                              // (measured: C4 k_shade 628 -> 553 ms, C3 90.6 -> 78.6; at 5 waves 791 / 91.8)
#endif
#define RT_SHADE_BOUND(F) (feat_three_waves(F) ? RT_F32_SHADE_WAVES : RT_SHADE_BOUND_RULE(F))
#else
#define RT_SHADE_BOUND(F) RT_SHADE_BOUND_RULE(F)
#endif
template <int FEAT>
__global__ __launch_bounds__(RT_SHADE_BLOCK, RT_SHADE_BOUND(FEAT)) void k_shade(DevScene sc, PathState in, PathState out, Ctl* ctl, uint32_t it_abs,
                                               uint32_t max_depth, uint32_t* queue_out, f64_t* lfx, f64_t* lfy,
                                               f64_t* lfz, DevStats* stats) {
    const uint32_t it = it_abs % kRing, itn = (it_abs + 1) % kRing;
    const uint32_t n_active = ctl->n_active[it];
    // whole block past the end: nothing to do (block-uniform exit: the block barriers below stay well-formed)
    if (blockIdx.x * blockDim.x >= n_active) return;
    // (Giving every XCD one contiguous eighth of the path list here, as k_trace does with the ray queue, was measured
    // and is 11-21 % slower for this kernel: C4 717 -> 871 ms, C3 109 -> 122.)
    const uint32_t bid = blockIdx.x;
    constexpr uint32_t kSW = RT_SHADE_BLOCK / 64;  // waves per block
    __shared__ uint32_t s_cnt[kSW][4];  // [wave][0 = output slots, 1..3 = ext / shadow / probe rays]
    __shared__ uint32_t s_base[2];    // block's base in the next path list / ray queue
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // Deal the block's 256 paths to its lanes by what their vertex needs (mesh hit / sphere-rect hit / escaped /
    // fold only / nothing), so that the lanes of a wave run the same branches: the shading kernels are VALU-bound
    // at 24-35 active lanes of 64 (PMC).  Paths are independent and the film staging is indexed by
    // (pixel, sample), so the order inside a block changes no result.  Scenes with Lambertian materials only
    // (FEAT == 0, e.g. C2) have nothing to separate and skip it; C3 +3.8 %, C4 +2.7 %, material_hdr(1) +2.9 %.
    uint32_t slot = bid * blockDim.x + threadIdx.x;
    if (FEAT != 0 || RT_SORT_FEAT0) {
        constexpr uint32_t kCls = RT_SORT_CLASSES;
        __shared__ uint32_t s_cls[kSW][kCls];
        __shared__ uint16_t s_perm[RT_SHADE_BLOCK];
        uint32_t key = kCls - 1u;
        if (slot < n_active) {
            const uint32_t fl0 = in.flags[slot];
            if (!(fl0 & kDead)) {
                if (fl0 & kFoldOnly)
                    key = kCls - 2u;
                else if (in.hit_prim[slot] < 0)
                    key = kCls - 3u;
                else
                {
                    const uint32_t hs0 = in.hit_slot[slot];
                    if (hs0 & kLeafOther)
                        key = kCls - 4u;
                    else  // mesh hit: with more than five classes also by material (two dragons of different materials)
                        key = kCls > 5u ? (sc.leaf_meta[hs0].mat_flags & kMetaMatMask) % (kCls - 4u) : 0u;
                }
            }
        }
        uint32_t rank = 0;
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (uint32_t c = 0; c < kCls; c++) {
            const unsigned long long m = __ballot(key == c);
            if (key == c) rank = (uint32_t)__popcll(m & below);
            if (lane == 0) s_cls[wave][c] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        uint32_t dest = rank;
#pragma unroll
        for (uint32_t c = 0; c < kCls; c++) {
#pragma unroll
            for (uint32_t w = 0; w < kSW; w++) {
                const uint32_t cnt = s_cls[w][c];
                if (c < key || (c == key && w < wave)) dest += cnt;
            }
        }
        s_perm[dest] = (uint16_t)threadIdx.x;
        __syncthreads();
        slot = bid * blockDim.x + s_perm[threadIdx.x];
    }
    ShadeA a;
    RT_PROF_DECL
    shade_a<FEAT>(sc, in, slot, slot < n_active, max_depth, a RT_PROF_PASS);
    // a vertex that will be shaded gets its output slot now (dense, block-contiguous); if the path
    // then ends without pending light terms the slot is marked dead and skipped next bounce
    uint32_t os;
    bool leader = threadIdx.x == 0;
    __shared__ uint32_t s_alive[kSW];
    __shared__ uint32_t s_done;            // waves of the block that have staged their rays
    __shared__ uint32_t s_stage[kSW][3][64];  // [wave][kind][rank]: queue entries waiting for the block's reservation
    {   // one atomic per block: wave counts -> LDS -> block base -> per-lane slot
        const unsigned long long m = __ballot(a.will_shade);
        // A wave without a vertex to shade -- escaped / fold-only / dead paths, which the dealing step gathers into
        // whole waves -- writes its film values, reports zero counts and ENDS instead of parking at the block's two
        // barrier pairs (a barrier waits only for the waves that are still alive).  Parked waves hold their 168
        // registers and a wave slot while the block's other waves run shade_b: C4 (42 % of the slots are escaped
        // paths) k_shade 741 -> 658 ms, C3 110.6 -> 101.3; the Lambert-only instance does not deal and sees nothing.
        if (m == 0ull) {
            if (a.live) {
                const uint32_t og = a.orig;
                lfx[og] = a.L.x;
                lfy[og] = a.L.y;
                lfz[og] = a.L.z;
            }
            if (lane == 0) {
                s_cnt[wave][0] = 0u;
                s_cnt[wave][1] = 0u;
                s_cnt[wave][2] = 0u;
                s_cnt[wave][3] = 0u;
                s_alive[wave] = 0u;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");  // the LDS writes have landed before the wave ends
            RT_PROF(8)
            RT_PROF_FLUSH
            return;
        }
        if (lane == 0) {
            s_cnt[wave][0] = (uint32_t)__popcll(m);
            s_alive[wave] = 1u;
        }
        __syncthreads();
        // the block's leader is lane 0 of its first wave that is still alive
        leader = lane == 0;
        for (uint32_t w = 0; w < wave; w++) leader = leader && s_alive[w] == 0u;
        if (leader) {
            uint32_t tot = 0;
            for (uint32_t w = 0; w < kSW; w++) tot += s_cnt[w][0];
            s_base[0] = tot ? atomicAdd(&ctl->n_active[itn], tot) : 0u;
            s_done = 0u;
        }
        __syncthreads();
        uint32_t off = s_base[0];
        for (uint32_t w = 0; w < wave; w++) off += s_cnt[w][0];
        os = off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
    RT_PROF(8)
    ShadeOut r{false, false, false, false};
    if (a.will_shade) r = shade_b<FEAT>(sc, in, out, slot, os, a RT_PROF_PASS);
    if (a.live && !r.keep) {  // retired: its radiance goes to the film staging slot of (pixel, sample)
        const uint32_t og = a.orig;
        lfx[og] = a.L.x;
        lfy[og] = a.L.y;
        lfz[og] = a.L.z;
    }
    // ---- rays of the next bounce: one queue reservation per block, laid out as
    // [extension][shadow][probe]; statistics: one atomic per block and counter, on this block's shard
    {
        const unsigned long long me = __ballot(r.emit_ext), ms = __ballot(r.emit_sh), mp = __ballot(r.emit_pr);
        const uint32_t ce = (uint32_t)__popcll(me), cs = (uint32_t)__popcll(ms), cp = (uint32_t)__popcll(mp);
        if (lane == 0) {
            s_cnt[wave][1] = ce;
            s_cnt[wave][2] = cs;
            s_cnt[wave][3] = cp;
        }
        // No barrier in front of the queue reservation (instances with class dealing): a wave stages its entries in LDS
        // and ENDS; the block's last wave to arrive (an LDS counter) reserves the queue range and writes everybody's
        // entries.  A wave that has finished shade_b no longer holds its registers until the block's slowest wave has:
        // k_shade -2.3 % on C4 and C3; the Lambert-only instance (waves of equal length) loses 2 % and keeps the barrier.
        if (RT_SHADE_LAST_WAVE && FEAT != 0) {
            const unsigned long long below0 = (1ull << lane) - 1ull;
            if (r.emit_ext) s_stage[wave][0][__popcll(me & below0)] = os | (kRayExt << 30);
            if (r.emit_sh) s_stage[wave][1][__popcll(ms & below0)] = os | (kRayShadow << 30);
            if (r.emit_pr) s_stage[wave][2][__popcll(mp & below0)] = os | (kRayProbe << 30);
            uint32_t n_part = 0;
            for (uint32_t w = 0; w < kSW; w++) n_part += s_alive[w];
            // release (every lane: all of them staged entries): this wave's s_stage / s_cnt writes are visible before it is
            // counted; acquire: the last arriver reads the other waves' entries only after it has seen their counts.
            // Workgroup scope, LDS address space only -- the wave's global stores need not have landed.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            uint32_t prev = 0;
            if (lane == 0) prev = __hip_atomic_fetch_add(&s_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            prev = (uint32_t)__builtin_amdgcn_readfirstlane((int)prev);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            RT_PROF(9)
            if (prev + 1u < n_part) {
                RT_PROF_FLUSH
                return;
            }
            // last wave: totals, one reservation, all entries kind by kind ([ext of the waves][shadow ...][probe ...])
            uint32_t cnt[3][kSW], tk[3] = {0, 0, 0}, tv = 0;
            for (uint32_t w = 0; w < kSW; w++) {
                const bool al = s_alive[w] != 0u;
                for (uint32_t k = 0; k < 3; k++) {
                    cnt[k][w] = al ? s_cnt[w][1 + k] : 0u;
                    tk[k] += cnt[k][w];
                }
                tv += s_cnt[w][0];
            }
            const uint32_t tot = tk[0] + tk[1] + tk[2];
            uint32_t qb = 0;
            if (lane == 0) {
                qb = tot ? atomicAdd(&ctl->n_rays[itn], tot) : 0u;
                DevStats* sh = stat_shard(stats);
                if (tk[0]) atomicAdd(&sh->r1, (unsigned long long)tk[0]);
                if (tk[1]) atomicAdd(&sh->r2, (unsigned long long)tk[1]);
                if (tk[2]) atomicAdd(&sh->r3, (unsigned long long)tk[2]);
                if (tv) atomicAdd(&sh->vertices, (unsigned long long)tv);
            }
            qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)qb);
            uint32_t o = qb;
            for (uint32_t k = 0; k < 3; k++)
                for (uint32_t w = 0; w < kSW; w++) {
                    if (lane < cnt[k][w]) queue_out[o + lane] = s_stage[w][k][lane];
                    o += cnt[k][w];
                }
            RT_PROF(9)
            RT_PROF_FLUSH
            return;
        }
        __syncthreads();
        if (leader) {
            uint32_t te = 0, tsd = 0, tp = 0, tv = 0;
            for (uint32_t w = 0; w < kSW; w++) {
                te += s_cnt[w][1];
                tsd += s_cnt[w][2];
                tp += s_cnt[w][3];
                tv += s_cnt[w][0];
            }
            const uint32_t tot = te + tsd + tp;
            s_base[1] = tot ? atomicAdd(&ctl->n_rays[itn], tot) : 0u;
            DevStats* sh = stat_shard(stats);
            if (te) atomicAdd(&sh->r1, (unsigned long long)te);
            if (tsd) atomicAdd(&sh->r2, (unsigned long long)tsd);
            if (tp) atomicAdd(&sh->r3, (unsigned long long)tp);
            if (tv) atomicAdd(&sh->vertices, (unsigned long long)tv);
        }
        __syncthreads();
        const unsigned long long below = (1ull << lane) - 1ull;
#if RT_QUEUE_BY_KIND
        // the block's rays kind by kind -- [extension of the 4 waves][shadow ...][probe ...] -- so that a traversal
        // wave's 128-entry reservation is mostly one kind (shadow rays all run towards the light, probes and
        // extensions anywhere): k_trace -0.6 % (C4) / -2.9 % (C3) / -2.6 % (C2) against wave-by-wave order
        uint32_t oe = s_base[1], osd = 0, op = 0, te = 0, tsd = 0;
        for (uint32_t w = 0; w < kSW; w++) {
            if (w < wave) {
                oe += s_cnt[w][1];
                osd += s_cnt[w][2];
                op += s_cnt[w][3];
            }
            te += s_cnt[w][1];
            tsd += s_cnt[w][2];
        }
        osd += s_base[1] + te;
        op += s_base[1] + te + tsd;
        if (r.emit_ext) queue_out[oe + (uint32_t)__popcll(me & below)] = os | (kRayExt << 30);
        if (r.emit_sh) queue_out[osd + (uint32_t)__popcll(ms & below)] = os | (kRayShadow << 30);
        if (r.emit_pr) queue_out[op + (uint32_t)__popcll(mp & below)] = os | (kRayProbe << 30);
        (void)cp;
#else
        uint32_t off = s_base[1];
        for (uint32_t w = 0; w < wave; w++) off += s_cnt[w][1] + s_cnt[w][2] + s_cnt[w][3];
        if (r.emit_ext) queue_out[off + (uint32_t)__popcll(me & below)] = os | (kRayExt << 30);
        if (r.emit_sh) queue_out[off + ce + (uint32_t)__popcll(ms & below)] = os | (kRayShadow << 30);
        if (r.emit_pr) queue_out[off + ce + cs + (uint32_t)__popcll(mp & below)] = os | (kRayProbe << 30);
#endif
    }
    RT_PROF(9)
    RT_PROF_FLUSH
}

// ---------------------------------------------------------------------- tail
// Once a lane's batch is exhausted and few paths are left, per-bounce launches are bound by the single
// longest ray of each launch (a few hundred dependent fetches), times the remaining bounces.  k_tail
// finishes those paths in ONE launch with the same closest_hit and the same shade_a / shade_b, ping-ponging a
// path's slot between the two state buffers until it retires.  Same arithmetic, same counters, no queues.
// Persistent waves with (a) path replacement: a lane whose path has retired takes the next unfinished path of the
// list (one atomic per wave and refill), so a wave does not sit on a register allocation for the sake of its one
// longest path; and (b) pooled rays: the pending rays of the wave's live paths (up to three each: shadow, probe,
// extension) are listed in LDS and dealt to ALL 64 lanes, so a path's three rays are traced side by side.
// The two state buffers alternate per bounce, wave-uniformly: new paths are taken on even passes only.
// COUNT: the instrumented build (RT_RENDER_COUNT_TRAVERSAL) also counts the tail's node / primitive tests; its rays
// are always counted (rt_stats.tail_*), so that the traversal kernel's own share is known exactly.
template <int FEAT, bool COUNT>
__global__ __launch_bounds__(256) void k_tail(DevScene sc, PathState buf0, PathState buf1, Ctl* ctl,
                                             uint32_t it_abs, uint32_t max_depth, f64_t* lfx, f64_t* lfy,
                                             f64_t* lfz, DevStats* stats) {
    __shared__ int2 lds_stack[kLdsStack * 256];
    __shared__ uint32_t s_job[4][192];  // per wave: slot of the path | ray kind << 30
    __shared__ int2 s_res[4][192];      // per wave: {prim, leaf slot} found for job j
    const uint32_t ring = it_abs % kRing;
    const uint32_t n_active = ctl->n_active[ring];
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
                if (!alive && idx < n_active) {
                    slot = idx;
                    alive = true;
                }
                if (base + want >= n_active) list_done = true;
            }
        }
        // the state a path's owner lane wrote in the previous pass is read by other lanes of the wave below
        if (pass) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        uint32_t fl = kDead;
        if (alive) fl = in.flags[slot];
        if (fl & kDead) alive = false;
        if (__ballot(alive) == 0ull) {
            if (list_done) break;
            continue;  // (an odd pass with nothing alive: the next one refills)
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
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t n_jobs = nsh + npr + nex;
        n_traced += n_jobs;
        for (uint32_t j = lane; j < n_jobs; j += 64u) {
            const uint32_t job = s_job[wave][j];
            const uint32_t js = job & kSlotMask, kind = job >> 30;
            const D3 o = ld3(in.ox, in.oy, in.oz, js);
            double t;
            uint32_t hs = 0;
            int32_t prim;
            if (kind == kRayShadow) {  // Visibility::unoccluded, hittable.rs:25-32
                const D3 d = ld3(in.spx, in.spy, in.spz, js) - o;
                prim = closest_hit<COUNT>(sc, o + d * kSmall, d, 0.0, kInf, t, ts, &tc);
            } else if (kind == kRayProbe) {
                prim = closest_hit<COUNT>(sc, o, ld3(in.pdx, in.pdy, in.pdz, js), kSmall, kInf, t, ts, &tc);
            } else {
                prim = closest_hit<COUNT>(sc, o, ld3(in.dx, in.dy, in.dz, js), kSmall, kInf, t, ts, &tc, &hs);
            }
            s_res[wave][j] = make_int2(prim, (int)hs);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (has_sh) in.sh_prim[slot] = s_res[wave][jsh].x;
        if (has_pr) in.pr_prim[slot] = s_res[wave][jpr].x;
        if (has_ex) {
            const int2 r = s_res[wave][jex];
            in.hit_prim[slot] = r.x;
            in.hit_slot[slot] = (uint32_t)r.y;
        }
        if (alive) {
            ShadeA a;
            RT_PROF_DECL
            shade_a<FEAT>(sc, in, slot, true, max_depth, a RT_PROF_PASS);
            ShadeOut r{false, false, false, false};
            if (a.will_shade) {
                r = shade_b<FEAT>(sc, in, out, slot, slot, a RT_PROF_PASS);
                n_v++;
            }
            n_r1 += r.emit_ext ? 1u : 0u;
            n_r2 += r.emit_sh ? 1u : 0u;
            n_r3 += r.emit_pr ? 1u : 0u;
            if (!r.keep) {
                const uint32_t og = a.orig;
                lfx[og] = a.L.x;
                lfy[og] = a.L.y;
                lfz[og] = a.L.z;
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

#ifndef RT_F32  // the film is f64 in both modes: compiled once
// ------------------------------------------------------------------- resolve
// util::increment_color order: each pixel's samples are added one by one, in sample order.
__global__ __launch_bounds__(256) void k_resolve(const double* __restrict__ lfx, const double* __restrict__ lfy,
                                                 const double* __restrict__ lfz, ChunkDesc ck,
                                                 const uint32_t* __restrict__ pix_list, double* rgb_sum, uint32_t* n) {
    const uint32_t p_local = blockIdx.x * blockDim.x + threadIdx.x;
    if (p_local >= ck.n_pixels) return;
    const uint32_t pix = pix_list[ck.pixel_base + p_local];
    double r = rgb_sum[(size_t)pix * 3 + 0], g = rgb_sum[(size_t)pix * 3 + 1], b = rgb_sum[(size_t)pix * 3 + 2];
    for (uint32_t s = 0; s < ck.n_samples; s++) {
        const uint32_t slot = s * ck.n_pixels + p_local;
        r += lfx[slot];
        g += lfy[slot];
        b += lfz[slot];
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

// bvh_build.cpp -- host BVH builder for the device layout of scene_dev.h.
//
// Replaces BvhNode::new (src/hittable.rs:637-752: one primitive per leaf, median
// split on a random axis, O(N log^2 N)).  The reference's traversal is exhaustive,
// so its topology never changes a result; we are free to build a
// binned-SAH binary tree with up to 4 primitives per leaf and collapse it into a
// 4-wide BVH (each node absorbs its larger-area grandchildren until it has four
// children): child boxes stored in the parent, nodes in depth-first order (children
// of hot top levels stay adjacent in L2/Infinity Cache), depth bounded for the
// traversal stack.
#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <chrono>
#include <cstdio>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace rtd {

namespace {

struct Box {
    double mn[3], mx[3];
    void reset() {
        for (int a = 0; a < 3; a++) {
            mn[a] = std::numeric_limits<double>::infinity();
            mx[a] = -std::numeric_limits<double>::infinity();
        }
    }
    void grow(const double* lo, const double* hi) {
        for (int a = 0; a < 3; a++) {
            mn[a] = std::min(mn[a], lo[a]);
            mx[a] = std::max(mx[a], hi[a]);
        }
    }
    void grow(const Box& b) { grow(b.mn, b.mx); }
    double half_area() const {
        double dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (!(dx >= 0.0) || !(dy >= 0.0) || !(dz >= 0.0)) return 0.0;
        // clamp so that one astronomically large primitive (the 2e4-wide floor) cannot overflow
        return dx * dy + dy * dz + dz * dx;
    }
};

inline float f_down(double x) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafter(f, -std::numeric_limits<float>::infinity());
    return f;
}
inline float f_up(double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafter(f, std::numeric_limits<float>::infinity());
    return f;
}

struct BinNode {
    Box box;
    int32_t left = -1, right = -1;  // children (internal)
    int32_t leaf_ref = 0;           // leaf: -1 - ((first*8 + count-1) | kLeafCodeOther?)
    bool is_leaf = false;
};

struct Deferred {
    size_t b, e;
    uint32_t depth;
    int32_t node;  // placeholder in the parent builder's `bin`
};

// T - 1 helper threads that run one job at a time, all of them and the caller (index 0): the data-parallel loops of the
// builder's large top nodes.  Created once per build: spawning threads per loop cost more than the loops.
struct ForkJoin {
    unsigned T;
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv, done_cv;
    const std::function<void(unsigned)>* job = nullptr;
    unsigned long long gen = 0;
    unsigned pending = 0;
    bool stop = false;
    explicit ForkJoin(unsigned n) : T(n) {
        for (unsigned k = 1; k < T; k++)
            th.emplace_back([this, k] {
                unsigned long long seen = 0;
                for (;;) {
                    const std::function<void(unsigned)>* f;
                    {
                        std::unique_lock<std::mutex> lk(m);
                        cv.wait(lk, [&] { return stop || gen != seen; });
                        if (stop) return;
                        seen = gen;
                        f = job;
                    }
                    (*f)(k);
                    std::lock_guard<std::mutex> lk(m);
                    if (--pending == 0) done_cv.notify_one();
                }
            });
    }
    void run(const std::function<void(unsigned)>& f) {
        {
            std::lock_guard<std::mutex> lk(m);
            job = &f;
            gen++;
            pending = T - 1;
        }
        cv.notify_all();
        f(0u);
        std::unique_lock<std::mutex> lk(m);
        done_cv.wait(lk, [&] { return pending == 0; });
    }
    ~ForkJoin() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
};

struct Builder {
    const rt_primitive* prims;
    std::vector<uint32_t> order_store;  // primitive ids, partitioned in place (owned by the top builder)
    std::vector<double> cx_store, cy_store, cz_store;
    uint32_t* order = nullptr;
    const double *cx = nullptr, *cy = nullptr, *cz = nullptr;
    std::vector<BinNode> bin;
    // parallel build: subtrees of at least `defer_min` primitives at depth >= `defer_depth` are left as
    // placeholders for worker threads (each builds into its own `bin`; the ranges of `order` are disjoint)
    uint32_t defer_depth = 0xffffffffu;
    size_t defer_min = 0, defer_max = ~(size_t)0;
    std::vector<Deferred> deferred;
    // The nodes above the deferred subtrees are few and large (the root: every primitive): their per-primitive loops --
    // bounds, centroid bounds, the three binning passes, the largest-primitive scan -- run on `par_threads` threads,
    // each over a contiguous chunk with its own accumulators, merged in chunk order (min / max / + only: the same
    // values as the serial loop, so the same tree).  1 in the worker builders.
    unsigned par_threads = 1;
    ForkJoin* pool = nullptr;  // the top builder's helper threads (created once per build, build_binary)
    static constexpr size_t kParMin = 65536;
    template <typename F>
    void par_chunks(size_t b, size_t e, F&& fn) const {  // fn(chunk index, begin, end)
        const size_t n = e - b;
        const unsigned T = (pool && par_threads > 1 && n >= kParMin) ? par_threads : 1u;
        if (T == 1) {
            fn(0u, b, e);
            return;
        }
        pool->run([&](unsigned k) { fn(k, b + n * k / T, b + n * (k + 1) / T); });
    }
    unsigned n_chunks(size_t b, size_t e) const { return (pool && par_threads > 1 && e - b >= kParMin) ? par_threads : 1u; }
    uint32_t sah_depth = 32;  // experiment knobs: RT_BVH_SAH_DEPTH, RT_BVH_BINS
    int n_bins = 16;
    size_t max_leaf = (size_t)kLeafTargetPrims;  // RT_BVH_LEAF

    Builder(const rt_primitive* p, size_t n) : prims(p) {
        if (const char* e = getenv("RT_BVH_SAH_DEPTH")) sah_depth = (uint32_t)std::max(1, atoi(e));
        if (const char* e = getenv("RT_BVH_BINS")) n_bins = std::min(64, std::max(2, atoi(e)));
        if (const char* e = getenv("RT_BVH_LEAF")) max_leaf = (size_t)std::min(kMaxLeafPrims, std::max(1, atoi(e)));
        order_store.resize(n);
        cx_store.resize(n);
        cy_store.resize(n);
        cz_store.resize(n);
        for (size_t i = 0; i < n; i++) {
            order_store[i] = (uint32_t)i;
            cx_store[i] = 0.5 * (p[i].bbox_min[0] + p[i].bbox_max[0]);
            cy_store[i] = 0.5 * (p[i].bbox_min[1] + p[i].bbox_max[1]);
            cz_store[i] = 0.5 * (p[i].bbox_min[2] + p[i].bbox_max[2]);
        }
        order = order_store.data();
        cx = cx_store.data();
        cy = cy_store.data();
        cz = cz_store.data();
    }
    // a worker's view of the same arrays
    Builder(const Builder& top, int) : prims(top.prims), order(top.order), cx(top.cx), cy(top.cy), cz(top.cz),
                                       sah_depth(top.sah_depth), n_bins(top.n_bins), max_leaf(top.max_leaf) {}
    double centroid(uint32_t id, int axis) const { return axis == 0 ? cx[id] : (axis == 1 ? cy[id] : cz[id]); }

    Box bounds(size_t b, size_t e) const {
        std::vector<Box> part(n_chunks(b, e));
        par_chunks(b, e, [&](unsigned k, size_t cb, size_t ce) {
            Box bx;
            bx.reset();
            for (size_t i = cb; i < ce; i++) bx.grow(prims[order[i]].bbox_min, prims[order[i]].bbox_max);
            part[k] = bx;
        });
        Box bx = part[0];
        for (size_t k = 1; k < part.size(); k++) bx.grow(part[k]);
        return bx;
    }

    // binary tree over order[b, e); returns the BinNode index
    int32_t build(size_t b, size_t e, uint32_t depth) {
        const int32_t me = (int32_t)bin.size();
        bin.push_back(BinNode{});
        bin[me].box = bounds(b, e);
        const size_t n = e - b;
        // Spheres and rects are large, few and expensive to test: each gets a leaf of its own, so that the
        // (4-at-once) box tests and the front-to-back order prune them instead of a leaf loop testing all.
        bool any_other = false;
        for (size_t i = b; i < e && n <= max_leaf; i++)
            any_other = any_other || prims[order[i]].kind != RT_PRIM_TRIANGLE;
        if (n <= max_leaf && (n == 1 || !any_other)) {
            bin[me].is_leaf = true;
            bin[me].leaf_ref = -1 - (int32_t)((uint32_t)(b * 8 + (n - 1)) | (any_other ? kLeafCodeOther : 0u));
            return me;
        }
        if (depth >= defer_depth && n >= defer_min && n <= defer_max) {  // a worker thread builds this subtree
            deferred.push_back(Deferred{b, e, depth, me});
            return me;
        }
        // centroid bounds
        double cmn[3], cmx[3];
        for (int a = 0; a < 3; a++) {
            cmn[a] = std::numeric_limits<double>::infinity();
            cmx[a] = -cmn[a];
        }
        {
            struct CB {
                double mn[3], mx[3];
            };
            std::vector<CB> part(n_chunks(b, e));
            par_chunks(b, e, [&](unsigned k, size_t cb, size_t ce) {
                CB c;
                for (int a = 0; a < 3; a++) {
                    c.mn[a] = std::numeric_limits<double>::infinity();
                    c.mx[a] = -c.mn[a];
                }
                for (size_t i = cb; i < ce; i++)
                    for (int a = 0; a < 3; a++) {
                        const double v = centroid(order[i], a);
                        c.mn[a] = std::min(c.mn[a], v);
                        c.mx[a] = std::max(c.mx[a], v);
                    }
                part[k] = c;
            });
            for (const CB& c : part)
                for (int a = 0; a < 3; a++) {
                    cmn[a] = std::min(cmn[a], c.mn[a]);
                    cmx[a] = std::max(cmx[a], c.mx[a]);
                }
        }
        size_t mid = b;
        bool split_done = false;
        if (depth < sah_depth) {  // SAH above, object median below: total binary depth <= sah_depth + ceil(log2(N/4))
            constexpr int NBMAX = 64;
            const int NB = n_bins;
            double best_cost = std::numeric_limits<double>::infinity();
            int best_axis = -1, best_bin = -1;
            for (int a = 0; a < 3; a++) {
                double ext = cmx[a] - cmn[a];
                if (!(ext > 0.0)) continue;
                Box bb[NBMAX];
                size_t cnt[NBMAX];
                for (int k = 0; k < NB; k++) {
                    bb[k].reset();
                    cnt[k] = 0;
                }
                double scale = (double)NB / ext;
                {
                    struct Bins {
                        Box bb[NBMAX];
                        size_t cnt[NBMAX];
                    };
                    std::vector<Bins> part(n_chunks(b, e));
                    par_chunks(b, e, [&](unsigned kc, size_t cb, size_t ce) {
                        Bins& bn = part[kc];
                        for (int k = 0; k < NB; k++) {
                            bn.bb[k].reset();
                            bn.cnt[k] = 0;
                        }
                        for (size_t i = cb; i < ce; i++) {
                            int k = (int)((centroid(order[i], a) - cmn[a]) * scale);
                            k = std::min(std::max(k, 0), NB - 1);
                            bn.cnt[k]++;
                            bn.bb[k].grow(prims[order[i]].bbox_min, prims[order[i]].bbox_max);
                        }
                    });
                    for (const Bins& bn : part)
                        for (int k = 0; k < NB; k++) {
                            cnt[k] += bn.cnt[k];
                            if (bn.cnt[k]) bb[k].grow(bn.bb[k]);
                        }
                }
                double right_area[NBMAX];
                size_t right_cnt[NBMAX];
                Box acc;
                acc.reset();
                size_t c = 0;
                for (int k = NB - 1; k > 0; k--) {
                    acc.grow(bb[k]);
                    c += cnt[k];
                    right_area[k] = acc.half_area();
                    right_cnt[k] = c;
                }
                acc.reset();
                c = 0;
                for (int k = 0; k < NB - 1; k++) {
                    acc.grow(bb[k]);
                    c += cnt[k];
                    if (c == 0 || right_cnt[k + 1] == 0) continue;
                    double cost = acc.half_area() * (double)c + right_area[k + 1] * (double)right_cnt[k + 1];
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = a;
                        best_bin = k;
                    }
                }
            }
            // One more candidate: the primitive of largest surface area alone against all the others.  Binning by
            // centroids cannot see it -- a 2e4-wide floor rect has its centroid among the mesh's -- and a box that
            // large left inside a subtree is fetched by every ray on the way to what it is mixed with (two_dragons,
            // 16 bins: three nested nodes with the floor's box, 9.1 node fetches per ray against 7.3 once it is
            // split off at the root).  Only evaluated when one primitive makes up a third of the node's area.
            if (n > 2) {
                size_t big = b;
                double big_area = -1.0;
                {
                    struct Big {
                        size_t i;
                        double a;
                    };
                    std::vector<Big> part(n_chunks(b, e));
                    par_chunks(b, e, [&](unsigned k, size_t cb, size_t ce) {
                        Big g{cb, -1.0};
                        for (size_t i = cb; i < ce; i++) {
                            Box pb;
                            pb.reset();
                            pb.grow(prims[order[i]].bbox_min, prims[order[i]].bbox_max);
                            const double a = pb.half_area();
                            if (a > g.a) {
                                g.a = a;
                                g.i = i;
                            }
                        }
                        part[k] = g;
                    });
                    for (const Big& g : part)  // chunk order: the first of equal areas wins, as in the serial scan
                        if (g.a > big_area) {
                            big_area = g.a;
                            big = g.i;
                        }
                }
                if (big_area * 3.0 >= bin[me].box.half_area()) {
                    Box rest;
                    rest.reset();
                    {
                        std::vector<Box> part(n_chunks(b, e));
                        par_chunks(b, e, [&](unsigned k, size_t cb, size_t ce) {
                            Box r;
                            r.reset();
                            for (size_t i = cb; i < ce; i++)
                                if (i != big) r.grow(prims[order[i]].bbox_min, prims[order[i]].bbox_max);
                            part[k] = r;
                        });
                        for (const Box& r : part) rest.grow(r);
                    }
                    const double cost = big_area + rest.half_area() * (double)(n - 1);
                    if (cost < best_cost) {
                        std::swap(order[b], order[big]);
                        mid = b + 1;
                        split_done = true;
                        best_axis = -1;
                    }
                }
            }
            if (best_axis >= 0) {
                double ext = cmx[best_axis] - cmn[best_axis];
                double scale = (double)NB / ext;
                auto it = std::partition(order + b, order + e, [&](uint32_t id) {
                    int k = (int)((centroid(id, best_axis) - cmn[best_axis]) * scale);
                    k = std::min(std::max(k, 0), NB - 1);
                    return k <= best_bin;
                });
                mid = (size_t)(it - order);
                split_done = mid > b && mid < e;
            }
        }
        if (!split_done) {  // object median on the widest centroid axis (also the depth-bound fallback)
            int axis = 0;
            double ext = cmx[0] - cmn[0];
            for (int a = 1; a < 3; a++)
                if (cmx[a] - cmn[a] > ext) {
                    ext = cmx[a] - cmn[a];
                    axis = a;
                }
            mid = b + n / 2;
            std::nth_element(order + b, order + mid, order + e, [&](uint32_t x, uint32_t y) {
                double a_ = centroid(x, axis), b_ = centroid(y, axis);
                return a_ < b_ || (a_ == b_ && x < y);
            });
        }
        const int32_t l = build(b, mid, depth + 1);
        const int32_t r = build(mid, e, depth + 1);
        bin[me].left = l;
        bin[me].right = r;
        return me;
    }
};

// Collapse: a 4-wide node takes a binary node's two children and keeps replacing the internal child
// of largest surface area by its own two children until it has four (or only leaves are left).
struct Collapser {
    const std::vector<BinNode>& bin;
    std::vector<DevNode>& nodes;
    uint32_t max_depth = 0;
    Collapser(const std::vector<BinNode>& b, std::vector<DevNode>& n) : bin(b), nodes(n) {}

    int32_t emit(int32_t bi, uint32_t depth) {  // bi is an internal binary node
        max_depth = std::max(max_depth, depth);
        int32_t kids[4];
        int nk = 0;
        kids[nk++] = bin[bi].left;
        kids[nk++] = bin[bi].right;
        while (nk < 4) {
            int best = -1;
            double best_area = -1.0;
            for (int k = 0; k < nk; k++)
                if (!bin[kids[k]].is_leaf) {
                    double a = bin[kids[k]].box.half_area();
                    if (a > best_area) {
                        best_area = a;
                        best = k;
                    }
                }
            if (best < 0) break;
            const int32_t c = kids[best];
            kids[best] = bin[c].left;
            kids[nk++] = bin[c].right;
        }
        const int32_t me = (int32_t)nodes.size();
        nodes.push_back(DevNode{});
        int32_t refs[4];
        for (int k = 0; k < 4; k++) {
            if (k >= nk)
                refs[k] = kNoChild;
            else if (bin[kids[k]].is_leaf)
                refs[k] = bin[kids[k]].leaf_ref;
            else
                refs[k] = emit(kids[k], depth + 1);
        }
        DevNode& nd = nodes[me];
        for (int k = 0; k < 4; k++) {
            nd.child[k] = refs[k];
            nd.pad[k] = 0;
            if (k < nk) {
                const Box& bx = bin[kids[k]].box;
                nd.lo_x[k] = f_down(bx.mn[0]); nd.lo_y[k] = f_down(bx.mn[1]); nd.lo_z[k] = f_down(bx.mn[2]);
                nd.hi_x[k] = f_up(bx.mx[0]);   nd.hi_y[k] = f_up(bx.mx[1]);   nd.hi_z[k] = f_up(bx.mx[2]);
            } else {
                nd.lo_x[k] = nd.lo_y[k] = nd.lo_z[k] = 0.0f;
                nd.hi_x[k] = nd.hi_y[k] = nd.hi_z[k] = 0.0f;
            }
        }
        return me;
    }
};

}  // namespace

// The binary tree over all primitives: serial top (depth < 5), worker threads below.  Returns the root's BinNode index.
static int32_t build_binary(Builder& bd, size_t n) {
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (hw > 1 && n >= 65536) {
        // subtrees of at most 1/64 of the scene go to the worker threads (a hundred-odd tasks, dealt dynamically: the
        // depth-5 rule of rounds 1-2 produced 8 tasks of very different size for two_dragons); the large nodes above them
        // are built here with their per-primitive loops spread over the threads
        bd.defer_depth = 0;
        bd.defer_min = 2048;
        bd.defer_max = std::max<size_t>(4096, n / 64);
        bd.par_threads = hw;
    }
    const auto t0 = std::chrono::steady_clock::now();
    int32_t root;
    {
        ForkJoin pool(bd.par_threads);
        if (bd.par_threads > 1) bd.pool = &pool;
        root = bd.build(0, n, 0);
        bd.pool = nullptr;
    }
    if (getenv("RT_DIAG"))
        fprintf(stderr, "[rt diag] host BVH: large top nodes %.1f ms (%zu subtrees left to the workers)\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), bd.deferred.size());
    const auto t1 = std::chrono::steady_clock::now();
    if (!bd.deferred.empty()) {
        // workers: one private Builder per deferred subtree, same splits as the serial build would make
        std::vector<Builder> subs;
        subs.reserve(bd.deferred.size());
        for (size_t i = 0; i < bd.deferred.size(); i++) subs.emplace_back(bd, 0);
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t i; (i = next.fetch_add(1)) < bd.deferred.size();) {
                const Deferred& d = bd.deferred[i];
                subs[i].bin.reserve(2 * (d.e - d.b));
                subs[i].build(d.b, d.e, d.depth);
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < hw; t++) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
        // splice: a sub-builder's node 0 replaces the placeholder, the rest is appended
        for (size_t i = 0; i < bd.deferred.size(); i++) {
            const int32_t hole = bd.deferred[i].node;
            const int32_t base = (int32_t)bd.bin.size() - 1;
            auto remap = [&](int32_t c) { return c < 0 ? c : (c == 0 ? hole : base + c); };
            const std::vector<BinNode>& sb = subs[i].bin;
            for (size_t k = 0; k < sb.size(); k++) {
                BinNode nd = sb[k];
                if (!nd.is_leaf) {
                    nd.left = remap(nd.left);
                    nd.right = remap(nd.right);
                }
                if (k == 0)
                    bd.bin[hole] = nd;
                else
                    bd.bin.push_back(nd);
            }
        }
    }
    if (getenv("RT_DIAG"))
        fprintf(stderr, "[rt diag] host BVH: worker subtrees + splice %.1f ms\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    return root;
}

static void build_once(const rt_primitive* prims, size_t n, uint32_t sah_depth_cap, BvhOut& out) {
    out.nodes.clear();
    out.order.clear();
    out.depth = 0;
    Builder bd(prims, n);
    bd.sah_depth = std::min(bd.sah_depth, sah_depth_cap);
    const int32_t root = build_binary(bd, n);
    out.nodes.reserve(bd.bin.size() / 2 + 2);
    if (bd.bin[root].is_leaf) {
        // the whole scene fits one leaf: wrap it so that node 0 is always internal
        DevNode nd{};
        const Box& bx = bd.bin[root].box;
        nd.lo_x[0] = f_down(bx.mn[0]); nd.lo_y[0] = f_down(bx.mn[1]); nd.lo_z[0] = f_down(bx.mn[2]);
        nd.hi_x[0] = f_up(bx.mx[0]);   nd.hi_y[0] = f_up(bx.mx[1]);   nd.hi_z[0] = f_up(bx.mx[2]);
        nd.child[0] = bd.bin[root].leaf_ref;
        nd.child[1] = nd.child[2] = nd.child[3] = kNoChild;
        out.nodes.push_back(nd);
    } else {
        Collapser col(bd.bin, out.nodes);
        col.emit(root, 0);
        out.depth = col.max_depth;
    }
    out.order = std::move(bd.order_store);
}

void build_sah_binary(const double* boxes, size_t n, std::vector<int32_t>& left, std::vector<int32_t>& right) {
    left.assign(n > 1 ? n - 1 : 0, 0);
    right.assign(n > 1 ? n - 1 : 0, 0);
    if (n < 2) return;
    // the builder reads only a primitive's box and kind
    std::vector<rt_primitive> fake(n);
    std::memset(fake.data(), 0, n * sizeof(rt_primitive));
    for (size_t i = 0; i < n; i++) {
        fake[i].kind = RT_PRIM_TRIANGLE;
        for (int a = 0; a < 3; a++) {
            fake[i].bbox_min[a] = boxes[i * 6 + a];
            fake[i].bbox_max[a] = boxes[i * 6 + 3 + a];
        }
    }
    Builder bd(fake.data(), n);
    bd.max_leaf = 1;
    const int32_t root = bd.build(0, n, 0);  // serial: a few thousand boxes
    // number the internal nodes depth-first from the root
    std::vector<int32_t> id(bd.bin.size(), -1);
    int32_t next = 0;
    std::vector<int32_t> stack{root};
    while (!stack.empty()) {
        const int32_t b = stack.back();
        stack.pop_back();
        if (bd.bin[b].is_leaf) continue;
        id[b] = next++;
        stack.push_back(bd.bin[b].right);
        stack.push_back(bd.bin[b].left);
    }
    auto ref = [&](int32_t b) -> int32_t {
        if (!bd.bin[b].is_leaf) return id[b];
        const uint32_t code = (uint32_t)(-1 - bd.bin[b].leaf_ref);
        return ~(int32_t)bd.order[(code & ~kLeafCodeOther) >> 3];
    };
    for (size_t b = 0; b < bd.bin.size(); b++)
        if (id[b] >= 0) {
            left[id[b]] = ref(bd.bin[b].left);
            right[id[b]] = ref(bd.bin[b].right);
        }
}

void build_bvh(const rt_primitive* prims, size_t n, BvhOut& out) {
    out.nodes.clear();
    out.order.clear();
    out.depth = 0;
    if (n == 0) return;
    // SAH splits as deep as the traversal stack allows: an adversarial distribution (SAH peeling one primitive
    // per level) is rebuilt with SAH confined to fewer top levels and object-median splits below, which bound
    // the binary depth by cap + ceil(log2 n).
    for (uint32_t cap : {64u, 16u, 8u, 0u}) {
        build_once(prims, n, cap, out);
        if (out.depth + 1 <= (uint32_t)kMaxBvhDepth) return;
    }
}

}  // namespace rtd

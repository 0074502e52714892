// bvh_gpu.hip -- BVH construction on the device.
//
// Replaces BvhNode::new (src/hittable.rs:637-752: recursive median split on a random axis with a sort at
// every level and one entropy RNG per node) for callers that re-commit geometry often.  The reference's
// traversal visits both children of every node it enters (hittable.rs:604-605), so its answer does not
// depend on the tree; the device traversal keeps that property because a primitive is gated
// only by the f64 slab test of its OWN box (geom.h: leaf_step).  Any topology therefore renders the same
// film bit for bit, and the builder is free to be a linear BVH:
//
//   kb_bounds   centroid bounds of the scene (block reduction + ordered-integer atomics)
//   kb_morton   63-bit Morton key of every centroid (21 bits per axis)
//   kb_hist / kb_scan / kb_scatter   stable LSD radix sort, 8 passes x 8 bits, 2048-key tiles; ranks inside a
//               wave come from eight __ballot()s per key (64-wide match), so a tile needs no LDS atomics
//   kb_tree     binary radix tree over the sorted keys (Karras 2012; duplicates split by index)
//   kb_refit    leaf-to-root f64 boxes, "all triangles" flags; second arrival at a node does the union
//   kb_emit     one launch per level of the 4-wide tree: a node adopts the grandchildren of larger surface
//               area until it has four children, subtrees of <= 4 triangles become one leaf, spheres and
//               rects stay alone in theirs; child boxes are rounded OUTWARD to f32 (scene_dev.h: DevNode)
//   kb_leaves   leaf slots: vertices (or sphere / rect parameters) gathered in sorted order
//
// Everything runs on one stream; the host only reads back the per-level queue length.
#include "bvh_gpu.h"
#include "bvh_build.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

namespace rtd {

namespace {

typedef unsigned long long ull;

constexpr int kSortThreads = 256;
constexpr int kSortItems = 8;
constexpr int kSortTile = kSortThreads * kSortItems;

struct BuildGlobals {
    ull cmin[3], cmax[3];  // centroid bounds, order-preserving integer encoding of f64
    uint32_t node_count;   // 4-wide nodes allocated so far
    uint32_t q_count[2];   // work items of the current / next level
    uint32_t unsorted;     // sort self-check: adjacent pairs out of order
    uint32_t n_tri;
    uint32_t overflow;     // node pool exhausted (cannot happen: one 4-wide node per binary node at most)
};

__device__ inline ull enc_f64(double d) {
    const ull b = (ull)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ inline double dec_f64(ull e) {
    const ull b = (e >> 63) ? (e & 0x7fffffffffffffffull) : ~e;
    return __longlong_as_double((long long)b);
}

__device__ inline float f_down(double x) {  // largest float <= x
    float f = (float)x;
    if ((double)f > x) {
        const uint32_t u = __float_as_uint(f);
        f = f > 0.0f ? __uint_as_float(u - 1u) : (f < 0.0f ? __uint_as_float(u + 1u) : __uint_as_float(0x80000001u));
    }
    return f;
}
__device__ inline float f_up(double x) {  // smallest float >= x
    float f = (float)x;
    if ((double)f < x) {
        const uint32_t u = __float_as_uint(f);
        f = f > 0.0f ? __uint_as_float(u + 1u) : (f < 0.0f ? __uint_as_float(u - 1u) : __uint_as_float(0x00000001u));
    }
    return f;
}

__global__ void kb_init(BuildGlobals* g) {
    for (int a = 0; a < 3; a++) {
        g->cmin[a] = ~0ull;
        g->cmax[a] = 0ull;
    }
    g->node_count = 1;  // the root
    g->q_count[0] = 1;
    g->q_count[1] = 0;
    g->unsorted = 0;
    g->n_tri = 0;
    g->overflow = 0;
}

__device__ inline double wave_min(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
    return v;
}
__device__ inline double wave_max(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}

__global__ __launch_bounds__(256) void kb_bounds(const rt_primitive* __restrict__ prims, uint32_t n, BuildGlobals* g) {
    double mn[3] = {1e308, 1e308, 1e308}, mx[3] = {-1e308, -1e308, -1e308};
    uint32_t tris = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const rt_primitive& p = prims[i];
        for (int a = 0; a < 3; a++) {
            const double c = (p.bbox_min[a] + p.bbox_max[a]) * 0.5;
            mn[a] = fmin(mn[a], c);
            mx[a] = fmax(mx[a], c);
        }
        tris += p.kind == RT_PRIM_TRIANGLE;
    }
    __shared__ double s_mn[4][3], s_mx[4][3];
    __shared__ uint32_t s_tris[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int a = 0; a < 3; a++) {
        const double lo = wave_min(mn[a]), hi = wave_max(mx[a]);
        if (lane == 0) {
            s_mn[wave][a] = lo;
            s_mx[wave][a] = hi;
        }
    }
    for (int o = 32; o > 0; o >>= 1) tris += __shfl_down(tris, o, 64);
    if (lane == 0) s_tris[wave] = tris;
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const double lo = fmin(fmin(s_mn[0][a], s_mn[1][a]), fmin(s_mn[2][a], s_mn[3][a]));
        const double hi = fmax(fmax(s_mx[0][a], s_mx[1][a]), fmax(s_mx[2][a], s_mx[3][a]));
        atomicMin(&g->cmin[a], enc_f64(lo));
        atomicMax(&g->cmax[a], enc_f64(hi));
    }
    if (threadIdx.x == 3) atomicAdd(&g->n_tri, s_tris[0] + s_tris[1] + s_tris[2] + s_tris[3]);
}

__device__ inline ull spread21(ull x) {  // bit i -> bit 3i
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(256) void kb_morton(const rt_primitive* __restrict__ prims, uint32_t n,
                                                 const BuildGlobals* g, ull* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const rt_primitive& p = prims[i];
    ull q[3];
    double scene_ext = 0.0, own_ext = 0.0;
    for (int a = 0; a < 3; a++) {
        const double lo = dec_f64(g->cmin[a]), hi = dec_f64(g->cmax[a]);
        const double c = (p.bbox_min[a] + p.bbox_max[a]) * 0.5;
        const double ext = hi - lo;
        double f = ext > 0.0 ? (c - lo) / ext * 2097152.0 : 0.0;
        f = fmin(fmax(f, 0.0), 2097151.0);
        q[a] = (ull)f;
        scene_ext = fmax(scene_ext, ext);
        own_ext = fmax(own_ext, p.bbox_max[a] - p.bbox_min[a]);
    }
    // A primitive that is large against the spread of the centroids (a floor rect 2e4 wide under a mesh, the walls
    // of the Cornell box around the statue) would, sorted by its centroid, inflate every box between its leaf and
    // the root.  Bit 63 -- free in a 63-bit Morton key -- sends such primitives to the root's second subtree: the
    // radix tree splits on the highest differing bit first, so they never share an inner node with the small ones.
    const ull large = (own_ext >= 0.25 * scene_ext && scene_ext > 0.0) ? (1ull << 63) : 0ull;
    keys[i] = spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2) | large;
    vals[i] = i;
}

// ---- radix sort: histogram of one 8-bit digit per tile, laid out [digit][tile]
__global__ __launch_bounds__(kSortThreads) void kb_hist(const ull* __restrict__ keys, uint32_t n, int shift,
                                                        uint32_t* hist, uint32_t n_tiles) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortTile;
#pragma unroll
    for (int r = 0; r < kSortItems; r++) {
        const uint32_t i = base + r * kSortThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}

// exclusive prefix sum of m counters in place, one block
__global__ __launch_bounds__(1024) void kb_scan(uint32_t* a, uint32_t m) {
    __shared__ uint32_t part[1024];
    const uint32_t chunk = (m + 1023u) / 1024u;
    const uint32_t b = threadIdx.x * chunk, e = b + chunk < m ? b + chunk : m;
    uint32_t s = 0;
    for (uint32_t i = b; i < e; i++) s += a[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {
        const uint32_t v = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;  // exclusive
    for (uint32_t i = b; i < e; i++) {
        const uint32_t v = a[i];
        a[i] = run;
        run += v;
    }
}

// stable scatter of one tile: wave w owns keys [w*512, (w+1)*512) of the tile, 8 rounds of 64
__global__ __launch_bounds__(kSortThreads) void kb_scatter(const ull* __restrict__ keys_in,
                                                           const uint32_t* __restrict__ vals_in, ull* keys_out,
                                                           uint32_t* vals_out, uint32_t n, int shift,
                                                           const uint32_t* __restrict__ hist, uint32_t n_tiles) {
    __shared__ uint32_t cnt[4][256];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int w = 0; w < 4; w++) cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortTile + wave * (64u * kSortItems);
    const ull below = (1ull << lane) - 1ull;
    ull k[kSortItems];
    uint32_t v[kSortItems], lr[kSortItems];
#pragma unroll
    for (int r = 0; r < kSortItems; r++) {
        const uint32_t i = base + r * 64u + lane;
        const bool valid = i < n;
        k[r] = valid ? keys_in[i] : ~0ull;
        v[r] = valid ? vals_in[i] : 0u;
        const uint32_t digit = (uint32_t)(k[r] >> shift) & 255u;
        ull m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (digit >> b) & 1u;
            const ull bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__popcll(m & below);
        const uint32_t old = cnt[wave][digit];  // every lane reads before the group's first lane adds
        if (valid && rank == 0) cnt[wave][digit] = old + (uint32_t)__popcll(m);
        lr[r] = old + rank;
    }
    __syncthreads();
    {   // digit d: global base of this tile, then the waves of the tile in order
        const uint32_t d = threadIdx.x;
        uint32_t gbase = hist[d * n_tiles + blockIdx.x];
        for (int w = 0; w < 4; w++) {
            const uint32_t c = cnt[w][d];
            cnt[w][d] = gbase;
            gbase += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortItems; r++) {
        const uint32_t i = base + r * 64u + lane;
        if (i < n) {
            const uint32_t digit = (uint32_t)(k[r] >> shift) & 255u;
            const uint32_t dst = cnt[wave][digit] + lr[r];
            keys_out[dst] = k[r];
            vals_out[dst] = v[r];
        }
    }
}

__global__ __launch_bounds__(256) void kb_check_sorted(const ull* __restrict__ keys, uint32_t n, BuildGlobals* g) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 < n && keys[i] > keys[i + 1]) atomicAdd(&g->unsorted, 1u);
}

// ---- binary radix tree (Karras 2012).  Internal nodes 0..n-2, node 0 = root; a child c >= 0 is an
// internal node, c < 0 is leaf ~c (position in sorted order).
struct Tree {
    int32_t* left;       // [n-1]
    int32_t* right;      // [n-1]
    int32_t* parent;     // [n-1] parent of an internal node (-1 for the root)
    int32_t* leaf_par;   // [n]
    uint32_t* first;     // [n-1] range of sorted positions covered
    uint32_t* last;      // [n-1]
    double* box;         // [n-1][6] min xyz, max xyz
    uint32_t* flag;      // [n-1] arrivals (refit)
    uint32_t* all_tri;   // [n-1] subtree holds triangles only
};

__device__ inline int key_delta(const ull* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const ull a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((uint32_t)(i ^ j));
    return __clzll((long long)(a ^ b));
}

__global__ __launch_bounds__(256) void kb_tree(const ull* __restrict__ keys, int n, Tree t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = key_delta(keys, n, i, i + 1) - key_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = key_delta(keys, n, i, i - d);
    int lmax = 2;
    while (key_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int s = lmax / 2; s >= 1; s /= 2)
        if (key_delta(keys, n, i, i + (l + s) * d) > dmin) l += s;
    const int j = i + l * d;
    const int dnode = key_delta(keys, n, i, j);
    int s = 0;
    for (int step = l;;) {
        step = (step + 1) / 2;
        if (key_delta(keys, n, i, i + (s + step) * d) > dnode) s += step;
        if (step <= 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int32_t lc = lo == gamma ? ~gamma : gamma;
    const int32_t rc = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    t.left[i] = lc;
    t.right[i] = rc;
    t.first[i] = (uint32_t)lo;
    t.last[i] = (uint32_t)hi;
    if (lc >= 0) t.parent[lc] = i; else t.leaf_par[~lc] = i;
    if (rc >= 0) t.parent[rc] = i; else t.leaf_par[~rc] = i;
    if (i == 0) t.parent[0] = -1;
}

__device__ inline double ld_coherent(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const ull*>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}
__device__ inline uint32_t ld_coherent(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline void child_box(const Tree& t, const rt_primitive* __restrict__ prims,
                                 const uint32_t* __restrict__ order, int32_t c, double* b, bool& tri) {
    if (c >= 0) {
        for (int a = 0; a < 6; a++) b[a] = ld_coherent(&t.box[(size_t)c * 6 + a]);
        tri = ld_coherent(&t.all_tri[c]) != 0;
    } else {
        const rt_primitive& p = prims[order[~c]];
        for (int a = 0; a < 3; a++) {
            b[a] = p.bbox_min[a];
            b[3 + a] = p.bbox_max[a];
        }
        tri = p.kind == RT_PRIM_TRIANGLE;
    }
}

// Tree rotations (Kensler 2008), fused into the refit: the thread that completes node p owns the whole subtree below
// it, so it may exchange one child of p with a grandchild on the other side when that shrinks the box -- and with it
// the surface-area cost -- of the child node that receives it:
//     p = (L, R = (a, b))  ->  p = (a, R' = (L, b))   if area(L u b) < area(R),   and the three mirror cases.
// A Morton-order tree splits space at fixed bit planes, so a large primitive or a cluster that straddles a plane
// inflates a whole subtree; one rotation per node on the way up undoes the worst of it (node fetches per ray of the
// device-built tree against the host SAH tree: DESIGN.md section 7).  Only with one primitive per leaf: a rotation breaks
// the contiguity of a subtree's range of sorted positions, which multi-primitive leaf codes rely on.
__device__ inline double box_area(const double* b) {
    const double dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
}
__device__ inline void set_parent(const Tree& t, int32_t c, int32_t par) {
    if (c >= 0)
        t.parent[c] = par;
    else
        t.leaf_par[~c] = par;
}
__device__ inline double union_area(const double* a, const double* b) {
    const double dx = fmax(a[3], b[3]) - fmin(a[0], b[0]), dy = fmax(a[4], b[4]) - fmin(a[1], b[1]),
                 dz = fmax(a[5], b[5]) - fmin(a[2], b[2]);
    return dx * dy + dy * dz + dz * dx;
}

__global__ __launch_bounds__(256) void kb_refit(const rt_primitive* __restrict__ prims,
                                                const uint32_t* __restrict__ order, int n, Tree t, int rotate) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    int p = t.leaf_par[j];
    while (p >= 0) {
        __threadfence();
        if (atomicAdd(&t.flag[p], 1u) == 0u) return;  // the sibling subtree is not finished: its thread goes on
        __threadfence();
        double bl[6], br[6];
        bool tl, tr;
        child_box(t, prims, order, t.left[p], bl, tl);
        child_box(t, prims, order, t.right[p], br, tr);
        if (rotate && kLeafTargetPrims == 1) {
            const int32_t L = t.left[p], R = t.right[p];
            // candidates: (side whose node is rebuilt, which of its children stays)
            double best_gain = 0.0;
            int best = -1;
            double gb[4][6];  // boxes of the grandchildren: R.left, R.right, L.left, L.right
            bool gt[4] = {false, false, false, false};
            if (R >= 0) {
                child_box(t, prims, order, t.left[R], gb[0], gt[0]);
                child_box(t, prims, order, t.right[R], gb[1], gt[1]);
                const double ar = box_area(br);
                const double g0 = ar - union_area(bl, gb[1]);  // L <-> R.left : R' = (L, R.right)
                const double g1 = ar - union_area(bl, gb[0]);  // L <-> R.right: R' = (R.left, L)
                if (g0 > best_gain) { best_gain = g0; best = 0; }
                if (g1 > best_gain) { best_gain = g1; best = 1; }
            }
            if (L >= 0) {
                child_box(t, prims, order, t.left[L], gb[2], gt[2]);
                child_box(t, prims, order, t.right[L], gb[3], gt[3]);
                const double al = box_area(bl);
                const double g2 = al - union_area(br, gb[3]);  // R <-> L.left : L' = (R, L.right)
                const double g3 = al - union_area(br, gb[2]);  // R <-> L.right: L' = (L.left, R)
                if (g2 > best_gain) { best_gain = g2; best = 2; }
                if (g3 > best_gain) { best_gain = g3; best = 3; }
            }
            if (best >= 0) {
                const bool right_side = best < 2;          // the node that is rebuilt
                const int32_t N = right_side ? R : L;      // rebuilt node
                const int32_t S = right_side ? L : R;      // p's other child, moves down into N
                const bool take_left = (best & 1) == 0;    // N's left child moves up to p
                const int32_t up = take_left ? t.left[N] : t.right[N];
                const double* stay = gb[(right_side ? 0 : 2) + (take_left ? 1 : 0)];
                const bool stay_tri = gt[(right_side ? 0 : 2) + (take_left ? 1 : 0)];
                const double* sb = right_side ? bl : br;
                const bool s_tri = right_side ? tl : tr;
                if (take_left) t.left[N] = S; else t.right[N] = S;
                set_parent(t, S, N);
                if (right_side) t.left[p] = up; else t.right[p] = up;
                set_parent(t, up, p);
                double nb[6];
                for (int a = 0; a < 3; a++) {
                    nb[a] = fmin(sb[a], stay[a]);
                    nb[3 + a] = fmax(sb[3 + a], stay[3 + a]);
                }
                for (int a = 0; a < 6; a++) t.box[(size_t)N * 6 + a] = nb[a];
                t.all_tri[N] = (s_tri && stay_tri) ? 1u : 0u;
                __threadfence();
                // p's two children are now `up` and N: re-read their boxes for the union below
                child_box(t, prims, order, t.left[p], bl, tl);
                child_box(t, prims, order, t.right[p], br, tr);
            }
        }
        for (int a = 0; a < 3; a++) {
            t.box[(size_t)p * 6 + a] = fmin(bl[a], br[a]);
            t.box[(size_t)p * 6 + 3 + a] = fmax(bl[3 + a], br[3 + a]);
        }
        t.all_tri[p] = (tl && tr) ? 1u : 0u;
        p = t.parent[p];
    }
}

// ---- SAH top (HLBVH-style): the Morton-order tree splits space at fixed bit planes, which is poor exactly where it
// matters most -- near the root, where every ray passes.  The tree is therefore cut where subtrees get small
// (<= `limit` primitives): the cut elements (clusters and stray single primitives) become the leaves of a binned-SAH
// binary tree built on the HOST over a few thousand boxes (bvh_build.cpp: build_sah_binary, ~1 ms), and the radix
// tree's own nodes above the cut -- exactly one fewer than there are cut elements -- are re-linked to that topology.
// Below the cut the Morton order stays (plus the refit's rotations).  One primitive per leaf only (see kb_refit).
__global__ __launch_bounds__(256) void kb_cut(const rt_primitive* __restrict__ prims, const uint32_t* __restrict__ order,
                                              int n, Tree t, uint32_t limit, int32_t* refs, double* boxes,
                                              int32_t* top_nodes, uint32_t* counts, uint32_t cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    if (t.last[i] - t.first[i] + 1u <= limit) return;  // below the cut
    const uint32_t k = atomicAdd(&counts[1], 1u);
    if (k < cap) top_nodes[k] = i;
    const int32_t ch[2] = {t.left[i], t.right[i]};
    for (int c = 0; c < 2; c++) {
        const int32_t r = ch[c];
        if (r >= 0 && t.last[r] - t.first[r] + 1u > limit) continue;  // another top node
        const uint32_t k2 = atomicAdd(&counts[0], 1u);
        if (k2 > cap) continue;  // (cap + 1 elements fit)
        refs[k2] = r;
        double b[6];
        bool tri;
        child_box(t, prims, order, r, b, tri);
        for (int a = 0; a < 6; a++) boxes[(size_t)k2 * 6 + a] = b[a];
    }
}
__global__ __launch_bounds__(256) void kb_relink(Tree t, const int32_t* __restrict__ node, const int32_t* __restrict__ nl,
                                                 const int32_t* __restrict__ nr, uint32_t m) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int32_t p = node[i];
    t.left[p] = nl[i];
    t.right[p] = nr[i];
    set_parent(t, nl[i], p);
    set_parent(t, nr[i], p);
}

// ---- SAH bottom: the topology INSIDE every cluster below the cut is rebuilt by a binned-SAH top-down build (16 bins,
// three axes, cost = area * count per side, down to one primitive per leaf -- the host builder's rule without its
// outlier candidate).  A cluster is a subtree of the radix tree: it covers the sorted positions [lo, hi] and owns exactly
// the internal nodes {lo .. hi-1} when its root is node lo, {lo+1 .. hi} when its root is node hi (Karras' numbering: a
// left child is the LAST index of its range, a right child the FIRST) -- hi - lo nodes for hi - lo + 1 leaves, so the new
// topology is written into the cluster's own nodes in preorder and the root keeps its id (and its link from above).
// One wave per cluster, everything in LDS (binary32 copies of the leaf boxes: decisions only, the exact boxes come from
// the refit that follows):
//   phase 1, ranges of more than kSerialMax primitives, the wave together: centroid bounds by a butterfly; lane
//            (axis, bin) of 48 scans the range and keeps its bin's box and count; prefix / suffix unions per lane; wave
//            arg-min of the cost (ties: lowest axis, then lowest bin -- the serial scan's order); ballot partition;
//   phase 2, the ranges that are left (<= kSerialMax), one lane each: the same build serially.
// Both process the smaller half first, so the explicit stacks hold <= log2 entries.
constexpr int kClusterBins = 16;
constexpr int kClusterMax = 1024;  // primitives per cluster (LDS: 24 B of boxes each); larger clusters keep the radix topology
constexpr int kSerialMax = 32;

struct ClusterLds {
    float bx[6][kClusterMax];
    uint16_t perm[kClusterMax];  // local leaf index per position
    uint16_t tmp[kClusterMax];
    uint32_t task_ab[kClusterMax / 2];  // a | b << 16
    uint16_t task_k[kClusterMax / 2];
    float bin[3 * kClusterBins][6];
    uint32_t bin_cnt[3 * kClusterBins];
    uint32_t stack_ab[16];
    uint16_t stack_k[16];
};

struct ClusterIds {
    uint32_t lo, hi;
    bool root_is_lo;
    __device__ int32_t node(uint32_t k) const {  // k-th node of the cluster in preorder, 0 = its root
        return (int32_t)(root_is_lo ? lo + k : (k == 0u ? hi : lo + k));
    }
};

__device__ inline float half_area3(const float* b) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
}
__device__ inline int cluster_bin(float c2, float cmn, float scale) {
    const int k = (int)((c2 - cmn) * scale);
    return k < 0 ? 0 : (k > kClusterBins - 1 ? kClusterBins - 1 : k);
}
// node k of the cluster splits positions [a, mid) | [mid, b)
__device__ inline void cluster_link(const Tree& t, const ClusterLds& L, const ClusterIds& id, uint32_t a, uint32_t mid,
                                    uint32_t b, uint32_t k) {
    const int32_t me = id.node(k);
    const uint32_t nl = mid - a, nr = b - mid;
    const int32_t lc = nl == 1u ? ~(int32_t)(id.lo + L.perm[a]) : id.node(k + 1u);
    const int32_t rc = nr == 1u ? ~(int32_t)(id.lo + L.perm[mid]) : id.node(k + nl);
    t.left[me] = lc;
    t.right[me] = rc;
    set_parent(t, lc, me);
    set_parent(t, rc, me);
}

// phase 2: one lane builds positions [a0, b0) (node k0) on its own
__device__ void cluster_serial(const Tree& t, ClusterLds& L, const ClusterIds& id, uint32_t a0, uint32_t b0, uint32_t k0) {
    struct Range {
        uint32_t a, b, k;
    };
    Range stack[8];  // smaller half first: <= log2(kSerialMax) pending
    int sp = 0;
    Range rg{a0, b0, k0};
    for (;;) {
        const uint32_t a = rg.a, b = rg.b, n = b - a;
        uint32_t mid = a + n / 2u;
        if (n > 2u) {
            float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t i = a; i < b; i++) {
                const uint32_t p = L.perm[i];
                for (int x = 0; x < 3; x++) {
                    const float c = L.bx[x][p] + L.bx[3 + x][p];
                    cmn[x] = fminf(cmn[x], c);
                    cmx[x] = fmaxf(cmx[x], c);
                }
            }
            float best_cost = INFINITY;
            int best_axis = -1, best_bin = -1;
            for (int x = 0; x < 3; x++) {
                const float ext = cmx[x] - cmn[x];
                if (!(ext > 0.0f)) continue;
                const float scale = (float)kClusterBins / ext;
                float bb[kClusterBins][6];
                uint32_t cnt[kClusterBins];
                for (int k = 0; k < kClusterBins; k++) {
                    for (int c = 0; c < 3; c++) {
                        bb[k][c] = INFINITY;
                        bb[k][3 + c] = -INFINITY;
                    }
                    cnt[k] = 0;
                }
                for (uint32_t i = a; i < b; i++) {
                    const uint32_t p = L.perm[i];
                    const int k = cluster_bin(L.bx[x][p] + L.bx[3 + x][p], cmn[x], scale);
                    cnt[k]++;
                    for (int c = 0; c < 3; c++) {
                        bb[k][c] = fminf(bb[k][c], L.bx[c][p]);
                        bb[k][3 + c] = fmaxf(bb[k][3 + c], L.bx[3 + c][p]);
                    }
                }
                float right_cost[kClusterBins];  // area * count of bins k .. last (< 0: empty)
                float acc[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                uint32_t c = 0;
                for (int k = kClusterBins - 1; k > 0; k--) {
                    if (cnt[k]) {
                        for (int d = 0; d < 3; d++) {
                            acc[d] = fminf(acc[d], bb[k][d]);
                            acc[3 + d] = fmaxf(acc[3 + d], bb[k][3 + d]);
                        }
                        c += cnt[k];
                    }
                    right_cost[k] = c ? half_area3(acc) * (float)c : -1.0f;
                }
                for (int d = 0; d < 3; d++) {
                    acc[d] = INFINITY;
                    acc[3 + d] = -INFINITY;
                }
                c = 0;
                for (int k = 0; k < kClusterBins - 1; k++) {
                    if (cnt[k]) {
                        for (int d = 0; d < 3; d++) {
                            acc[d] = fminf(acc[d], bb[k][d]);
                            acc[3 + d] = fmaxf(acc[3 + d], bb[k][3 + d]);
                        }
                        c += cnt[k];
                    }
                    if (c == 0u || right_cost[k + 1] < 0.0f) continue;
                    const float cost = half_area3(acc) * (float)c + right_cost[k + 1];
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = x;
                        best_bin = k;
                    }
                }
            }
            if (best_axis >= 0) {
                const float scale = (float)kClusterBins / (cmx[best_axis] - cmn[best_axis]);
                uint32_t i = a, j = b;  // in-place partition: bins <= best_bin first
                while (i < j) {
                    const uint16_t pi = L.perm[i];
                    if (cluster_bin(L.bx[best_axis][pi] + L.bx[3 + best_axis][pi], cmn[best_axis], scale) <= best_bin) {
                        i++;
                    } else {
                        j--;
                        L.perm[i] = L.perm[j];
                        L.perm[j] = pi;
                    }
                }
                if (i > a && i < b) mid = i;  // (always: both sides of a candidate hold primitives)
            }
        }
        cluster_link(t, L, id, a, mid, b, rg.k);
        const uint32_t nl = mid - a, nr = b - mid;
        const Range rl{a, mid, rg.k + 1u}, rr{mid, b, rg.k + nl};
        const bool l_in = nl >= 2u, r_in = nr >= 2u;
        if (l_in && r_in) {
            if (nl <= nr) {
                stack[sp++] = rr;
                rg = rl;
            } else {
                stack[sp++] = rl;
                rg = rr;
            }
        } else if (l_in) {
            rg = rl;
        } else if (r_in) {
            rg = rr;
        } else if (sp > 0) {
            rg = stack[--sp];
        } else {
            break;
        }
    }
}

__global__ __launch_bounds__(64) void kb_cluster_sah(const rt_primitive* __restrict__ prims,
                                                     const uint32_t* __restrict__ order, Tree t,
                                                     const int32_t* __restrict__ refs, uint32_t n_cut, uint32_t* rebuilt) {
    __shared__ ClusterLds L;
    const int32_t root = refs[blockIdx.x];
    if (root < 0) return;  // a stray single primitive
    ClusterIds id;
    id.lo = t.first[root];
    id.hi = t.last[root];
    id.root_is_lo = (uint32_t)root == id.lo;
    const uint32_t m = id.hi - id.lo + 1u;
    if (m < 3u || m > (uint32_t)kClusterMax) return;  // two leaves have one topology
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < m; i += 64u) {
        const rt_primitive& p = prims[order[id.lo + i]];
        for (int x = 0; x < 3; x++) {
            L.bx[x][i] = (float)p.bbox_min[x];
            L.bx[3 + x][i] = (float)p.bbox_max[x];
        }
        L.perm[i] = (uint16_t)i;
    }
    uint32_t n_tasks = 0;
    int sp = 0;
    if (m <= (uint32_t)kSerialMax) {
        L.task_ab[0] = m << 16;
        L.task_k[0] = 0;
        n_tasks = 1;
    } else {
        L.stack_ab[0] = m << 16;
        L.stack_k[0] = 0;
        sp = 1;
    }
    __syncthreads();
    // ---- phase 1 (every value that steers the control flow is wave-uniform)
    const int ax = lane < 48u ? (int)(lane >> 4) : 2, kbin = (int)(lane & 15u);
    while (sp > 0) {
        --sp;
        uint32_t a = L.stack_ab[sp] & 0xffffu, b = L.stack_ab[sp] >> 16, k = L.stack_k[sp];
        for (;;) {  // [a, b) has more than kSerialMax primitives
            const uint32_t n = b - a;
            float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t i = a + lane; i < b; i += 64u) {
                const uint32_t p = L.perm[i];
                for (int x = 0; x < 3; x++) {
                    const float c = L.bx[x][p] + L.bx[3 + x][p];
                    cmn[x] = fminf(cmn[x], c);
                    cmx[x] = fmaxf(cmx[x], c);
                }
            }
            for (int off = 32; off >= 1; off >>= 1)
                for (int x = 0; x < 3; x++) {
                    cmn[x] = fminf(cmn[x], __shfl_xor(cmn[x], off));
                    cmx[x] = fmaxf(cmx[x], __shfl_xor(cmx[x], off));
                }
            const float my_mn = ax == 0 ? cmn[0] : (ax == 1 ? cmn[1] : cmn[2]);
            const float my_ext = (ax == 0 ? cmx[0] : (ax == 1 ? cmx[1] : cmx[2])) - my_mn;
            const bool axis_ok = my_ext > 0.0f;
            const float my_scale = axis_ok ? (float)kClusterBins / my_ext : 0.0f;
            float bb[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
            uint32_t cnt = 0;
            for (uint32_t i = a; i < b; i++) {
                const uint32_t p = L.perm[i];
                if (cluster_bin(L.bx[ax][p] + L.bx[3 + ax][p], my_mn, my_scale) == kbin) {
                    cnt++;
                    for (int c = 0; c < 3; c++) {
                        bb[c] = fminf(bb[c], L.bx[c][p]);
                        bb[3 + c] = fmaxf(bb[3 + c], L.bx[3 + c][p]);
                    }
                }
            }
            if (lane < 48u) {
                for (int c = 0; c < 6; c++) L.bin[lane][c] = bb[c];
                L.bin_cnt[lane] = cnt;
            }
            __syncthreads();
            float cost = INFINITY;
            if (lane < 48u && axis_ok && kbin < kClusterBins - 1) {
                float lb[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                float rb[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                uint32_t cl = 0, cr = 0;
                for (int j = 0; j < kClusterBins; j++) {
                    const int q = ax * kClusterBins + j;
                    const uint32_t cj = L.bin_cnt[q];
                    if (!cj) continue;
                    if (j <= kbin) {
                        cl += cj;
                        for (int c = 0; c < 3; c++) {
                            lb[c] = fminf(lb[c], L.bin[q][c]);
                            lb[3 + c] = fmaxf(lb[3 + c], L.bin[q][3 + c]);
                        }
                    } else {
                        cr += cj;
                        for (int c = 0; c < 3; c++) {
                            rb[c] = fminf(rb[c], L.bin[q][c]);
                            rb[3 + c] = fmaxf(rb[3 + c], L.bin[q][3 + c]);
                        }
                    }
                }
                if (cl && cr) cost = half_area3(lb) * (float)cl + half_area3(rb) * (float)cr;
            }
            uint32_t who = lane;
            for (int off = 32; off >= 1; off >>= 1) {
                const float oc = __shfl_xor(cost, off);
                const uint32_t ow = __shfl_xor(who, off);
                if (oc < cost || (oc == cost && ow < who)) {
                    cost = oc;
                    who = ow;
                }
            }
            uint32_t mid = a + n / 2u;
            if (cost < INFINITY) {
                const int best_axis = (int)(who >> 4), best_bin = (int)(who & 15u);
                const float bmn = best_axis == 0 ? cmn[0] : (best_axis == 1 ? cmn[1] : cmn[2]);
                const float bscale = (float)kClusterBins / ((best_axis == 0 ? cmx[0] : (best_axis == 1 ? cmx[1] : cmx[2])) - bmn);
                uint32_t total_left = 0;
                for (uint32_t base = a; base < b; base += 64u) {
                    const uint32_t i = base + lane;
                    const uint32_t p = i < b ? L.perm[i] : 0u;
                    const bool goes_left = i < b && cluster_bin(L.bx[best_axis][p] + L.bx[3 + best_axis][p], bmn, bscale) <= best_bin;
                    total_left += (uint32_t)__popcll(__ballot(goes_left));
                }
                if (total_left > 0u && total_left < n) {
                    uint32_t lcur = a, rcur = a + total_left;
                    const unsigned long long below = (1ull << lane) - 1ull;
                    for (uint32_t base = a; base < b; base += 64u) {
                        const uint32_t i = base + lane;
                        const bool in = i < b;
                        const uint32_t p = in ? L.perm[i] : 0u;
                        const bool goes_left = in && cluster_bin(L.bx[best_axis][p] + L.bx[3 + best_axis][p], bmn, bscale) <= best_bin;
                        const unsigned long long bl = __ballot(goes_left), br = __ballot(in && !goes_left);
                        if (goes_left)
                            L.tmp[lcur + (uint32_t)__popcll(bl & below)] = (uint16_t)p;
                        else if (in)
                            L.tmp[rcur + (uint32_t)__popcll(br & below)] = (uint16_t)p;
                        lcur += (uint32_t)__popcll(bl);
                        rcur += (uint32_t)__popcll(br);
                    }
                    __syncthreads();
                    for (uint32_t i = a + lane; i < b; i += 64u) L.perm[i] = L.tmp[i];
                    mid = a + total_left;
                }
            }
            __syncthreads();
            if (lane == 0u) cluster_link(t, L, id, a, mid, b, k);
            // children: a single primitive is linked already; small ranges become lane tasks; the larger of two big ones waits
            const uint32_t nl = mid - a, nr = b - mid;
            const uint32_t kl = k + 1u, kr = k + nl;
            const bool l_big = nl > (uint32_t)kSerialMax, r_big = nr > (uint32_t)kSerialMax;
            if (!l_big && nl >= 2u) {
                L.task_ab[n_tasks] = a | (mid << 16);
                L.task_k[n_tasks] = (uint16_t)kl;
                n_tasks++;
            }
            if (!r_big && nr >= 2u) {
                L.task_ab[n_tasks] = mid | (b << 16);
                L.task_k[n_tasks] = (uint16_t)kr;
                n_tasks++;
            }
            if (l_big && r_big) {
                if (nl <= nr) {
                    L.stack_ab[sp] = mid | (b << 16);
                    L.stack_k[sp] = (uint16_t)kr;
                    sp++;
                    b = mid;
                    k = kl;
                } else {
                    L.stack_ab[sp] = a | (mid << 16);
                    L.stack_k[sp] = (uint16_t)kl;
                    sp++;
                    a = mid;
                    k = kr;
                }
            } else if (l_big) {
                b = mid;
                k = kl;
            } else if (r_big) {
                a = mid;
                k = kr;
            } else {
                break;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    // ---- phase 2
    for (uint32_t task = lane; task < n_tasks; task += 64u)
        cluster_serial(t, L, id, L.task_ab[task] & 0xffffu, L.task_ab[task] >> 16, L.task_k[task]);
    if (lane == 0u) atomicAdd(rebuilt, 1u);
}

// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner 2018), the device builder's optional fast topology.
// Bottom-up agglomeration along the Morton order: every cluster looks for the neighbour within +-radius positions whose
// union with it has the smallest surface area; mutual nearest neighbours merge into a new node; the array is compacted;
// repeat until one cluster is left.  Unlike the radix tree it decides with AREAS, so a huge primitive (the floor)
// simply stays unmerged until the end and hangs off the root, and clusters that straddle a Morton bit plane still find
// each other.  Deterministic: ties go to the lower position, node numbers come from an atomic counter but the
// topology does not depend on them.  One primitive per leaf.
struct PlocArrays {
    double* box;    // [count][6]
    int32_t* ref;   // >= 0 internal node, < 0 ~sorted leaf position
};
__global__ __launch_bounds__(256) void kp_init(const rt_primitive* __restrict__ prims, const uint32_t* __restrict__ order,
                                               uint32_t n, PlocArrays a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const rt_primitive& p = prims[order[i]];
    for (int k = 0; k < 3; k++) {
        a.box[(size_t)i * 6 + k] = p.bbox_min[k];
        a.box[(size_t)i * 6 + 3 + k] = p.bbox_max[k];
    }
    a.ref[i] = ~(int32_t)i;
}
constexpr int kPlocMaxRadius = 64;
__global__ __launch_bounds__(256) void kp_nearest(PlocArrays a, const uint32_t* __restrict__ count_p, int radius,
                                                  uint32_t* nn) {
    __shared__ double s_box[(256 + 2 * kPlocMaxRadius) * 6];
    const uint32_t count = *count_p;
    const int base = (int)(blockIdx.x * 256u) - radius;
    if (blockIdx.x * 256u >= count) return;
    const int span = 256 + 2 * radius;
    for (int k = threadIdx.x; k < span * 6; k += 256) {
        const int c = base + k / 6;
        s_box[k] = (c >= 0 && (uint32_t)c < count) ? a.box[(size_t)c * 6 + k % 6] : 0.0;
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const double* me = &s_box[(threadIdx.x + radius) * 6];
    double best = 1e308;
    uint32_t best_j = i;
    for (int d = -radius; d <= radius; d++) {
        if (d == 0) continue;
        const long long j = (long long)i + d;
        if (j < 0 || j >= (long long)count) continue;
        const double ar = union_area(me, &s_box[(threadIdx.x + radius + d) * 6]);
        if (ar < best) {  // strict: the lowest position wins a tie
            best = ar;
            best_j = (uint32_t)j;
        }
    }
    nn[i] = best_j;
}
// merge mutual pairs (the lower position keeps the merged cluster, the higher one is dropped) and count survivors per block
__global__ __launch_bounds__(256) void kp_merge(PlocArrays a, const uint32_t* __restrict__ count_p,
                                                const uint32_t* __restrict__ nn, Tree t, const rt_primitive* __restrict__ prims,
                                                const uint32_t* __restrict__ order, uint32_t* node_counter, uint32_t* keep,
                                                uint32_t* block_sum) {
    const uint32_t count = *count_p;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t k = 0;
    if (i < count) {
        const uint32_t j = nn[i];
        const bool mutual = j != i && nn[j] == i;
        k = 1;
        if (mutual && i > j) k = 0;
        if (mutual && i < j) {
            const uint32_t id = atomicAdd(node_counter, 1u);
            const int32_t l = a.ref[i], r = a.ref[j];
            t.left[id] = l;
            t.right[id] = r;
            set_parent(t, l, (int32_t)id);
            set_parent(t, r, (int32_t)id);
            bool tri = true;
            for (int c = 0; c < 2; c++) {
                const int32_t ch = c ? r : l;
                tri = tri && (ch >= 0 ? t.all_tri[ch] != 0u : prims[order[~ch]].kind == RT_PRIM_TRIANGLE);
            }
            t.all_tri[id] = tri ? 1u : 0u;
            t.first[id] = 0u;  // (ranges of sorted positions mean nothing in this tree: never "leafable", kb_emit)
            t.last[id] = 1u;
            for (int c = 0; c < 3; c++) {
                const double lo = fmin(a.box[(size_t)i * 6 + c], a.box[(size_t)j * 6 + c]);
                const double hi = fmax(a.box[(size_t)i * 6 + 3 + c], a.box[(size_t)j * 6 + 3 + c]);
                t.box[(size_t)id * 6 + c] = lo;
                t.box[(size_t)id * 6 + 3 + c] = hi;
            }
            // the merged cluster is published by kp_compact from t.box (this kernel must not overwrite a.box: the
            // partner's thread may still be reading it)
            keep[i] = 2u + id;  // >= 2: survivor that became node `id`
        } else
            keep[i] = k;
    }
    __shared__ uint32_t s_cnt[4];
    const unsigned long long m = __ballot(k != 0u);
    if ((threadIdx.x & 63u) == 0) s_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_sum[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}
__global__ __launch_bounds__(256) void kp_compact(PlocArrays in, PlocArrays out, uint32_t* count_p, Tree t,
                                                  const uint32_t* __restrict__ keep, const uint32_t* __restrict__ block_off,
                                                  uint32_t n_blocks_total) {
    const uint32_t count = *count_p;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t kv = i < count ? keep[i] : 0u;
    const bool k = kv != 0u;
    __shared__ uint32_t s_cnt[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(k);
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = block_off[blockIdx.x];
    for (uint32_t w = 0; w < wave; w++) off += s_cnt[w];
    off += (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (k) {
        if (kv >= 2u) {
            const uint32_t id = kv - 2u;
            for (int c = 0; c < 6; c++) out.box[(size_t)off * 6 + c] = t.box[(size_t)id * 6 + c];
            out.ref[off] = (int32_t)id;
        } else {
            for (int c = 0; c < 6; c++) out.box[(size_t)off * 6 + c] = in.box[(size_t)i * 6 + c];
            out.ref[off] = in.ref[i];
        }
    }
    (void)n_blocks_total;
}
// the new count = survivors of all blocks (kb_scan has turned block_sum into exclusive offsets; `last` = the last block's own sum)
__global__ void kp_count(uint32_t* count_p, const uint32_t* __restrict__ block_off, const uint32_t* __restrict__ last_sum,
                         uint32_t n_blocks) {
    *count_p = block_off[n_blocks - 1] + *last_sum;
}
__global__ void kp_save_last(const uint32_t* __restrict__ block_sum, uint32_t n_blocks, uint32_t* last_sum) {
    *last_sum = block_sum[n_blocks - 1];
}
// the tree is complete: its root is the one cluster left; the emission queue starts there
__global__ void kp_finish(PlocArrays a, Tree t, ull* queue0, int32_t* root_out) {
    const int32_t root = a.ref[0];
    t.parent[root] = -1;
    queue0[0] = (ull)(uint32_t)root;  // binary root -> 4-wide node 0
    *root_out = root;
}

// ---- 4-wide emission, one level per launch.  Work item = (binary node) | (4-wide node index << 32).
__device__ inline bool leafable(const Tree& t, int32_t c) {
    return t.all_tri[c] != 0u && t.last[c] - t.first[c] + 1u <= (uint32_t)kLeafTargetPrims;
}

__global__ __launch_bounds__(256) void kb_emit(const rt_primitive* __restrict__ prims,
                                               const uint32_t* __restrict__ order, Tree t, const ull* q_in, ull* q_out,
                                               BuildGlobals* g, int cur, DevNode* nodes, uint32_t max_nodes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g->q_count[cur]) return;
    const ull item = q_in[i];
    const int32_t b = (int32_t)(uint32_t)(item & 0xffffffffull);
    const uint32_t out = (uint32_t)(item >> 32);
    int32_t cand[4];
    int nc;
    if (leafable(t, b)) {  // only the root can arrive here: the whole scene is one leaf, wrapped in node 0
        cand[0] = b;
        nc = 1;
    } else {
        cand[0] = t.left[b];
        cand[1] = t.right[b];
        nc = 2;
        while (nc < 4) {
            int best = -1;
            double best_a = -1.0;
            for (int k = 0; k < nc; k++) {
                const int32_t c = cand[k];
                if (c >= 0 && !leafable(t, c)) {
                    const double a = box_area(&t.box[(size_t)c * 6]);
                    if (a > best_a) {
                        best_a = a;
                        best = k;
                    }
                }
            }
            if (best < 0) break;
            const int32_t c = cand[best];
            cand[best] = t.left[c];
            cand[nc++] = t.right[c];
        }
    }
    DevNode nd;
    for (int k = 0; k < 4; k++) {
        nd.pad[k] = 0;
        if (k >= nc) {
            nd.child[k] = kNoChild;
            nd.lo_x[k] = nd.lo_y[k] = nd.lo_z[k] = 0.0f;
            nd.hi_x[k] = nd.hi_y[k] = nd.hi_z[k] = 0.0f;
            continue;
        }
        const int32_t c = cand[k];
        double bx[6];
        if (c >= 0) {
            for (int a = 0; a < 6; a++) bx[a] = t.box[(size_t)c * 6 + a];
        } else {
            const rt_primitive& p = prims[order[~c]];
            for (int a = 0; a < 3; a++) {
                bx[a] = p.bbox_min[a];
                bx[3 + a] = p.bbox_max[a];
            }
        }
        nd.lo_x[k] = f_down(bx[0]); nd.lo_y[k] = f_down(bx[1]); nd.lo_z[k] = f_down(bx[2]);
        nd.hi_x[k] = f_up(bx[3]);   nd.hi_y[k] = f_up(bx[4]);   nd.hi_z[k] = f_up(bx[5]);
        if (c < 0) {
            const uint32_t j = (uint32_t)~c;
            const bool other = prims[order[j]].kind != RT_PRIM_TRIANGLE;
            nd.child[k] = -1 - (int32_t)((j * 8u) | (other ? kLeafCodeOther : 0u));
        } else if (leafable(t, c)) {
            const uint32_t first = t.first[c], count = t.last[c] - first + 1u;
            nd.child[k] = -1 - (int32_t)(first * 8u + (count - 1u));
        } else {
            const uint32_t idx = atomicAdd(&g->node_count, 1u);
            if (idx >= max_nodes) {
                g->overflow = 1;
                nd.child[k] = kNoChild;
                continue;
            }
            nd.child[k] = (int32_t)idx;
            const uint32_t slot = atomicAdd(&g->q_count[cur ^ 1], 1u);
            q_out[slot] = (ull)(uint32_t)c | ((ull)idx << 32);
        }
    }
    nodes[out] = nd;
}

__global__ void kb_next_level(BuildGlobals* g, int cur) { g->q_count[cur] = 0; }
__global__ void kb_reset_emit(BuildGlobals* g) {  // a second emission (the rotation-free retry) starts over
    g->node_count = 1;
    g->q_count[0] = 1;
    g->q_count[1] = 0;
    g->overflow = 0;
}

// the scene is a single primitive: node 0 wraps its leaf
__global__ void kb_single(const rt_primitive* __restrict__ prims, DevNode* nodes) {
    const rt_primitive& p = prims[0];
    DevNode nd;
    for (int k = 0; k < 4; k++) {
        nd.pad[k] = 0;
        nd.child[k] = kNoChild;
        nd.lo_x[k] = nd.lo_y[k] = nd.lo_z[k] = 0.0f;
        nd.hi_x[k] = nd.hi_y[k] = nd.hi_z[k] = 0.0f;
    }
    nd.lo_x[0] = f_down(p.bbox_min[0]); nd.lo_y[0] = f_down(p.bbox_min[1]); nd.lo_z[0] = f_down(p.bbox_min[2]);
    nd.hi_x[0] = f_up(p.bbox_max[0]);   nd.hi_y[0] = f_up(p.bbox_max[1]);   nd.hi_z[0] = f_up(p.bbox_max[2]);
    nd.child[0] = -1 - (int32_t)(p.kind != RT_PRIM_TRIANGLE ? kLeafCodeOther : 0u);
    nodes[0] = nd;
}

// leaf slots in sorted order (same contents as rt_scene_commit's host loop, abi.hip)
__global__ __launch_bounds__(256) void kb_leaves(const rt_primitive* __restrict__ prims,
                                                 const DevMesh* __restrict__ meshes,
                                                 const uint32_t* __restrict__ order, uint32_t n, uint32_t* leaf_prim,
                                                 double* leaf_tri, double* leaf_nrm, LeafMeta* leaf_meta) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t id = order[i];
    const rt_primitive& p = prims[id];
    double* o = leaf_tri + (size_t)i * 9;
    uint32_t mx = (p.mat_index & kMetaMatMask) | (p.flip ? kMetaFlip : 0u);
    if (leaf_nrm)
        for (int a = 0; a < 9; a++) leaf_nrm[(size_t)i * 9 + a] = 0.0;
    if (p.kind == RT_PRIM_TRIANGLE) {
        const DevMesh& m = meshes[p.mesh_index];
        for (int v = 0; v < 3; v++) {
            const uint32_t vi = m.ind[p.tri_ind + v];
            for (int a = 0; a < 3; a++) o[v * 3 + a] = m.p[3 * (size_t)vi + a];
            if (m.n && leaf_nrm)
                for (int a = 0; a < 3; a++) leaf_nrm[(size_t)i * 9 + v * 3 + a] = m.n[3 * (size_t)vi + a];
        }
        if (m.n && leaf_nrm) mx |= kMetaHasNormals;
        leaf_prim[i] = id;
    } else {
        for (int a = 0; a < 5; a++) o[a] = p.v[a];
        const ull meta = (ull)(p.kind & 0xffu) | ((ull)(uint32_t)(p.xform_index + 1) << 32);
        o[5] = __longlong_as_double((long long)meta);
        o[6] = o[7] = o[8] = 0.0;
        leaf_prim[i] = id | kLeafOther;
    }
    leaf_meta[i] = LeafMeta{mx, p.light_index};
}

struct Scratch {
    std::vector<void*> ptrs;
    ~Scratch() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    hipError_t get(T** out, size_t count) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(p);
        *out = (T*)p;
        return e;
    }
};

int bfail(char* err, size_t n, int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    if (err && n) vsnprintf(err, n, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace

#define B_TRY(expr)                                                                                          \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return bfail(err, err_len, e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, "%s failed: %s (%s:%d)", \
                         #expr, hipGetErrorString(e_), __FILE__, __LINE__);                                  \
    } while (0)

int build_bvh_device(hipStream_t stream, const rt_primitive* d_prims, const DevMesh* d_meshes, uint32_t n,
                     bool any_normals, DeviceBvh* out, char* err, size_t err_len) {
    *out = DeviceBvh{};
    if (n == 0) return RT_OK;
    if (n >= (1u << 27)) return bfail(err, err_len, RT_ERR_UNSUPPORTED, "more than 2^27 primitives");
    Scratch tmp;   // freed on every exit
    Scratch keep;  // results: released to the caller on success
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    B_TRY(hipEventCreate(&ev0));
    B_TRY(hipEventCreate(&ev1));
    struct EvGuard {
        hipEvent_t a, b;
        ~EvGuard() {
            (void)hipEventDestroy(a);
            (void)hipEventDestroy(b);
        }
    } evg{ev0, ev1};
    B_TRY(hipEventRecord(ev0, stream));

    BuildGlobals* g;
    ull *keys[2], *queue[2];
    uint32_t *vals[2], *hist;
    const uint32_t n_tiles = (n + kSortTile - 1) / kSortTile;
    const uint32_t nb = (n + 255u) / 256u;
    B_TRY(tmp.get(&g, 1));
    B_TRY(tmp.get(&keys[0], n));
    B_TRY(tmp.get(&keys[1], n));
    B_TRY(tmp.get(&vals[0], n));
    B_TRY(tmp.get(&vals[1], n));
    B_TRY(tmp.get(&hist, 256 * (size_t)n_tiles));
    hipLaunchKernelGGL(kb_init, dim3(1), dim3(1), 0, stream, g);
    hipLaunchKernelGGL(kb_bounds, dim3(std::min<uint32_t>(nb, 1024u)), dim3(256), 0, stream, d_prims, n, g);
    hipLaunchKernelGGL(kb_morton, dim3(nb), dim3(256), 0, stream, d_prims, n, g, keys[0], vals[0]);
    int cur = 0;
    for (int shift = 0; shift < 64; shift += 8) {
        hipLaunchKernelGGL(kb_hist, dim3(n_tiles), dim3(kSortThreads), 0, stream, keys[cur], n, shift, hist, n_tiles);
        hipLaunchKernelGGL(kb_scan, dim3(1), dim3(1024), 0, stream, hist, 256u * n_tiles);
        hipLaunchKernelGGL(kb_scatter, dim3(n_tiles), dim3(kSortThreads), 0, stream, keys[cur], vals[cur],
                           keys[cur ^ 1], vals[cur ^ 1], n, shift, hist, n_tiles);
        cur ^= 1;
    }
    const ull* skeys = keys[cur];
    const uint32_t* order = vals[cur];
    hipLaunchKernelGGL(kb_check_sorted, dim3(nb), dim3(256), 0, stream, skeys, n, g);

    DeviceBvh r;
    B_TRY(keep.get(&r.leaf_prim, n));
    B_TRY(keep.get(&r.leaf_tri, (size_t)n * 9));
    if (any_normals) B_TRY(keep.get(&r.leaf_nrm, (size_t)n * 9));
    B_TRY(keep.get(&r.leaf_meta, n));
    hipLaunchKernelGGL(kb_leaves, dim3(nb), dim3(256), 0, stream, d_prims, d_meshes, order, n, r.leaf_prim, r.leaf_tri,
                       r.leaf_nrm, r.leaf_meta);

    DevNode* pool = nullptr;  // one 4-wide node per binary internal node at most
    const uint32_t max_nodes = n > 1 ? n - 1 : 1;
    B_TRY(tmp.get(&pool, max_nodes));
    uint32_t levels = 1;
    BuildGlobals hg;
    if (n == 1) {
        hipLaunchKernelGGL(kb_single, dim3(1), dim3(1), 0, stream, d_prims, pool);
        B_TRY(hipMemcpyAsync(&hg, g, sizeof(hg), hipMemcpyDeviceToHost, stream));
        B_TRY(hipStreamSynchronize(stream));
        hg.node_count = 1;
    } else {
        Tree t;
        const size_t ni = n - 1;
        B_TRY(tmp.get(&t.left, ni));
        B_TRY(tmp.get(&t.right, ni));
        B_TRY(tmp.get(&t.parent, ni));
        B_TRY(tmp.get(&t.leaf_par, n));
        B_TRY(tmp.get(&t.first, ni));
        B_TRY(tmp.get(&t.last, ni));
        B_TRY(tmp.get(&t.box, ni * 6));
        B_TRY(tmp.get(&t.flag, ni));
        B_TRY(tmp.get(&t.all_tri, ni));
        B_TRY(tmp.get(&queue[0], ni));
        B_TRY(tmp.get(&queue[1], ni));
        // Topology: the Morton-order radix tree, cut into clusters of <= RT_LBVH_SAH_CLUSTER primitives (default 1024; 0 = no
        // cut); above the cut a binned-SAH tree built on the host over the clusters' boxes (kb_cut / kb_relink), inside
        // every cluster a binned-SAH tree built by one wave (kb_cluster_sah; RT_LBVH_SAH_BOTTOM=0 keeps the Morton order
        // there).  RT_LBVH_ROTATE_PASSES refit passes apply tree rotations on top: default 0 with the SAH bottom (they
        // cost a cornell box 20 % more node fetches there and buy the dragons 0.3 %), 2 without it.  A tree too deep for the
        // traversal stack is rebuilt plain (no rotations, no SAH) before giving up.
        const uint32_t sah_limit = getenv("RT_LBVH_SAH_CLUSTER") ? (uint32_t)atoi(getenv("RT_LBVH_SAH_CLUSTER")) : 1024u;
        const bool cluster_sah = getenv("RT_LBVH_SAH_BOTTOM") ? atoi(getenv("RT_LBVH_SAH_BOTTOM")) != 0 : true;
        const bool sah_applies = kLeafTargetPrims == 1 && sah_limit > 0 && n > 4u * sah_limit;
        const int rot_default = getenv("RT_LBVH_ROTATE_PASSES") ? atoi(getenv("RT_LBVH_ROTATE_PASSES")) : (sah_applies && cluster_sah ? 0 : 2);
        // ---- RT_DEVICE_BUILDER=ploc: the PLOC topology instead of the Morton-order tree (the fastest build -- C4 10.6 ms
        // against 18.8 -- at the plain Morton tree's quality: 1.18-1.20 x the host tree's node fetches, where the default
        // reaches 0.92-1.04 in 10-39 ms; profiles/r03_lbvh_ploc_sweep.txt, r03_lbvh_sah_bottom.txt).  Falls back to the
        // default when too deep.
        const char* which = getenv("RT_DEVICE_BUILDER");
        bool ploc_done = false;
        if (kLeafTargetPrims == 1 && which && std::strcmp(which, "ploc") == 0) {
            const int radius = std::min(kPlocMaxRadius, std::max(1, getenv("RT_PLOC_RADIUS") ? atoi(getenv("RT_PLOC_RADIUS")) : 16));
            const int ploc_rot = getenv("RT_PLOC_ROTATE_PASSES") ? atoi(getenv("RT_PLOC_ROTATE_PASSES")) : 1;
            PlocArrays pa[2];
            uint32_t *d_nn = nullptr, *d_keep = nullptr, *d_bsum = nullptr, *d_cnt = nullptr, *d_nodes = nullptr, *d_last = nullptr;
            int32_t* d_root = nullptr;
            for (int k = 0; k < 2; k++) {
                B_TRY(tmp.get(&pa[k].box, (size_t)n * 6));
                B_TRY(tmp.get(&pa[k].ref, n));
            }
            B_TRY(tmp.get(&d_nn, n));
            B_TRY(tmp.get(&d_keep, n));
            B_TRY(tmp.get(&d_bsum, nb + 1));
            B_TRY(tmp.get(&d_cnt, 1));
            B_TRY(tmp.get(&d_nodes, 1));
            B_TRY(tmp.get(&d_last, 1));
            B_TRY(tmp.get(&d_root, 1));
            B_TRY(hipMemcpyAsync(d_cnt, &n, sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            B_TRY(hipMemsetAsync(d_nodes, 0, sizeof(uint32_t), stream));
            hipLaunchKernelGGL(kp_init, dim3(nb), dim3(256), 0, stream, d_prims, order, n, pa[0]);
            uint32_t count = n;
            int cur_a = 0, passes = 0;
            bool stuck = false;
            while (count > 1) {
                const uint32_t blocks = (count + 255u) / 256u;
                hipLaunchKernelGGL(kp_nearest, dim3(blocks), dim3(256), 0, stream, pa[cur_a], d_cnt, radius, d_nn);
                hipLaunchKernelGGL(kp_merge, dim3(blocks), dim3(256), 0, stream, pa[cur_a], d_cnt, d_nn, t, d_prims, order, d_nodes,
                                   d_keep, d_bsum);
                hipLaunchKernelGGL(kp_save_last, dim3(1), dim3(1), 0, stream, d_bsum, blocks, d_last);
                hipLaunchKernelGGL(kb_scan, dim3(1), dim3(1024), 0, stream, d_bsum, blocks);
                hipLaunchKernelGGL(kp_compact, dim3(blocks), dim3(256), 0, stream, pa[cur_a], pa[cur_a ^ 1], d_cnt, t, d_keep, d_bsum,
                                   blocks);
                hipLaunchKernelGGL(kp_count, dim3(1), dim3(1), 0, stream, d_cnt, d_bsum, d_last, blocks);
                cur_a ^= 1;
                uint32_t next = 0;
                B_TRY(hipMemcpyAsync(&next, d_cnt, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                B_TRY(hipStreamSynchronize(stream));
                if (next >= count || ++passes > 4096) {  // (cannot happen: the closest pair is always mutual)
                    stuck = true;
                    break;
                }
                count = next;
            }
            if (!stuck) {
                hipLaunchKernelGGL(kb_reset_emit, dim3(1), dim3(1), 0, stream, g);
                hipLaunchKernelGGL(kp_finish, dim3(1), dim3(1), 0, stream, pa[cur_a], t, queue[0], d_root);
                for (int pass = 0; pass < ploc_rot; pass++) {  // the refit's tree rotations on top (boxes recomputed: same values)
                    B_TRY(hipMemsetAsync(t.flag, 0, ni * sizeof(uint32_t), stream));
                    hipLaunchKernelGGL(kb_refit, dim3(nb), dim3(256), 0, stream, d_prims, order, (int)n, t, 1);
                }
                int q = 0;
                uint32_t cnt = 1;
                levels = 0;
                bool too_deep = false;
                while (cnt > 0) {
                    if (levels >= (uint32_t)kMaxBvhDepth) {
                        too_deep = true;
                        break;
                    }
                    hipLaunchKernelGGL(kb_emit, dim3((cnt + 255u) / 256u), dim3(256), 0, stream, d_prims, order, t, queue[q],
                                       queue[q ^ 1], g, q, pool, max_nodes);
                    hipLaunchKernelGGL(kb_next_level, dim3(1), dim3(1), 0, stream, g, q);
                    B_TRY(hipMemcpyAsync(&hg, g, sizeof(hg), hipMemcpyDeviceToHost, stream));
                    B_TRY(hipStreamSynchronize(stream));
                    cnt = hg.q_count[q ^ 1];
                    q ^= 1;
                    levels++;
                }
                if (!too_deep) {
                    ploc_done = true;
                    r.ploc_passes = (uint32_t)passes;
                    r.rotation_passes = (uint32_t)ploc_rot;
                }
            }
        }
        bool plain_retry = false;  // second attempt: neither rotations nor SAH
        for (int rot_passes = rot_default; !ploc_done; rot_passes = 0, plain_retry = true) {
            B_TRY(hipMemsetAsync(t.flag, 0, ni * sizeof(uint32_t), stream));
            B_TRY(hipMemsetAsync(queue[0], 0, sizeof(ull), stream));  // first item: binary root 0 -> node 0
            hipLaunchKernelGGL(kb_reset_emit, dim3(1), dim3(1), 0, stream, g);
            hipLaunchKernelGGL(kb_tree, dim3(nb), dim3(256), 0, stream, skeys, (int)n, t);
            bool sah_top_done = false;
            r.sah_top_clusters = r.sah_bottom_clusters = 0;
            if (!plain_retry && sah_applies) {
                // plain refit first (the cut needs the clusters' boxes), then the SAH top, then the rotation passes
                hipLaunchKernelGGL(kb_refit, dim3(nb), dim3(256), 0, stream, d_prims, order, (int)n, t, 0);
                const uint32_t cap = 1u << 16;
                int32_t *d_refs = nullptr, *d_top = nullptr;
                double* d_boxes = nullptr;
                uint32_t* d_counts = nullptr;
                B_TRY(tmp.get(&d_refs, cap + 1));
                B_TRY(tmp.get(&d_top, cap));
                B_TRY(tmp.get(&d_boxes, (size_t)(cap + 1) * 6));
                B_TRY(tmp.get(&d_counts, 2));
                B_TRY(hipMemsetAsync(d_counts, 0, 2 * sizeof(uint32_t), stream));
                hipLaunchKernelGGL(kb_cut, dim3(nb), dim3(256), 0, stream, d_prims, order, (int)n, t, sah_limit, d_refs, d_boxes,
                                   d_top, d_counts, cap);
                uint32_t counts[2] = {0, 0};
                B_TRY(hipMemcpyAsync(counts, d_counts, sizeof(counts), hipMemcpyDeviceToHost, stream));
                B_TRY(hipStreamSynchronize(stream));
                const uint32_t n_cut = counts[0], n_top = counts[1];
                if (n_top >= 1 && n_top <= cap && n_cut == n_top + 1) {
                    if (cluster_sah) {  // SAH bottom: every cluster's own nodes re-linked (device), overlaps the host's SAH top
                        uint32_t* d_rebuilt = nullptr;
                        B_TRY(tmp.get(&d_rebuilt, 1));
                        B_TRY(hipMemsetAsync(d_rebuilt, 0, sizeof(uint32_t), stream));
                        hipLaunchKernelGGL(kb_cluster_sah, dim3(n_cut), dim3(64), 0, stream, d_prims, order, t, d_refs, n_cut,
                                           d_rebuilt);
                        B_TRY(hipMemcpyAsync(&r.sah_bottom_clusters, d_rebuilt, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                    }
                    std::vector<int32_t> refs(n_cut), top(n_top), sl, sr;
                    std::vector<double> boxes((size_t)n_cut * 6);
                    B_TRY(hipMemcpyAsync(refs.data(), d_refs, n_cut * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
                    B_TRY(hipMemcpyAsync(top.data(), d_top, n_top * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
                    B_TRY(hipMemcpyAsync(boxes.data(), d_boxes, boxes.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
                    B_TRY(hipStreamSynchronize(stream));
                    build_sah_binary(boxes.data(), n_cut, sl, sr);
                    // local node 0 is the SAH root: it takes the radix tree's root, node 0
                    for (uint32_t k = 0; k < n_top; k++)
                        if (top[k] == 0) {
                            std::swap(top[k], top[0]);
                            break;
                        }
                    if (top[0] == 0) {
                        std::vector<int32_t> packed((size_t)n_top * 2);
                        auto ref = [&](int32_t c) { return c >= 0 ? top[c] : refs[~c]; };
                        for (uint32_t k = 0; k < n_top; k++) {
                            packed[k] = ref(sl[k]);
                            packed[n_top + k] = ref(sr[k]);
                        }
                        // (reuse the device buffers: d_top <- node ids, d_refs is too small for two arrays: the boxes buffer)
                        int32_t* d_lr = reinterpret_cast<int32_t*>(d_boxes);
                        B_TRY(hipMemcpyAsync(d_top, top.data(), n_top * sizeof(int32_t), hipMemcpyHostToDevice, stream));
                        B_TRY(hipMemcpyAsync(d_lr, packed.data(), packed.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
                        hipLaunchKernelGGL(kb_relink, dim3((n_top + 255u) / 256u), dim3(256), 0, stream, t, d_top, d_lr, d_lr + n_top,
                                           n_top);
                        B_TRY(hipStreamSynchronize(stream));  // the host vectors go out of scope
                        sah_top_done = true;
                        r.sah_top_clusters = n_cut;
                    }
                }
                B_TRY(hipMemsetAsync(t.flag, 0, ni * sizeof(uint32_t), stream));
            }
            // (more cut elements than the SAH top takes -- beyond ~30 M primitives: the Morton-order tree with the
            // rotations' default instead)
            if (!plain_retry && sah_applies && !sah_top_done && !getenv("RT_LBVH_ROTATE_PASSES")) rot_passes = 2;
            hipLaunchKernelGGL(kb_refit, dim3(nb), dim3(256), 0, stream, d_prims, order, (int)n, t, rot_passes > 0 ? 1 : 0);
            for (int pass = 1; pass < rot_passes; pass++) {
                B_TRY(hipMemsetAsync(t.flag, 0, ni * sizeof(uint32_t), stream));
                hipLaunchKernelGGL(kb_refit, dim3(nb), dim3(256), 0, stream, d_prims, order, (int)n, t, 1);
            }
            int q = 0;
            uint32_t count = 1;
            levels = 0;
            bool too_deep = false;
            while (count > 0) {
                if (levels >= (uint32_t)kMaxBvhDepth) {
                    too_deep = true;
                    break;
                }
                hipLaunchKernelGGL(kb_emit, dim3((count + 255u) / 256u), dim3(256), 0, stream, d_prims, order, t, queue[q],
                                   queue[q ^ 1], g, q, pool, max_nodes);
                hipLaunchKernelGGL(kb_next_level, dim3(1), dim3(1), 0, stream, g, q);
                B_TRY(hipMemcpyAsync(&hg, g, sizeof(hg), hipMemcpyDeviceToHost, stream));
                B_TRY(hipStreamSynchronize(stream));
                count = hg.q_count[q ^ 1];
                q ^= 1;
                levels++;
            }
            if (!too_deep) {
                r.rotation_passes = (uint32_t)rot_passes;
                break;
            }
            if (plain_retry || (rot_passes == 0 && r.sah_top_clusters == 0))
                return bfail(err, err_len, RT_ERR_UNSUPPORTED,
                             "device-built BVH is deeper than the traversal stack allows (%d levels): commit with the host builder",
                             kMaxBvhDepth);
        }
    }
    B_TRY(hipGetLastError());
    if (hg.unsorted) return bfail(err, err_len, RT_ERR_HIP, "device radix sort self-check failed (%u inversions)", hg.unsorted);
    if (hg.overflow) return bfail(err, err_len, RT_ERR_HIP, "device BVH node pool overflow");
    r.n_nodes = hg.node_count;
    r.depth = levels > 0 ? levels - 1 : 0;
    r.n_triangles = hg.n_tri;
    // the pool was sized for the worst case: keep an exact copy
    B_TRY(keep.get(&r.nodes, r.n_nodes));
    B_TRY(hipMemcpyAsync(r.nodes, pool, (size_t)r.n_nodes * sizeof(DevNode), hipMemcpyDeviceToDevice, stream));
    B_TRY(hipEventRecord(ev1, stream));
    B_TRY(hipStreamSynchronize(stream));
    B_TRY(hipEventElapsedTime(&r.build_ms, ev0, ev1));
    keep.ptrs.clear();  // ownership passes to the caller
    *out = r;
    return RT_OK;
}

}  // namespace rtd

// bvh_quality.cpp -- host-side quality check of the BVH4 the library builds (measurement tool, not product).
//
// Builds a preset with the host scene code of librt_amd.so, runs rtd::build_bvh (the same builder rt_scene_commit
// uses; its RT_BVH_* environment knobs apply) and reports
//   * the SAH cost of the 4-wide tree (expected node fetches + primitive tests of a uniformly random line), and
//   * the node fetches / primitive tests per ray of a nearest-first, pruned traversal (the device's order) for three
//     ray populations of the preset's own camera: primary rays, cosine-distributed bounce rays leaving their hit
//     points, and shadow rays from those points towards the first light's centre.
// Intersections here are plain f64 Moeller-Trumbore / slab tests: good enough to prune like the device does, not a
// parity statement.
//   build:  g++ -O2 -std=c++17 -Iinclude -Irustraytracer_amd/csrc tools/bvh_quality.cpp -o /tmp/bvh_quality \
//               -Lrustraytracer_amd -l:librt_amd.so -Wl,-rpath,$PWD/rustraytracer_amd -Wl,-rpath,/opt/rocm/lib
//   run:    [RT_BVH_BINS=32 ...] /tmp/bvh_quality two_dragons 871414 0 [aspect] [rays]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bvh_build.h"
#include "rt_host.h"

using namespace rtd;

struct V {
    double x, y, z;
};
static V operator+(V a, V b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V operator-(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V operator*(V a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static double dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V cross(V a, V b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static V norm(V a) { return a * (1.0 / std::sqrt(dot(a, a))); }

struct Scene {
    const rt_scene_desc* d;
    BvhOut bvh;
};

static bool hit_prim(const Scene& s, uint32_t pi, V o, V dir, double tmax, double& t, V& n) {
    const rt_primitive& p = s.d->prims[pi];
    if (p.kind == RT_PRIM_TRIANGLE) {
        const rt_mesh& m = s.d->meshes[p.mesh_index];
        const uint32_t* id = m.ind + p.tri_ind;
        auto vtx = [&](uint32_t i) { return V{m.p[3 * i], m.p[3 * i + 1], m.p[3 * i + 2]}; };
        const V a = vtx(id[0]), e1 = vtx(id[1]) - a, e2 = vtx(id[2]) - a, pv = cross(dir, e2);
        const double det = dot(e1, pv);
        if (det == 0.0) return false;
        const double inv = 1.0 / det;
        const V tv = o - a;
        const double u = dot(tv, pv) * inv;
        if (u < 0.0 || u > 1.0) return false;
        const V qv = cross(tv, e1);
        const double v = dot(dir, qv) * inv;
        if (v < 0.0 || u + v > 1.0) return false;
        t = dot(e2, qv) * inv;
        n = norm(cross(e1, e2));
        return t > 1e-4 && t < tmax;
    }
    if (p.kind == RT_PRIM_SPHERE) {
        const V c{p.v[0], p.v[1], p.v[2]}, oc = o - c;
        const double a = dot(dir, dir), hb = dot(oc, dir), cc = dot(oc, oc) - p.v[3] * p.v[3], disc = hb * hb - a * cc;
        if (disc < 0.0) return false;
        const double sq = std::sqrt(disc);
        t = (-hb - sq) / a;
        if (!(t > 1e-3 && t < tmax)) t = (-hb + sq) / a;
        if (!(t > 1e-3 && t < tmax)) return false;
        n = norm(o + dir * t - c);
        return true;
    }
    if (p.xform_index >= 0) {  // transformed rect: its box is a fair stand-in for this purpose
        double tn = 1e-3, tf = tmax;
        const double oo[3] = {o.x, o.y, o.z}, dd[3] = {dir.x, dir.y, dir.z};
        for (int a = 0; a < 3; a++) {
            const double i = 1.0 / dd[a];
            double t0 = (p.bbox_min[a] - oo[a]) * i, t1 = (p.bbox_max[a] - oo[a]) * i;
            if (t0 > t1) std::swap(t0, t1);
            tn = std::max(tn, t0);
            tf = std::min(tf, t1);
        }
        if (tf <= tn) return false;
        t = tn;
        n = V{0, 1, 0};
        return true;
    }
    const int ax = p.kind == RT_PRIM_XY_RECT ? 2 : (p.kind == RT_PRIM_XZ_RECT ? 1 : 0);
    const int a0 = p.kind == RT_PRIM_YZ_RECT ? 1 : 0, a1 = p.kind == RT_PRIM_XY_RECT ? 1 : 2;
    const double oo[3] = {o.x, o.y, o.z}, dd[3] = {dir.x, dir.y, dir.z};
    t = (p.v[4] - oo[ax]) / dd[ax];
    if (!(t > 1e-3 && t < tmax)) return false;
    const double c0 = oo[a0] + t * dd[a0], c1 = oo[a1] + t * dd[a1];
    if (c0 < p.v[0] || c0 > p.v[2] || c1 < p.v[1] || c1 > p.v[3]) return false;
    n = V{ax == 0 ? 1.0 : 0.0, ax == 1 ? 1.0 : 0.0, ax == 2 ? 1.0 : 0.0};
    return true;
}

struct Count {
    unsigned long long rays = 0, nodes = 0, prims = 0;
};

// nearest-first, pruned; returns the closest primitive or -1
static int trace(const Scene& s, V o, V dir, double tmax, double& t_hit, V& n_hit, Count& c) {
    const double inv[3] = {1.0 / dir.x, 1.0 / dir.y, 1.0 / dir.z}, oo[3] = {o.x, o.y, o.z};
    struct E {
        int32_t node;
        double t;
    };
    E stack[256];
    int sp = 0;
    int32_t cur = 0;
    int best = -1;
    double best_t = tmax;
    c.rays++;
    for (;;) {
        if (cur >= 0) {
            const DevNode& nd = s.bvh.nodes[cur];
            c.nodes++;
            E hit[4];
            int nh = 0;
            for (int k = 0; k < 4; k++) {
                if (nd.child[k] == kNoChild) continue;
                const double lo[3] = {nd.lo_x[k], nd.lo_y[k], nd.lo_z[k]}, hi[3] = {nd.hi_x[k], nd.hi_y[k], nd.hi_z[k]};
                double tn = 1e-3, tf = best_t;
                for (int a = 0; a < 3; a++) {
                    double t0 = (lo[a] - oo[a]) * inv[a], t1 = (hi[a] - oo[a]) * inv[a];
                    if (t0 > t1) std::swap(t0, t1);
                    if (t0 > tn) tn = t0;  // NaN-dropping like the device's max/min
                    if (t1 < tf) tf = t1;
                }
                if (!(tf <= tn)) hit[nh++] = E{nd.child[k], tn};
            }
            std::sort(hit, hit + nh, [](const E& a, const E& b) { return a.t > b.t; });  // farthest first onto the stack
            for (int k = 0; k < nh; k++) stack[sp++] = hit[k];
        } else {
            const uint32_t code = (uint32_t)(-1 - cur) & ~kLeafCodeOther;
            const uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
            for (uint32_t i = 0; i < cnt; i++) {
                c.prims++;
                double t;
                V n;
                const uint32_t pi = s.bvh.order[first + i];
                if (hit_prim(s, pi, o, dir, best_t, t, n) && t < best_t) {
                    best_t = t;
                    best = (int)pi;
                    n_hit = n;
                }
            }
        }
        for (;;) {
            if (sp == 0) {
                t_hit = best_t;
                return best;
            }
            const E e = stack[--sp];
            if (e.t <= best_t) {
                cur = e.node;
                break;
            }
        }
    }
}

// ---- what a wider tree would fetch: the same tree re-collapsed to W children per node (a node keeps opening its
// internal child of largest area while the result still fits), traversed the same way
struct WideNode {
    std::vector<int32_t> child;  // >= 0: wide node index, < 0: leaf code of the BVH4 (as DevNode::child)
    std::vector<float> lo, hi;   // 3 floats per child
};
struct Wide {
    std::vector<WideNode> nodes;
    int width = 8;
};
static int32_t widen(const BvhOut& b, int32_t n4, Wide& w) {
    struct C {
        int32_t ref;
        float lo[3], hi[3];
    };
    std::vector<C> kids;
    auto push_children = [&](int32_t node) {
        const DevNode& nd = b.nodes[node];
        for (int k = 0; k < 4; k++)
            if (nd.child[k] != kNoChild)
                kids.push_back(C{nd.child[k], {nd.lo_x[k], nd.lo_y[k], nd.lo_z[k]}, {nd.hi_x[k], nd.hi_y[k], nd.hi_z[k]}});
    };
    push_children(n4);
    for (;;) {
        int best = -1;
        double best_area = -1.0;
        for (size_t i = 0; i < kids.size(); i++)
            if (kids[i].ref >= 0) {
                const DevNode& nd = b.nodes[kids[i].ref];
                int nc = 0;
                for (int k = 0; k < 4; k++) nc += nd.child[k] != kNoChild;
                if ((int)kids.size() - 1 + nc > w.width) continue;
                const double dx = (double)kids[i].hi[0] - kids[i].lo[0], dy = (double)kids[i].hi[1] - kids[i].lo[1],
                             dz = (double)kids[i].hi[2] - kids[i].lo[2], a = dx * dy + dy * dz + dz * dx;
                if (a > best_area) {
                    best_area = a;
                    best = (int)i;
                }
            }
        if (best < 0) break;
        const int32_t open = kids[best].ref;
        kids.erase(kids.begin() + best);
        push_children(open);
    }
    const int32_t me = (int32_t)w.nodes.size();
    w.nodes.emplace_back();
    for (const C& c : kids) {
        const int32_t r = c.ref >= 0 ? widen(b, c.ref, w) : c.ref;
        WideNode& wn = w.nodes[me];
        wn.child.push_back(r);
        for (int a = 0; a < 3; a++) {
            wn.lo.push_back(c.lo[a]);
            wn.hi.push_back(c.hi[a]);
        }
    }
    return me;
}

static double sah_cost(const BvhOut& b) {
    // expected fetches of a random line that hits the root box: sum over nodes of area(node box) / area(root box),
    // a node's box being the union of its child boxes; leaves add their own area (one primitive test each)
    auto area = [](const float* lo, const float* hi) {
        const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    std::vector<double> node_area(b.nodes.size(), 0.0);
    double leaf_sum = 0.0;
    auto box_of = [&](const DevNode& nd, float* lo, float* hi) {
        lo[0] = lo[1] = lo[2] = INFINITY;
        hi[0] = hi[1] = hi[2] = -INFINITY;
        for (int k = 0; k < 4; k++)
            if (nd.child[k] != kNoChild) {
                lo[0] = std::min(lo[0], nd.lo_x[k]); lo[1] = std::min(lo[1], nd.lo_y[k]); lo[2] = std::min(lo[2], nd.lo_z[k]);
                hi[0] = std::max(hi[0], nd.hi_x[k]); hi[1] = std::max(hi[1], nd.hi_y[k]); hi[2] = std::max(hi[2], nd.hi_z[k]);
            }
    };
    double total = 0.0;
    float rlo[3], rhi[3];
    box_of(b.nodes[0], rlo, rhi);
    const double root = area(rlo, rhi);
    for (const DevNode& nd : b.nodes) {
        float lo[3], hi[3];
        box_of(nd, lo, hi);
        total += area(lo, hi);
        for (int k = 0; k < 4; k++)
            if (nd.child[k] != kNoChild && nd.child[k] < 0) {
                const float l[3] = {nd.lo_x[k], nd.lo_y[k], nd.lo_z[k]}, h[3] = {nd.hi_x[k], nd.hi_y[k], nd.hi_z[k]};
                leaf_sum += area(l, h);
            }
    }
    std::printf("sah: node term %.3f leaf term %.3f (relative to the root box; the 2e4-wide floor dominates the root)\n",
                total / root, leaf_sum / root);
    return (total + leaf_sum) / root;
}

static int trace_wide(const Scene& s, const Wide& w, V o, V dir, double tmax, Count& c) {
    const double inv[3] = {1.0 / dir.x, 1.0 / dir.y, 1.0 / dir.z}, oo[3] = {o.x, o.y, o.z};
    struct E {
        int32_t node;
        double t;
    };
    E stack[512];
    int sp = 0;
    int32_t cur = 0;
    int best = -1;
    double best_t = tmax;
    c.rays++;
    for (;;) {
        if (cur >= 0) {
            const WideNode& nd = w.nodes[cur];
            c.nodes++;
            E hit[16];
            int nh = 0;
            for (size_t k = 0; k < nd.child.size(); k++) {
                double tn = 1e-3, tf = best_t;
                for (int a = 0; a < 3; a++) {
                    double t0 = (nd.lo[3 * k + a] - oo[a]) * inv[a], t1 = (nd.hi[3 * k + a] - oo[a]) * inv[a];
                    if (t0 > t1) std::swap(t0, t1);
                    if (t0 > tn) tn = t0;
                    if (t1 < tf) tf = t1;
                }
                if (!(tf <= tn)) hit[nh++] = E{nd.child[k], tn};
            }
            std::sort(hit, hit + nh, [](const E& a, const E& b) { return a.t > b.t; });
            for (int k = 0; k < nh; k++) stack[sp++] = hit[k];
        } else {
            const uint32_t code = (uint32_t)(-1 - cur) & ~kLeafCodeOther;
            const uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
            for (uint32_t i = 0; i < cnt; i++) {
                c.prims++;
                double t;
                V n;
                const uint32_t pi = s.bvh.order[first + i];
                if (hit_prim(s, pi, o, dir, best_t, t, n) && t < best_t) {
                    best_t = t;
                    best = (int)pi;
                }
            }
        }
        for (;;) {
            if (sp == 0) return best;
            const E e = stack[--sp];
            if (e.t <= best_t) {
                cur = e.node;
                break;
            }
        }
    }
}

static unsigned long long rng_state = 88172645463325252ull;
static double rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) * (1.0 / 9007199254740992.0);
}

int main(int argc, char** argv) {
    const char* preset = argc > 1 ? argv[1] : "two_dragons";
    const unsigned long long faces = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 871414ull;
    const int variant = argc > 3 ? std::atoi(argv[3]) : 0;
    const double aspect = argc > 4 ? std::atof(argv[4]) : 16.0 / 9.0;
    const int n_rays = argc > 5 ? std::atoi(argv[5]) : 200000;
    rrh_scene* hs = nullptr;
    if (rrh_scene_build(preset, aspect, faces, nullptr, variant, &hs) != 0) {
        std::fprintf(stderr, "scene: %s\n", rrh_last_error());
        return 1;
    }
    Scene s;
    s.d = rrh_scene_desc(hs);
    const rt_camera cam = *rrh_scene_camera(hs);
    const auto t0 = std::chrono::steady_clock::now();
    build_bvh(s.d->prims, s.d->n_prims, s.bvh);
    const double build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%s: %llu primitives, %zu nodes, depth %u, build %.2f s\n", preset, (unsigned long long)s.d->n_prims,
                s.bvh.nodes.size(), s.bvh.depth, build_s);
    sah_cost(s.bvh);
    V light_c{0, 0, 0};
    for (uint64_t i = 0; i < s.d->n_prims; i++)
        if (s.d->prims[i].light_index >= 0) {
            const rt_primitive& p = s.d->prims[i];
            light_c = V{0.5 * (p.bbox_min[0] + p.bbox_max[0]), 0.5 * (p.bbox_min[1] + p.bbox_max[1]),
                        0.5 * (p.bbox_min[2] + p.bbox_max[2])};
            break;
        }
    Count cp, cb, cs, cb2;
    Wide w8, w16;
    w8.width = 8;
    w16.width = 16;
    widen(s.bvh, 0, w8);
    widen(s.bvh, 0, w16);
    std::printf("re-collapsed: %zu nodes of up to 8 children, %zu of up to 16\n", w8.nodes.size(), w16.nodes.size());
    Count c8, c16;
    const V org{cam.origin[0], cam.origin[1], cam.origin[2]}, ulc{cam.upper_left_corner[0], cam.upper_left_corner[1], cam.upper_left_corner[2]};
    const V ho{cam.horizontal_offset[0], cam.horizontal_offset[1], cam.horizontal_offset[2]};
    const V vo{cam.vertical_offset[0], cam.vertical_offset[1], cam.vertical_offset[2]};
    for (int i = 0; i < n_rays; i++) {
        const V dir = ulc + ho * rnd() - vo * rnd() - org;
        double t;
        V n;
        int hit = trace(s, org, dir, 1e308, t, n, cp);
        trace_wide(s, w8, org, dir, 1e308, c8);
        trace_wide(s, w16, org, dir, 1e308, c16);
        V p = org + dir * t, wo = dir;
        for (int bounce = 0; bounce < 2 && hit >= 0; bounce++) {
            if (dot(n, wo) > 0.0) n = n * -1.0;
            // shadow ray towards the light's centre, then a cosine-distributed bounce
            double ts;
            V ns;
            const int hs_ = trace(s, p, light_c - p, 1e308, ts, ns, cs);
            trace_wide(s, w8, p, light_c - p, 1e308, c8);
            trace_wide(s, w16, p, light_c - p, 1e308, c16);
            const double r1 = rnd(), r2 = rnd(), ph = 6.283185307179586 * r1, sr = std::sqrt(r2);
            const V a = std::fabs(n.x) > 0.9 ? V{0, 1, 0} : V{1, 0, 0}, tn = norm(cross(n, a)), bn = cross(n, tn);
            const V d2 = tn * (std::cos(ph) * sr) + bn * (std::sin(ph) * sr) + n * std::sqrt(1.0 - r2);
            Count& cc = bounce == 0 ? cb : cb2;
            hit = trace(s, p, d2, 1e308, t, n, cc);
            trace_wide(s, w8, p, d2, 1e308, c8);
            trace_wide(s, w16, p, d2, 1e308, c16);
            p = p + d2 * t;
            wo = d2;
        }
    }
    auto pr = [](const char* name, const Count& c) {
        std::printf("%-14s rays %9llu  nodes/ray %7.3f  prims/ray %6.3f\n", name, c.rays, c.rays ? (double)c.nodes / c.rays : 0.0,
                    c.rays ? (double)c.prims / c.rays : 0.0);
    };
    pr("primary", cp);
    pr("shadow", cs);
    pr("bounce 1", cb);
    pr("bounce 2", cb2);
    Count all;
    for (const Count* c : {&cp, &cs, &cb, &cb2}) {
        all.rays += c->rays;
        all.nodes += c->nodes;
        all.prims += c->prims;
    }
    pr("all", all);
    pr("all, 8-wide", c8);
    pr("all, 16-wide", c16);
    rrh_scene_destroy(hs);
    return 0;
}

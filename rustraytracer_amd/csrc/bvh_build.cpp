// bvh_build.cpp -- host BVH builder for the device layout of scene_dev.h.
//
// Replaces BvhNode::new (src/hittable.rs:637-752: one primitive per leaf, median
// split on a random axis, O(N log^2 N)).  The reference's traversal is exhaustive,
// so its topology never changes a result; we are free to build a
// binned-SAH BVH2 with up to 4 primitives per leaf, child boxes stored in the
// parent, nodes in depth-first order (children of hot top levels stay adjacent in
// L2/Infinity Cache) and depth bounded for the traversal stack.
#include "bvh_build.h"

#include <algorithm>
#include <cmath>
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

struct Builder {
    const rt_primitive* prims;
    std::vector<uint32_t> order;  // primitive ids, partitioned in place
    std::vector<double> cx, cy, cz;
    std::vector<DevNode>& nodes;
    uint32_t max_depth = 0;

    Builder(const rt_primitive* p, size_t n, std::vector<DevNode>& out) : prims(p), nodes(out) {
        order.resize(n);
        cx.resize(n);
        cy.resize(n);
        cz.resize(n);
        for (size_t i = 0; i < n; i++) {
            order[i] = (uint32_t)i;
            cx[i] = 0.5 * (p[i].bbox_min[0] + p[i].bbox_max[0]);
            cy[i] = 0.5 * (p[i].bbox_min[1] + p[i].bbox_max[1]);
            cz[i] = 0.5 * (p[i].bbox_min[2] + p[i].bbox_max[2]);
        }
    }
    double centroid(uint32_t id, int axis) const { return axis == 0 ? cx[id] : (axis == 1 ? cy[id] : cz[id]); }

    Box bounds(size_t b, size_t e) const {
        Box bx;
        bx.reset();
        for (size_t i = b; i < e; i++) bx.grow(prims[order[i]].bbox_min, prims[order[i]].bbox_max);
        return bx;
    }

    // returns child ref; fills `bx` with the subtree bounds
    int32_t build(size_t b, size_t e, uint32_t depth, Box& bx) {
        max_depth = std::max(max_depth, depth);
        bx = bounds(b, e);
        const size_t n = e - b;
        if (n <= (size_t)kMaxLeafPrims) return -1 - (int32_t)(b * 8 + (n - 1));
        // centroid bounds
        double cmn[3], cmx[3];
        for (int a = 0; a < 3; a++) {
            cmn[a] = std::numeric_limits<double>::infinity();
            cmx[a] = -cmn[a];
        }
        for (size_t i = b; i < e; i++)
            for (int a = 0; a < 3; a++) {
                double c = centroid(order[i], a);
                cmn[a] = std::min(cmn[a], c);
                cmx[a] = std::max(cmx[a], c);
            }
        size_t mid = b;
        bool split_done = false;
        if (depth < 32) {
            constexpr int NB = 16;
            double best_cost = std::numeric_limits<double>::infinity();
            int best_axis = -1, best_bin = -1;
            for (int a = 0; a < 3; a++) {
                double ext = cmx[a] - cmn[a];
                if (!(ext > 0.0)) continue;
                Box bb[NB];
                size_t cnt[NB];
                for (int k = 0; k < NB; k++) {
                    bb[k].reset();
                    cnt[k] = 0;
                }
                double scale = (double)NB / ext;
                for (size_t i = b; i < e; i++) {
                    int k = (int)((centroid(order[i], a) - cmn[a]) * scale);
                    k = std::min(std::max(k, 0), NB - 1);
                    cnt[k]++;
                    bb[k].grow(prims[order[i]].bbox_min, prims[order[i]].bbox_max);
                }
                double right_area[NB];
                size_t right_cnt[NB];
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
            if (best_axis >= 0) {
                double ext = cmx[best_axis] - cmn[best_axis];
                double scale = (double)NB / ext;
                auto it = std::partition(order.begin() + b, order.begin() + e, [&](uint32_t id) {
                    int k = (int)((centroid(id, best_axis) - cmn[best_axis]) * scale);
                    k = std::min(std::max(k, 0), NB - 1);
                    return k <= best_bin;
                });
                mid = (size_t)(it - order.begin());
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
            std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e, [&](uint32_t x, uint32_t y) {
                double a_ = centroid(x, axis), b_ = centroid(y, axis);
                return a_ < b_ || (a_ == b_ && x < y);
            });
        }
        const int32_t me = (int32_t)nodes.size();
        nodes.push_back(DevNode{});
        Box lb, rb;
        const int32_t l = build(b, mid, depth + 1, lb);
        const int32_t r = build(mid, e, depth + 1, rb);
        DevNode& nd = nodes[me];
        for (int a = 0; a < 3; a++) {
            nd.lmin[a] = f_down(lb.mn[a]);
            nd.lmax[a] = f_up(lb.mx[a]);
            nd.rmin[a] = f_down(rb.mn[a]);
            nd.rmax[a] = f_up(rb.mx[a]);
        }
        nd.left = l;
        nd.right = r;
        nd.pad0 = nd.pad1 = 0;
        return me;
    }
};

}  // namespace

void build_bvh(const rt_primitive* prims, size_t n, BvhOut& out) {
    out.nodes.clear();
    out.order.clear();
    out.depth = 0;
    if (n == 0) return;
    out.nodes.reserve(n);
    Builder bd(prims, n, out.nodes);
    Box bx;
    int32_t root = bd.build(0, n, 0, bx);
    if (root < 0) {
        // the whole scene fits one leaf: wrap it so that node 0 is always internal
        DevNode nd{};
        for (int a = 0; a < 3; a++) {
            nd.lmin[a] = f_down(bx.mn[a]);
            nd.lmax[a] = f_up(bx.mx[a]);
            nd.rmin[a] = nd.rmax[a] = 0.0f;
        }
        nd.left = root;
        nd.right = kNoChild;
        out.nodes.push_back(nd);
    }
    out.order = std::move(bd.order);
    out.depth = bd.max_depth;
}

}  // namespace rtd

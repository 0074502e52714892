// ORACLE -- TEST INFRASTRUCTURE ONLY (see vecmath.h header).  Bit-level PARITY UNPINNED: the reference has no
// tests, fixtures or seedable output.  What ties this restatement to the reference:
//   tests/test_reference_png_pin.py  region statistics of the reference's own examples/cornell_statue.png
//   tests/test_lobe_tables.py        independent mpmath tables for every lobe's f / pdf / sample_f
//   tests/test_oracle_arms.py        uv-mesh hit records and triangle emitters against numpy / quadrature
//   tests/test_oracle_kat.py         hand-derived known-answer tests
//
// oracle.cpp -- CPU restatement (C++17, f64) of the reference's path-tracing hot
// path, quirks included, with the injectable counter RNG of
// include/rt_abi.h replacing the reference's two unseeded streams in program
// order.  Every function cites the reference lines it follows
// (paths relative to /root/reference/src).  Compile with -ffp-contract=off.
#include "oracle_api.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "vecmath.h"

namespace orc {

// consts.rs:30-42 (Q1: truncated PI)
static const double PI = 3.14159265358979;
static const double SMALL = 0.001;
static const double INF = 1e308;
static const double INV_PI = 1.0 / PI;

struct Ray {
    V3 o, d;
};
inline V3 ray_at(const Ray& r, double t) { return r.o + r.d * t; }  // geometry.rs:227-229

// hittable.rs:50-72, only the fields the path reads.
struct Hit {
    double t;
    V3 n;  // geometric normal, faces the ray after set_front (Q7)
    V3 p;
    bool front;
    double u, v;
    uint32_t mat;
    int32_t prim;
    V3 dpdu;
    V3 sh_n, sh_dpdu, sh_dpdv;  // Shading (hittable.rs:43-49); never flipped (Q7)
    V3 wo;                      // -ray.dir, un-normalised (Q2)
};

struct Counters {
    uint64_t paths = 0, r1 = 0, r2 = 0, r3 = 0, vertices = 0, nodes = 0, tris = 0, others = 0;
    void add(const Counters& o) {
        paths += o.paths; r1 += o.r1; r2 += o.r2; r3 += o.r3; vertices += o.vertices;
        nodes += o.nodes; tris += o.tris; others += o.others;
    }
};

// ---------------------------------------------------------------- scene copy
struct Mesh {
    std::vector<double> p, n, uv;
    std::vector<uint32_t> ind;
};

struct BvhNode {
    double bmin[3], bmax[3];
    int32_t left, right;  // children (internal) or -1
    int32_t prim;         // leaf primitive or -1
};

// distribution.rs:4-91 Distribution1D over `n` values func[0..n) (a window of the image, see Dist2D)
struct Dist1D {
    const double* func = nullptr;
    size_t n = 0;
    std::vector<double> cdf;
    double func_int = 0.0;
};
// distribution.rs:93-150 Distribution2D
struct Dist2D {
    std::vector<double> img;         // light.rs:615-626: luminance * sin(theta), (2w) x (2h)
    std::vector<Dist1D> p_cond_v;
    std::vector<double> marginal_func;
    Dist1D p_marginal;
};

struct Scene {
    std::vector<Mesh> meshes;
    std::vector<rt_primitive> prims;
    std::vector<rt_xform> xforms;
    std::vector<rt_material> mats;
    std::vector<rt_texture> texs;
    std::vector<std::vector<uint8_t>> hdr;  // texel copies of the RT_TEX_HDR textures (texs[i].rgbe points here)
    std::vector<rt_light> lights;
    int32_t env_light = -1;  // index of the Light::Infinite, if any
    Dist2D env_dist;
    std::vector<BvhNode> nodes;
    int32_t root = -1;
};

// ------------------------------------------------------------------ helpers
// util.rs:567-576
static void make_coordinate_system(V3 v1, V3& v2, V3& v3o) {
    if (std::fabs(v1.x) > std::fabs(v1.y))
        v2 = v3(-v1.z, 0.0, v1.x) * (1.0 / dm_sqrt(v1.x * v1.x + v1.z * v1.z));
    else
        v2 = v3(0.0, v1.z, -v1.y) * (1.0 / dm_sqrt(v1.y * v1.y + v1.z * v1.z));
    v3o = cross(v1, v2);
}
// util.rs:578-581
static V3 face_forward(V3 n, V3 v) { return dot(n, v) < 0.0 ? -n : n; }
// util.rs:591-593
static bool same_hemisphere(V3 v, V3 w) { return v.z * w.z > 0.0; }
// util.rs:203-206
static V3 reflect(V3 v, V3 n) {
    double scale = 2.0 * dot(v, n);
    return -v + n * scale;
}
// util.rs:376-385
static bool refract(V3 vec, V3 n, double eta, V3& out) {
    double cos_theta_i = dot(n, vec) / norm(vec);
    double sin2_theta_i = rmax(0.0, 1.0 - cos_theta_i * cos_theta_i);
    double sin2_theta_t = eta * eta * sin2_theta_i;
    if (sin2_theta_t >= 1.0) return false;
    double cos_theta_t = dm_sqrt(1.0 - sin2_theta_t);
    out = eta * (-vec) + (eta * cos_theta_i - cos_theta_t) * v3(n.x, n.y, n.z);
    return true;
}
// util.rs:79-94
static void concentric_sample_disk(double u0, double u1, double& dx, double& dy) {
    double ox = 2.0 * u0 - 1.0, oy = 2.0 * u1 - 1.0;
    if (ox == 0.0 && oy == 0.0) {
        dx = 0.0; dy = 0.0;
        return;
    }
    double theta, r;
    if (std::fabs(ox) > std::fabs(oy)) {
        r = ox;
        theta = PI / 4.0 * (oy / ox);
    } else {
        r = oy;
        theta = PI / 2.0 - PI / 4.0 * (ox / oy);
    }
    dx = r * dm_cos(theta);
    dy = r * dm_sin(theta);
}
// util.rs:127-148; r1, r2 are the two util::rand() draws.
static V3 rand_cosine_dir(double r1, double r2) {
    double u1 = 2.0 * r1 - 1.0, u2 = 2.0 * r2 - 1.0;
    if (u1 == 0.0 && u2 == 0.0) return v3(0.0, 0.0, 1.0);
    double theta, r;
    if (std::fabs(u1) > std::fabs(u2)) {
        r = u1;
        theta = PI / 4.0 * (u2 / u1);
    } else {
        r = u2;
        theta = PI / 2.0 - PI / 4.0 * (u1 / u2);
    }
    double x = r * dm_cos(theta);
    double y = r * dm_sin(theta);
    double z = dm_sqrt(rmax(0.0, 1.0 - x * x - y * y));
    return v3(x, y, z);
}

// Projective3::transform_point / transform_vector / inverse_* on an affine 3x4.
static V3 xf_point(const double* m, V3 p) {
    return v3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
static V3 xf_vector(const double* m, V3 v) {
    return v3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
              m[8] * v.x + m[9] * v.y + m[10] * v.z);
}

// ------------------------------------------------------------ HitRecord ops
// hittable.rs:75-117 HitRecord::new
static void hit_new(Hit& h, V3 p, double u, double v, V3 wo, V3 dpdu, V3 dpdv, double t, uint32_t mat) {
    V3 n = normalize(cross(dpdu, dpdv));
    h.p = p;
    h.n = n;
    h.t = t;
    h.front = false;
    h.u = u;
    h.v = v;
    h.mat = mat;
    h.dpdu = dpdu;
    h.wo = wo;
    h.sh_n = n;
    h.sh_dpdu = normalize(dpdu);
    h.sh_dpdv = normalize(dpdv);
    h.prim = 0;
}
// hittable.rs:186-189
static void set_front(Hit& h, const Ray& ray) {
    h.front = dot(ray.d, h.n) < 0.0;
    if (!h.front) h.n = -h.n;
}

// ------------------------------------------------------- primitive tests
// intersects.rs:10-175, one body for the three axis-aligned rects.
// axis: 2 = XY (k on z), 1 = XZ (k on y), 0 = YZ (k on x).
static bool rect_intersect(const Scene& sc, const rt_primitive& pr, const Ray& ray, double t0, double t1,
                           Hit& h) {
    Ray tr = ray;
    const rt_xform* xf = pr.xform_index >= 0 ? &sc.xforms[pr.xform_index] : nullptr;
    if (xf) {  // Ray::transform, geometry.rs:231-235
        tr.d = xf_vector(xf->inv, ray.d);
        tr.o = xf_point(xf->inv, ray.o);
    }
    double a0 = pr.v[0], b0 = pr.v[1], a1 = pr.v[2], b1 = pr.v[3], k = pr.v[4];
    double t, a, b;
    V3 dpdu, dpdv;
    if (pr.kind == RT_PRIM_XY_RECT) {
        t = (k - tr.o.z) / tr.d.z;
        if (t < t0 || t > t1) return false;
        a = tr.o.x + t * tr.d.x;
        b = tr.o.y + t * tr.d.y;
        dpdu = v3(1, 0, 0);
        dpdv = v3(0, 1, 0);
    } else if (pr.kind == RT_PRIM_XZ_RECT) {
        t = (k - tr.o.y) / tr.d.y;
        if (t < t0 || t > t1) return false;
        a = tr.o.x + t * tr.d.x;
        b = tr.o.z + t * tr.d.z;
        dpdu = v3(1, 0, 0);
        dpdv = v3(0, 0, 1);
    } else {
        t = (k - tr.o.x) / tr.d.x;
        if (t < t0 || t > t1) return false;
        a = tr.o.y + t * tr.d.y;
        b = tr.o.z + t * tr.d.z;
        dpdu = v3(0, 1, 0);
        dpdv = v3(0, 0, 1);
    }
    if (a < a0 || b < b0 || a > a1 || b > b1) return false;
    double u = (a - a0) / (a1 - a0), v = (b - b0) / (b1 - b0);
    V3 p = ray_at(tr, t);
    if (xf) {
        p = xf_point(xf->fwd, p);
        dpdu = xf_vector(xf->fwd, dpdu);
        dpdv = xf_vector(xf->fwd, dpdv);
    }
    hit_new(h, p, u, v, -ray.d, dpdu, dpdv, t, pr.mat_index);
    set_front(h, ray);
    return true;
}

// intersects.rs:177-258
static bool sphere_intersect(const rt_primitive& pr, const Ray& ray, double tmin, double tmax, Hit& h) {
    V3 center = v3(pr.v[0], pr.v[1], pr.v[2]);
    double r = pr.v[3];
    V3 diff = ray.o - center;
    double a = dot(ray.d, ray.d);
    double b = dot(diff, ray.d);
    double c = dot(diff, diff) - r * r;
    double disc = b * b - a * c;
    if (disc < 0.0) return false;
    double inv_a = 1.0 / a;
    double root = dm_sqrt(disc);
    double ans = (-b - root) * inv_a;
    double t;
    if (ans < tmax && ans > tmin) {
        t = ans;
    } else {
        ans = (-b + root) * inv_a;
        if (ans < tmax && ans > tmin)
            t = ans;
        else
            return false;
    }
    Ray tr{ray.o - center, ray.d};
    // make_sphere_record, intersects.rs:216-258
    V3 p = ray_at(tr, t);
    p = p * r / norm(p);
    if (p.x == 0.0 && p.y == 0.0) p.x = 1e-5 * r;
    double phi = dm_atan2(p.y, p.x);
    if (phi < 0.0) phi = phi + 2.0 * PI;
    double phi_max = 2.0 * PI;
    double theta_min = 0.0, theta_max = PI;
    double u = phi / phi_max;
    double theta = dm_acos(clampd(p.z / r, -1.0, 1.0));
    double v = (theta - theta_min) / (theta_max - theta_min);
    double z_r = dm_sqrt(p.x * p.x + p.y * p.y);
    double inv_z_r = 1.0 / z_r;
    double cos_phi = p.x * inv_z_r;
    double sin_phi = p.y * inv_z_r;
    V3 dpdu = v3(-phi_max * p.y, phi_max * p.x, 0.0);
    V3 dpdv = (theta_max - theta_min) * v3(p.z * cos_phi, p.z * sin_phi, -r * dm_sin(theta));
    hit_new(h, p, u, v, -tr.d, dpdu, dpdv, t, pr.mat_index);
    set_front(h, tr);
    h.p = h.p + center;
    return true;
}

// hittable.rs:292-452 Mesh::intersects_triangle (Q4, Q6).  tmin is ignored.
static bool triangle_intersect(const Scene& sc, const rt_primitive& pr, const Ray& ray, double tmax, Hit& h) {
    const Mesh& m = sc.meshes[pr.mesh_index];
    uint32_t i1 = m.ind[pr.tri_ind], i2 = m.ind[pr.tri_ind + 1], i3 = m.ind[pr.tri_ind + 2];
    V3 p0 = v3(m.p[3 * i1], m.p[3 * i1 + 1], m.p[3 * i1 + 2]);
    V3 p1 = v3(m.p[3 * i2], m.p[3 * i2 + 1], m.p[3 * i2 + 2]);
    V3 p2 = v3(m.p[3 * i3], m.p[3 * i3 + 1], m.p[3 * i3 + 2]);
    V3 dir = ray.d;
    V3 p0t = p0 - ray.o, p1t = p1 - ray.o, p2t = p2 - ray.o;
    // dir.abs().argmax(): first maximum (strict >)
    double ax = std::fabs(dir.x), ay = std::fabs(dir.y), az = std::fabs(dir.z);
    int kz = 0;
    double best = ax;
    if (ay > best) { best = ay; kz = 1; }
    if (az > best) { best = az; kz = 2; }
    int kx = (kz + 1) % 3, ky = (kx + 1) % 3;
    V3 d = v3(dir[kx], dir[ky], dir[kz]);
    p0t = v3(p0t[kx], p0t[ky], p0t[kz]);
    p1t = v3(p1t[kx], p1t[ky], p1t[kz]);
    p2t = v3(p2t[kx], p2t[ky], p2t[kz]);
    double s_x = -d.x / d.z, s_y = -d.y / d.z, s_z = 1.0 / d.z;
    p0t.x += s_x * p0t.z; p0t.y += s_y * p0t.z;
    p1t.x += s_x * p1t.z; p1t.y += s_y * p1t.z;
    p2t.x += s_x * p2t.z; p2t.y += s_y * p2t.z;
    double e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    double e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    double e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if ((e0 < 0.0 || e1 < 0.0 || e2 < 0.0) && (e0 > 0.0 || e1 > 0.0 || e2 > 0.0)) return false;
    double det = e0 + e1 + e2;
    if (std::fabs(det) < SMALL / 10000.0) return false;
    p0t.z *= s_z; p1t.z *= s_z; p2t.z *= s_z;
    double t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0.0 && (t_scaled >= 0.0 || t_scaled < tmax * det))
        return false;
    else if (det > 0.0 && (t_scaled <= 0.0 || t_scaled > tmax * det))
        return false;
    double inv_det = 1.0 / det;
    double b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det;
    double t = t_scaled * inv_det;
    if (t < SMALL / 10.0) return false;
    // get_uv, hittable.rs:454-468
    double uv0[2] = {0, 0}, uv1[2] = {1, 0}, uv2[2] = {1, 1};
    if (!m.uv.empty()) {
        uv0[0] = m.uv[2 * i1]; uv0[1] = m.uv[2 * i1 + 1];
        uv1[0] = m.uv[2 * i2]; uv1[1] = m.uv[2 * i2 + 1];
        uv2[0] = m.uv[2 * i3]; uv2[1] = m.uv[2 * i3 + 1];
    }
    double duv02x = uv0[0] - uv2[0], duv02y = uv0[1] - uv2[1];
    double duv12x = uv1[0] - uv2[0], duv12y = uv1[1] - uv2[1];
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    double determinant = duv02x * duv12y - duv02y * duv12x;
    V3 dpdu, dpdv;
    if (std::fabs(determinant) < SMALL / 10000.0) {
        V3 n = cross(p2 - p0, p1 - p0);
        if (norm2(n) == 0.0) return false;
        make_coordinate_system(n, dpdu, dpdv);
    } else {
        double invd = 1.0 / determinant;
        dpdu = (duv12y * dp02 - duv02y * dp12) * invd;
        dpdv = (-duv12x * dp02 + duv02x * dp12) * invd;
    }
    V3 p_hit = b0 * p0 + b1 * p1 + b2 * p2;
    double u_hit = b0 * uv0[0] + b1 * uv1[0] + b2 * uv2[0];
    double v_hit = b0 * uv0[1] + b1 * uv1[1] + b2 * uv2[1];
    V3 normal;
    if (m.n.empty()) {
        normal = cross(dp02, dp12);
    } else {
        V3 n1 = v3(m.n[3 * i1], m.n[3 * i1 + 1], m.n[3 * i1 + 2]);
        V3 n2 = v3(m.n[3 * i2], m.n[3 * i2 + 1], m.n[3 * i2 + 2]);
        V3 n3 = v3(m.n[3 * i3], m.n[3 * i3 + 1], m.n[3 * i3 + 2]);
        normal = b0 * n1 + b1 * n2 + b2 * n3;
    }
    hit_new(h, p_hit, u_hit, v_hit, -ray.d, dpdu, dpdv, t, pr.mat_index);
    h.n = normalize(cross(dp02, dp12));
    h.sh_n = normalize(normal);
    V3 ss = normalize(dpdu);
    V3 ts = normalize(cross(h.sh_n, ss));
    if (norm2(ts) > 0.0) {
        ss = normalize(cross(ts, h.sh_n));
    } else {
        V3 a, b;
        make_coordinate_system(h.sh_n, a, b);
        ss = normalize(a);
        ts = normalize(b);
    }
    // set_shading_geometry(ss, ts, .., is_auth = true), hittable.rs:191-210
    V3 n = normalize(cross(ss, ts));
    h.sh_n = n;
    h.n = face_forward(h.n, h.sh_n);
    h.sh_dpdu = ss;
    h.sh_dpdv = ts;
    set_front(h, ray);
    h.u = u_hit;
    h.v = v_hit;
    return true;
}

// primitive.rs:372-425 intersects_obj (no prim_index stamp, FlipFace transparent)
static bool intersects_obj(const Scene& sc, const rt_primitive& pr, const Ray& ray, double tmin, double tmax,
                           Hit& h, Counters* c) {
    switch (pr.kind) {
        case RT_PRIM_SPHERE:
            if (c) c->others++;
            return sphere_intersect(pr, ray, tmin, tmax, h);
        case RT_PRIM_TRIANGLE:
            if (c) c->tris++;
            return triangle_intersect(sc, pr, ray, tmax, h);
        default:
            if (c) c->others++;
            return rect_intersect(sc, pr, ray, tmin, tmax, h);
    }
}
// primitive.rs:247-316 Primitive::intersects (FlipFace flips `front` only; stamps prim_index)
static bool prim_intersects(const Scene& sc, int32_t index, const Ray& ray, double tmin, double tmax, Hit& h,
                            Counters* c) {
    const rt_primitive& pr = sc.prims[index];
    if (!intersects_obj(sc, pr, ray, tmin, tmax, h, c)) return false;
    if (pr.flip) h.front = !h.front;
    h.prim = index;
    return true;
}

// hittable.rs:494-508 BoundingBox::intersects (Q5)
static bool box_intersects(const double* bmin, const double* bmax, const Ray& ray, double tmin, double tmax) {
    for (int a = 0; a < 3; a++) {
        double inv_d = 1.0 / ray.d[a];
        double val1 = (bmin[a] - ray.o[a]) * inv_d;
        double val2 = (bmax[a] - ray.o[a]) * inv_d;
        double t0 = rmin(val1, val2);
        double t1 = rmax(val1, val2);
        tmin = rmax(tmin, t0);
        tmax = rmin(tmax, t1);
        if (tmax <= tmin) return false;
    }
    return true;
}

// ------------------------------------------------------------------- BVH
// hittable.rs:637-752 BvhNode::new: one primitive per leaf, median split after a
// sort on bbox.min[axis].  The reference draws the axis from entropy per node
// (hittable.rs:645-652); here it is a hash of (start,end) so that builds repeat.
static int32_t bvh_build(Scene& sc, std::vector<int32_t>& idx, size_t start, size_t end) {
    auto axis_of = [](size_t s, size_t e) { return (int)(rng_mix(s * 0x9E3779B97F4A7C15ull + e) % 3); };
    auto leaf = [&](int32_t prim) {
        BvhNode n;
        for (int a = 0; a < 3; a++) {
            n.bmin[a] = sc.prims[prim].bbox_min[a];
            n.bmax[a] = sc.prims[prim].bbox_max[a];
        }
        n.left = n.right = -1;
        n.prim = prim;
        sc.nodes.push_back(n);
        return (int32_t)sc.nodes.size() - 1;
    };
    size_t num = end - start;
    int axis = axis_of(start, end);
    int32_t l, r;
    if (num == 1) return leaf(idx[start]);
    if (num == 2) {
        // handle_two (hittable.rs:725-752): left = `next`, right = `curr`
        bool less = sc.prims[idx[start]].bbox_min[axis] < sc.prims[idx[start + 1]].bbox_min[axis];
        size_t curr = less ? start : start + 1, next = less ? start + 1 : start;
        l = leaf(idx[next]);
        r = leaf(idx[curr]);
    } else {
        std::stable_sort(idx.begin() + start, idx.begin() + end, [&](int32_t a, int32_t b) {
            return sc.prims[a].bbox_min[axis] < sc.prims[b].bbox_min[axis];
        });
        size_t mid = start + num / 2;
        l = bvh_build(sc, idx, start, mid);
        r = bvh_build(sc, idx, mid, end);
    }
    BvhNode n;
    for (int a = 0; a < 3; a++) {  // BoundingBox::union, hittable.rs:551-563
        n.bmin[a] = rmin(sc.nodes[l].bmin[a], sc.nodes[r].bmin[a]);
        n.bmax[a] = rmax(sc.nodes[l].bmax[a], sc.nodes[r].bmax[a]);
    }
    n.left = l;
    n.right = r;
    n.prim = -1;
    sc.nodes.push_back(n);
    return (int32_t)sc.nodes.size() - 1;
}

// Tie rule of the ABI: equal t -> larger prim index (reference: later subtree, random).
static inline bool better(const Hit& a, const Hit& b) { return a.t < b.t || (a.t == b.t && a.prim > b.prim); }

// hittable.rs:591-634 BvhNode::intersects, reference-shaped: no tmax shrinking,
// both children always visited, full records moved up the recursion.
static bool bvh_exhaustive(const Scene& sc, int32_t ni, const Ray& ray, double tmin, double tmax, Hit& out,
                           Counters* c) {
    const BvhNode& n = sc.nodes[ni];
    if (c) c->nodes++;
    if (!box_intersects(n.bmin, n.bmax, ray, tmin, tmax)) return false;
    if (n.prim >= 0) return prim_intersects(sc, n.prim, ray, tmin, tmax, out, c);
    Hit l, r;
    bool hl = bvh_exhaustive(sc, n.left, ray, tmin, tmax, l, c);
    bool hr = bvh_exhaustive(sc, n.right, ray, tmin, tmax, r, c);
    if (hl && hr) {
        out = better(l, r) ? l : r;
        return true;
    }
    if (hl) { out = l; return true; }
    if (hr) { out = r; return true; }
    return false;
}

// Same answer with pruning: a subtree is skipped when its box fails the
// reference's own slab test against [tmin, max(best_t, tmin)*(1+1e-9)] -- by monotonicity
// of the slab arithmetic every ancestor box passes whenever the leaf box does.
static bool bvh_ordered(const Scene& sc, const Ray& ray, double tmin, double tmax, Hit& out, Counters* c) {
    if (sc.root < 0) return false;
    int32_t stack[128];
    int sp = 0;
    stack[sp++] = sc.root;
    bool found = false;
    Hit tmp;
    while (sp > 0) {
        const BvhNode& n = sc.nodes[stack[--sp]];
        if (c) c->nodes++;
        // a triangle hit may lie below tmin (hittable.rs:360 accepts t >= 1e-4 whatever tmin is): the
        // pruning interval [tmin, lim] must stay non-empty or a still closer such hit would be culled
        double lim = found ? rmax(out.t * (1.0 + 1e-9), tmin * (1.0 + 1e-9) + 1e-300) : tmax;
        if (!box_intersects(n.bmin, n.bmax, ray, tmin, lim)) continue;
        if (n.prim >= 0) {
            // leaf: the reference tests the leaf box against the ORIGINAL interval
            if (!box_intersects(n.bmin, n.bmax, ray, tmin, tmax)) continue;
            if (prim_intersects(sc, n.prim, ray, tmin, tmax, tmp, c)) {
                if (!found || better(tmp, out)) out = tmp;
                found = true;
            }
        } else {
            stack[sp++] = n.left;
            stack[sp++] = n.right;
        }
    }
    return found;
}

static bool brute_force(const Scene& sc, const Ray& ray, double tmin, double tmax, Hit& out, Counters* c) {
    bool found = false;
    Hit tmp;
    for (int32_t i = 0; i < (int32_t)sc.prims.size(); i++) {
        if (!box_intersects(sc.prims[i].bbox_min, sc.prims[i].bbox_max, ray, tmin, tmax)) continue;
        if (prim_intersects(sc, i, ray, tmin, tmax, tmp, c)) {
            if (!found || better(tmp, out)) out = tmp;
            found = true;
        }
    }
    return found;
}

// optional per-thread log of every root closest-hit query (oracle_sample_rays)
struct RayLog {
    std::vector<rt_ray> rays;
    std::vector<rt_hit> hits;
};
static thread_local RayLog* g_ray_log = nullptr;

static bool closest_hit_impl(const Scene& sc, int mode, const Ray& ray, double tmin, double tmax, Hit& out,
                             Counters* c) {
    // ABI rule (include/rt_abi.h): a ray with a NaN component misses.  In the reference such a ray passes
    // every slab test (f64::min/max drop NaN) and "hits" whichever triangle its random tree visits first.
    if (ray.o.x != ray.o.x || ray.o.y != ray.o.y || ray.o.z != ray.o.z || ray.d.x != ray.d.x || ray.d.y != ray.d.y ||
        ray.d.z != ray.d.z)
        return false;
    if (mode == ORACLE_TRAVERSAL_EXHAUSTIVE) {
        if (sc.root < 0) return false;
        return bvh_exhaustive(sc, sc.root, ray, tmin, tmax, out, c);
    }
    if (mode == ORACLE_TRAVERSAL_BRUTE) return brute_force(sc, ray, tmin, tmax, out, c);
    return bvh_ordered(sc, ray, tmin, tmax, out, c);
}
static bool closest_hit(const Scene& sc, int mode, const Ray& ray, double tmin, double tmax, Hit& out,
                        Counters* c) {
    const bool found = closest_hit_impl(sc, mode, ray, tmin, tmax, out, c);
    if (g_ray_log) {
        rt_ray r;
        r.origin[0] = ray.o.x; r.origin[1] = ray.o.y; r.origin[2] = ray.o.z;
        r.dir[0] = ray.d.x; r.dir[1] = ray.d.y; r.dir[2] = ray.d.z;
        r.tmin = tmin;
        r.tmax = tmax;
        rt_hit h;
        h.t = found ? out.t : RT_INFINITY;
        h.prim = found ? out.prim : -1;
        h.reserved = 0;
        g_ray_log->rays.push_back(r);
        g_ray_log->hits.push_back(h);
    }
    return found;
}

// ------------------------------------------------------------- textures
// material.rs:542-565 (Q19).  The recursion follows Checkered -> even/odd ids.
// Rust `f64 as u32`: saturating, NaN -> 0
static uint32_t sat_u32(double x) {
    if (!(x > 0.0)) return 0u;
    if (x >= 4294967295.0) return 4294967295u;
    return (uint32_t)x;
}
static size_t sat_usize(double x) {
    if (!(x > 0.0)) return 0;
    if (x >= 18446744073709551615.0) return (size_t)-1;
    return (size_t)x;
}
// material.rs:570-587 Texture::Hdr arm of get_value; texels arrive as to_rgbe8(data[i]) (rt_abi.h)
static V3 hdr_value(const rt_texture& t, double u, double v) {
    const uint32_t width = t.width, height = t.height;
    uint32_t x = sat_u32(std::round((1.0 - u) * (double)width));
    uint32_t y = sat_u32(std::round(v * (double)height));
    x = x % width;
    y = y % height;
    const uint8_t* px = t.rgbe + 4 * ((size_t)y * width + x);
    const double sc = std::ldexp(1.0, (int)px[3] - 128);  // (2f64).powi(e - 128)
    return v3(((double)px[0] + 0.5) * sc / 256.0, ((double)px[1] + 0.5) * sc / 256.0, ((double)px[2] + 0.5) * sc / 256.0);
}
static V3 texture_value(const Scene& sc, uint32_t index, double u, double v, int depth = 0) {
    const rt_texture& t = sc.texs[index];
    if (t.kind == RT_TEX_HDR) return hdr_value(t, u, v);
    if (t.kind == RT_TEX_CHECKERED && depth < 8) {
        double mult = dm_sin(t.frequency * u * 2.0 * PI) * dm_sin(t.frequency * v * 2.0 * PI);
        if (mult < 0.0) return texture_value(sc, t.even, u, v, depth + 1);
        return texture_value(sc, t.odd, u, v, depth + 1);
    }
    return v3(t.color[0], t.color[1], t.color[2]);
}

// ------------------------------------------------------------ Fresnel
// bxdf.rs:113-136
static double fr_dielectric(double cos_theta_i, double eta_i, double eta_t) {
    cos_theta_i = clampd(cos_theta_i, -1.0, 1.0);
    double index_i = eta_i, index_t = eta_t;
    if (cos_theta_i < 0.0) {
        index_i = eta_t;
        index_t = eta_i;
        cos_theta_i = std::fabs(cos_theta_i);
    }
    double sin_theta_i = dm_sqrt(rmax(0.0, 1.0 - cos_theta_i * cos_theta_i));
    double sin_theta_t = index_i / index_t * sin_theta_i;
    double cos_theta_t = dm_sqrt(rmax(0.0, 1.0 - sin_theta_t * sin_theta_t));
    if (sin_theta_t >= 1.0) return 1.0;
    double r_parl = ((index_t * cos_theta_i) - (index_i * cos_theta_t)) /
                    ((index_t * cos_theta_i) + (index_i * cos_theta_t));
    double r_perp = ((index_i * cos_theta_i) - (index_t * cos_theta_t)) /
                    ((index_i * cos_theta_i) + (index_t * cos_theta_t));
    return (r_parl * r_parl + r_perp * r_perp) / 2.0;
}
// bxdf.rs:141-170
static V3 fr_conductor(double cos_theta_i, V3 eta, V3 eta_k) {
    cos_theta_i = clampd(cos_theta_i, -1.0, 1.0);
    double c2 = cos_theta_i * cos_theta_i;
    double s2 = 1.0 - c2;
    V3 eta2 = cmul(eta, eta);
    V3 etak2 = cmul(eta_k, eta_k);
    V3 t0 = (eta2 - etak2) - v3(s2, s2, s2);
    V3 a2pb2 = cmul(t0, t0) + cmul(eta2, etak2) * 4.0;
    a2pb2 = v3(dm_sqrt(a2pb2.x), dm_sqrt(a2pb2.y), dm_sqrt(a2pb2.z));
    V3 t1 = a2pb2 + v3(c2, c2, c2);
    V3 a = (a2pb2 + t0) * 0.5;
    a = v3(dm_sqrt(a.x), dm_sqrt(a.y), dm_sqrt(a.z));
    V3 t2 = a * (2.0 * cos_theta_i);
    V3 rs = cdiv(t1 - t2, t1 + t2);
    V3 t3 = a2pb2 * c2 + v3(s2 * s2, s2 * s2, s2 * s2);
    V3 t4 = t2 * s2;
    V3 rp = cmul(rs, cdiv(t3 - t4, t3 + t4));
    return (rp + rs) * 0.5;
}

// ------------------------------------------------------ BxDF / BSDF
enum LobeKind { LOBE_LAMBERT = 0, LOBE_MICROFACET = 1, LOBE_FRESNEL_SPECULAR = 2, LOBE_SPECULAR_REFL = 3, LOBE_MICRO_TRANS = 4 };
enum FresnelKind { FR_DIELECTRIC = 0, FR_CONDUCTOR = 1, FR_NOOP = 2 };

struct Lobe {
    int kind;
    uint8_t type;  // BSDF_* flags
    V3 color;      // Lambert colour / microfacet colour / FresnelSpecular r
    V3 t;          // FresnelSpecular t
    int fresnel;
    double eta_i, eta_t;  // FresnelDielectric
    V3 eta, k;            // FresnelConductor
    double alpha_x, alpha_y;
    double eta_a, eta_b;  // FresnelSpecular, MicrofacetTransmission
};

struct Bsdf {  // bsdf.rs:13-20
    V3 ns, ng, ss, ts;
    int n = 0;
    Lobe lobes[2];
};

// bxdf.rs:190-211
static V3 fresnel_evaluate(const Lobe& l, double cos_theta_i) {
    if (l.fresnel == FR_DIELECTRIC) {
        double v = fr_dielectric(std::fabs(cos_theta_i), l.eta_t, l.eta_i);
        return v3(v, v, v);
    }
    if (l.fresnel == FR_CONDUCTOR) return fr_conductor(std::fabs(cos_theta_i), l.eta, l.k);
    return white();
}

// bxdf.rs:12-56
static double cos_sq_theta(V3 v) { return v.z * v.z; }
static double sin_sq_theta(V3 v) { return rmax(0.0, 1.0 - cos_sq_theta(v)); }
static double sin_theta(V3 v) { return dm_sqrt(sin_sq_theta(v)); }
static double tan_theta(V3 v) { return sin_theta(v) / v.z; }
static double tan_sq_theta(V3 v) { return sin_sq_theta(v) / cos_sq_theta(v); }
static double cos_phi(V3 v) {
    double s = sin_theta(v);
    return s == 0.0 ? 1.0 : clampd(v.x / s, -1.0, 1.0);
}
static double sin_phi(V3 v) {
    double s = sin_theta(v);
    return s == 0.0 ? 1.0 : clampd(v.y / s, -1.0, 1.0);
}
static double cos_sq_phi(V3 v) { return cos_phi(v) * cos_phi(v); }
static double sin_sq_phi(V3 v) { return sin_phi(v) * sin_phi(v); }

// microfacet.rs:53-68 TrowbridgeReitz d
static double tr_d(double ax, double ay, V3 wh) {
    double t2 = tan_sq_theta(wh);
    if (t2 == INFINITY) return 0.0;
    double cos4 = cos_sq_theta(wh) * cos_sq_theta(wh);
    double e = (cos_sq_phi(wh) / (ax * ax) + sin_sq_phi(wh) / (ay * ay)) * t2;
    return 1.0 / (PI * ax * ay * cos4 * (1.0 + e) * (1.0 + e));
}
// microfacet.rs:109-123
static double tr_lambda(double ax, double ay, V3 w) {
    double abs_tan = std::fabs(tan_theta(w));
    if (abs_tan == INFINITY) return 0.0;
    double alpha = dm_sqrt(cos_sq_phi(w) * ax * ax + sin_sq_phi(w) * ay * ay);
    double a2t2 = (alpha * abs_tan) * (alpha * abs_tan);
    return (-1.0 + dm_sqrt(1.0 + a2t2)) / 2.0;
}
// microfacet.rs:143-157
static double tr_g1(double ax, double ay, V3 w) { return 1.0 / (1.0 + tr_lambda(ax, ay, w)); }
static double tr_g(double ax, double ay, V3 wo, V3 wi) {
    return 1.0 / (1.0 + tr_lambda(ax, ay, wo) + tr_lambda(ax, ay, wi));
}
// microfacet.rs:163-172 (sample_visible_area = true in every preset)
static double tr_pdf(double ax, double ay, V3 wo, V3 wh) {
    return tr_d(ax, ay, wh) * tr_g1(ax, ay, wo) * std::fabs(dot(wo, wh)) / std::fabs(wo.z);
}
// microfacet.rs:470-512
static void tr_sample_11(double cos_theta, double u1, double u2, double& sx, double& sy) {
    if (cos_theta > 0.9999) {
        double r = dm_sqrt(u1 / (1.0 - u1));
        double phi = 2.0 * PI * u2;
        sx = r * dm_cos(phi);
        sy = r * dm_sin(phi);
        return;
    }
    double sin_t = rmax(0.0, dm_sqrt(1.0 - cos_theta * cos_theta));
    double tan_t = sin_t / cos_theta;
    double a = 1.0 / tan_t;
    double g1 = 2.0 / (1.0 + dm_sqrt(1.0 + 1.0 / (a * a)));
    a = 2.0 * u1 / g1 - 1.0;
    double tmp = 1.0 / (a * a - 1.0);
    if (tmp > 1e10) tmp = 1e10;
    double b = tan_t;
    double d = dm_sqrt(rmax(0.0, b * b * tmp * tmp - (a * a - b * b) * tmp));
    double slope_x1 = b * tmp - d;
    double slope_x2 = b * tmp + d;
    sx = (a < 0.0 || slope_x2 > 1.0 / tan_t) ? slope_x1 : slope_x2;
    double s, nu2;
    if (u2 > 0.5) {
        s = 1.0;
        nu2 = 2.0 * (u2 - 0.5);
    } else {
        s = -1.0;
        nu2 = 2.0 * (0.5 - u2);
    }
    double z = (nu2 * (nu2 * (nu2 * 0.27385 - 0.73369) + 0.46341)) /
               (nu2 * (nu2 * (nu2 * 0.093073 + 0.309420) - 1.0) + 0.597999);
    sy = s * z * dm_sqrt(1.0 + sx * sx);
}
// microfacet.rs:448-468
static V3 tr_sample(V3 wi, double ax, double ay, double u1, double u2) {
    V3 wi_s = normalize(v3(ax * wi.x, ay * wi.y, wi.z));
    double sx, sy;
    tr_sample_11(wi_s.z, u1, u2, sx, sy);
    double sp = sin_phi(wi_s), cp = cos_phi(wi_s);
    double tmp = cp * sx - sp * sy;
    sy = sp * sx + cp * sy;
    sx = tmp;
    sx = ax * sx;
    sy = ay * sy;
    return normalize(v3(-sx, -sy, 1.0));
}
// microfacet.rs:240-282 (visible-area branch)
static V3 tr_sample_wh(double ax, double ay, V3 wo, double u0, double u1) {
    bool flip = wo.z < 0.0;
    V3 wh = tr_sample(flip ? -wo : wo, ax, ay, u0, u1);
    return flip ? -wh : wh;
}
// microfacet.rs:442-446
static double tr_roughness_to_alpha(double roughness) {
    roughness = rmax(roughness, 1e-5);
    double x = dm_log(roughness);
    return 1.62142 + 0.819955 * x + 0.1734 * x * x + 0.0171201 * x * x * x + 0.000640711 * x * x * x * x;
}

static bool matches_flags(uint8_t flag, uint8_t other) { return (flag & other) == flag; }  // bxdf.rs:66-68

// bxdf.rs:328-337, 366-392, 464
static V3 bxdf_f(const Lobe& l, V3 wo, V3 wi) {
    switch (l.kind) {
        case LOBE_LAMBERT: return l.color * INV_PI;
        case LOBE_MICROFACET: {
            double cos_o = std::fabs(wo.z), cos_i = std::fabs(wi.z);
            V3 wh = wi + wo;
            if (cos_i == 0.0 || cos_o == 0.0) return black();
            if (is_black(wh)) return black();
            wh = normalize(wh);
            V3 f = fresnel_evaluate(l, dot(wi, face_forward(wh, v3(0, 0, 1))));
            V3 comp1 = l.color * tr_d(l.alpha_x, l.alpha_y, wh) * tr_g(l.alpha_x, l.alpha_y, wo, wi);
            return cmul(comp1, f * (1.0 / (4.0 * cos_i * cos_o)));
        }
        case LOBE_MICRO_TRANS: {  // bxdf.rs:393-441 (mode == RADIANCE)
            if (same_hemisphere(wo, wi)) return black();
            double cos_theta_o = wo.z, cos_theta_i = wi.z;
            if (cos_theta_i == 0.0 || cos_theta_o == 0.0) return black();
            double eta = wo.z > 0.0 ? l.eta_b / l.eta_a : l.eta_a / l.eta_b;
            V3 wh = normalize(wo + wi * eta);
            if (wh.z < 0.0) wh = -wh;
            if (dot(wo, wh) * dot(wi, wh) > 0.0) return black();
            V3 f = fresnel_evaluate(l, dot(wo, wh));
            double sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
            double factor = 1.0 / eta;
            V3 c = cmul(white() - f, l.color);
            return c * std::fabs(tr_d(l.alpha_x, l.alpha_y, wh) * tr_g(l.alpha_x, l.alpha_y, wo, wi) * eta * eta *
                                 std::fabs(dot(wi, wh)) * std::fabs(dot(wo, wh)) * factor * factor /
                                 (cos_theta_i * cos_theta_o * sqrt_denom * sqrt_denom));
        }
        default: return black();
    }
}
// bxdf.rs:721-741, 777-791, 829-835
static double bxdf_pdf(const Lobe& l, V3 wo, V3 wi) {
    switch (l.kind) {
        case LOBE_LAMBERT:
        case LOBE_SPECULAR_REFL: return same_hemisphere(wo, wi) ? std::fabs(wi.z) * INV_PI : 0.0;
        case LOBE_MICROFACET: {
            if (!same_hemisphere(wo, wi)) return 0.0;
            V3 wh = normalize(wo + wi);
            return tr_pdf(l.alpha_x, l.alpha_y, wo, wh) / (4.0 * dot(wo, wh));
        }
        case LOBE_MICRO_TRANS: {  // bxdf.rs:742-763
            if (same_hemisphere(wo, wi)) return 0.0;
            double eta = wo.z > 0.0 ? l.eta_b / l.eta_a : l.eta_a / l.eta_b;
            V3 wh = normalize(wo + wi * eta);
            if (dot(wo, wh) * dot(wi, wh) > 0.0) return 0.0;
            double sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
            double dwh_dwi = std::fabs(eta * eta * dot(wi, wh)) / (sqrt_denom * sqrt_denom);
            return tr_pdf(l.alpha_x, l.alpha_y, wo, wh) * dwh_dwi;
        }
        default: return 0.0;
    }
}
// bxdf.rs:532-607, 640-684, 815-827.  `rng` supplies default_sample_f's entropy draws.
static void bxdf_sample_f(const Lobe& l, V3 wo, double u0, double u1, Rng& rng, V3& f, V3& wi, double& pdf) {
    switch (l.kind) {
        case LOBE_LAMBERT: {  // default_sample_f: ignores (u0,u1), draws 2
            double r1 = rng.next();
            double r2 = rng.next();
            wi = rand_cosine_dir(r1, r2);
            if (wo.z < 0.0) wi.z *= -1.0;
            pdf = bxdf_pdf(l, wo, wi);
            f = bxdf_f(l, wo, wi);
            return;
        }
        case LOBE_MICROFACET: {
            if (wo.z == 0.0) { f = black(); wi = black(); pdf = 0.0; return; }
            V3 wh = tr_sample_wh(l.alpha_x, l.alpha_y, wo, u0, u1);
            // Q14: the wo.wh < 0 early-out has no `return` (bxdf.rs:598-600)
            wi = reflect(wo, wh);
            if (!same_hemisphere(wo, wi)) { f = black(); wi = black(); pdf = 0.0; return; }
            pdf = tr_pdf(l.alpha_x, l.alpha_y, wo, wh) / (4.0 * dot(wo, wh));
            f = bxdf_f(l, wo, wi);
            return;
        }
        case LOBE_FRESNEL_SPECULAR: {  // Q15
            double fr = fr_dielectric(wo.z / norm(wo), l.eta_a, l.eta_b);
            if (u0 < fr) {
                wi = v3(-wo.x, -wo.y, wo.z);
                pdf = fr;
                f = l.color * fr;
                return;
            }
            bool entering = wo.z > 0.0;
            double eta_i = entering ? l.eta_a : l.eta_b;
            double eta_t = entering ? l.eta_b : l.eta_a;
            V3 dirv;
            if (refract(wo, face_forward(v3(0, 0, 1), wo), eta_i / eta_t, dirv)) {
                V3 ft = l.t * (1.0 - fr);
                ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));  // mode == RADIANCE
                f = ft;
                wi = dirv;
                pdf = 1.0 - fr;
                return;
            }
            f = black(); wi = black(); pdf = 0.0;
            return;
        }
        case LOBE_MICRO_TRANS: {  // bxdf.rs:608-638
            if (wo.z == 0.0) { f = black(); wi = black(); pdf = 0.0; return; }
            V3 wh = tr_sample_wh(l.alpha_x, l.alpha_y, wo, u0, u1);
            if (dot(wo, wh) < 0.0) { f = black(); wi = black(); pdf = 0.0; return; }
            double eta = wo.z > 0.0 ? l.eta_a / l.eta_b : l.eta_b / l.eta_a;
            V3 t;
            if (!refract(wo, wh, eta, t)) { f = black(); wi = black(); pdf = 0.0; return; }
            wi = t;
            pdf = bxdf_pdf(l, wo, wi);
            f = bxdf_f(l, wo, wi);
            return;
        }
        case LOBE_SPECULAR_REFL: {  // bxdf.rs:543-552
            wi = v3(-wo.x, -wo.y, wo.z);
            f = cmul(l.color, fresnel_evaluate(l, wi.z));
            pdf = 1.0;
            return;
        }
    }
    f = black(); wi = black(); pdf = 0.0;
}

static V3 w2l(const Bsdf& b, V3 v) { return v3(dot(v, b.ss), dot(v, b.ts), dot(v, b.ns)); }  // bsdf.rs:67-69
static V3 l2w(const Bsdf& b, V3 v) {                                                         // bsdf.rs:71-77
    return v3(b.ss.x * v.x + b.ts.x * v.y + b.ns.x * v.z, b.ss.y * v.x + b.ts.y * v.y + b.ns.y * v.z,
              b.ss.z * v.x + b.ts.z * v.y + b.ns.z * v.z);
}
static int num_components(const Bsdf& b, uint8_t flags) {
    int c = 0;
    for (int i = 0; i < b.n; i++)
        if (matches_flags(b.lobes[i].type, flags)) c++;
    return c;
}
// bsdf.rs:83-98 (Q10: `A && B || C` precedence kept)
static V3 bsdf_f(const Bsdf& b, V3 wow, V3 wiw, uint8_t flags) {
    V3 wi = w2l(b, wiw), wo = w2l(b, wow);
    bool refl = dot(wiw, b.ng) * dot(wow, b.ng) > 0.0;
    V3 f = black();
    for (int i = 0; i < b.n; i++) {
        const Lobe& l = b.lobes[i];
        if ((matches_flags(l.type, flags) && (refl && (l.type & RT_BSDF_REFLECTION) > 0)) ||
            (!refl && (l.type & RT_BSDF_TRANSMISSION) > 0))
            f = f + bxdf_f(l, wo, wi);
    }
    return f;
}
// bsdf.rs:166-189 (Q11: lobe pdfs are summed, not averaged)
static double bsdf_pdf(const Bsdf& b, V3 wow, V3 wiw, uint8_t flags) {
    int nc = num_components(b, RT_BSDF_ALL);
    if (nc == 0) return 0.0;
    V3 wo = w2l(b, wow), wi = w2l(b, wiw);
    if (wo.z == 0.0) return 0.0;
    double pdf = 0.0;
    int matching = 0;
    for (int i = 0; i < nc; i++)
        if (matches_flags(b.lobes[i].type, flags)) {
            matching++;
            pdf += bxdf_pdf(b.lobes[i], wo, wi);
        }
    return matching > 0 ? pdf : 0.0;
}
// bsdf.rs:102-164
static void bsdf_sample_f(const Bsdf& b, V3 wow, double u0, double u1, uint8_t type, Rng& rng, V3& color, V3& wiw,
                          double& pdf, uint8_t& sampled) {
    int matching = num_components(b, type);
    color = black(); wiw = black(); pdf = 0.0; sampled = 0;
    if (matching == 0) return;
    int comp = std::min((int)(uint32_t)std::floor(u0 * (double)matching), matching - 1);
    int count = comp, used = -1;
    for (int i = 0; i < b.n; i++)
        if (matches_flags(b.lobes[i].type, type)) {
            if (count == 0) { used = i; break; }
            count--;
        }
    const Lobe& l = b.lobes[used];
    V3 wo = normalize(w2l(b, wow));
    if (wo.z == 0.0) return;
    V3 f, wi;
    double p;
    bxdf_sample_f(l, wo, u0, u1, rng, f, wi, p);
    if (p == 0.0) return;
    V3 wiw_ = l2w(b, wi);
    if ((l.type & RT_BSDF_SPECULAR) == 0 && matching > 1)
        for (int i = 0; i < b.n; i++)
            if (i != used && matches_flags(b.lobes[i].type, type)) p += bxdf_pdf(b.lobes[i], wo, wi);
    if (matching > 1) p = p / (double)matching;
    if ((l.type & RT_BSDF_SPECULAR) == 0) {
        bool refl = dot(wiw_, b.ng) * dot(wow, b.ng) > 0.0;
        f = black();
        for (int i = 0; i < b.n; i++) {
            const Lobe& li = b.lobes[i];
            if (matches_flags(li.type, type) && ((refl && (li.type & RT_BSDF_REFLECTION) > 0) ||
                                                 (!refl && (li.type & RT_BSDF_TRANSMISSION) > 0)))
                f = f + bxdf_f(li, wo, wi);
        }
    }
    color = f; wiw = wiw_; pdf = p; sampled = l.type;
}

// bsdf.rs:26-35 Bsdf::new
static void bsdf_init(Bsdf& b, const Hit& h) {
    b.ns = h.sh_n;
    b.ng = h.n;
    b.ss = h.sh_dpdu;
    b.ts = normalize(cross(h.sh_n, h.sh_dpdu));
    b.n = 0;
}
static Lobe make_lambert(V3 c) {
    Lobe l{};
    l.kind = LOBE_LAMBERT;
    l.type = RT_BSDF_REFLECTION | RT_BSDF_DIFFUSE;
    l.color = c;
    return l;
}
static Lobe make_microfacet(V3 c, double ax, double ay) {  // + make_trowbridge_reitz clamp, microfacet.rs:340-348
    Lobe l{};
    l.kind = LOBE_MICROFACET;
    l.type = RT_BSDF_REFLECTION | RT_BSDF_GLOSSY;
    l.color = c;
    l.alpha_x = rmax(ax, 1e-3);
    l.alpha_y = rmax(ay, 1e-3);
    return l;
}

// material.rs:80-244 Material::compute_scattering(record, arena, RADIANCE, allow_lobes = true)
static int compute_scattering(const Scene& sc, const Hit& h, Bsdf& b) {
    const rt_material& m = sc.mats[h.mat];
    b.n = 0;
    switch (m.kind) {
        case RT_MAT_MATTE: {
            V3 color = texture_value(sc, m.tex[0], h.u, h.v);
            if (!is_black(color)) {
                bsdf_init(b, h);
                b.lobes[b.n++] = make_lambert(color);  // sigma == 0 only (scope)
            }
            break;
        }
        case RT_MAT_LIGHT: break;
        case RT_MAT_PLASTIC: {
            V3 color = texture_value(sc, m.tex[0], h.u, h.v);
            bool inited = false;
            if (!is_black(color)) {
                bsdf_init(b, h);
                inited = true;
                b.lobes[b.n++] = make_lambert(color);
            }
            V3 spec = texture_value(sc, m.tex[1], h.u, h.v);
            if (!is_black(spec)) {
                if (!inited) bsdf_init(b, h);
                double rough = m.f[0];
                if (m.remap_roughness) rough = tr_roughness_to_alpha(rough);
                Lobe l = make_microfacet(spec, rough, rough);
                l.fresnel = FR_DIELECTRIC;
                l.eta_i = 1.5;
                l.eta_t = 1.0;
                b.lobes[b.n++] = l;
            }
            break;
        }
        case RT_MAT_GLASS: {
            V3 r = texture_value(sc, m.tex[0], h.u, h.v);
            V3 t = texture_value(sc, m.tex[1], h.u, h.v);
            bsdf_init(b, h);
            if (is_black(r) && is_black(t)) break;
            double urough = m.f[0], vrough = m.f[1];
            if (urough == 0.0 && vrough == 0.0) {  // is_specular && allow_lobes -> FresnelSpecular
                Lobe l{};
                l.kind = LOBE_FRESNEL_SPECULAR;
                l.type = RT_BSDF_TRANSMISSION | RT_BSDF_REFLECTION | RT_BSDF_SPECULAR;
                l.color = r;
                l.t = t;
                l.eta_a = m.f[2];
                l.eta_b = 1.0;
                b.lobes[b.n++] = l;
                break;
            }
            // material.rs:161-189: MicrofacetReflection + MicrofacetTransmission over one TrowbridgeReitz
            if (m.remap_roughness) {
                urough = tr_roughness_to_alpha(urough);
                vrough = tr_roughness_to_alpha(vrough);
            }
            if (!is_black(r)) {
                Lobe l = make_microfacet(r, urough, vrough);
                l.fresnel = FR_DIELECTRIC;
                l.eta_i = m.f[2];
                l.eta_t = 1.0;
                b.lobes[b.n++] = l;
            }
            if (!is_black(t)) {
                Lobe l = make_microfacet(t, urough, vrough);  // same make_trowbridge_reitz clamp
                l.kind = LOBE_MICRO_TRANS;
                l.type = RT_BSDF_TRANSMISSION | RT_BSDF_GLOSSY;
                l.fresnel = FR_DIELECTRIC;  // bxdf.rs:957-970: FresnelDielectric{eta_a, eta_b}
                l.eta_i = m.f[2];
                l.eta_t = 1.0;
                l.eta_a = m.f[2];
                l.eta_b = 1.0;
                b.lobes[b.n++] = l;
            }
            break;
        }
        case RT_MAT_METAL: {
            bsdf_init(b, h);
            V3 ur = texture_value(sc, m.tex[3] == RT_NO_TEXTURE ? m.tex[2] : m.tex[3], h.u, h.v);
            V3 vr = texture_value(sc, m.tex[4] == RT_NO_TEXTURE ? m.tex[2] : m.tex[4], h.u, h.v);
            double ua = m.remap_roughness ? tr_roughness_to_alpha(ur.x) : ur.x;
            double va = m.remap_roughness ? tr_roughness_to_alpha(vr.x) : vr.x;
            Lobe l = make_microfacet(white(), ua, va);
            l.fresnel = FR_CONDUCTOR;
            l.eta = texture_value(sc, m.tex[0], h.u, h.v);
            l.k = texture_value(sc, m.tex[1], h.u, h.v);
            b.lobes[b.n++] = l;
            break;
        }
        case RT_MAT_MIRROR: {
            bsdf_init(b, h);
            V3 color = texture_value(sc, m.tex[0], h.u, h.v);
            if (!is_black(color)) {
                Lobe l{};
                l.kind = LOBE_SPECULAR_REFL;
                l.type = RT_BSDF_REFLECTION | RT_BSDF_SPECULAR;
                l.color = color;
                l.fresnel = FR_NOOP;
                b.lobes[b.n++] = l;
            }
            break;
        }
    }
    return b.n;
}

// ------------------------------------------------------------- lights
// primitive.rs:339-359
static double prim_area(const Scene& sc, const rt_primitive& pr) {
    switch (pr.kind) {
        case RT_PRIM_SPHERE: return 2.0 * PI * pr.v[3];  // Q8
        case RT_PRIM_TRIANGLE: {
            const Mesh& m = sc.meshes[pr.mesh_index];
            uint32_t i0 = m.ind[pr.tri_ind], i1 = m.ind[pr.tri_ind + 1], i2 = m.ind[pr.tri_ind + 2];
            V3 p0 = v3(m.p[3 * i0], m.p[3 * i0 + 1], m.p[3 * i0 + 2]);
            V3 p1 = v3(m.p[3 * i1], m.p[3 * i1 + 1], m.p[3 * i1 + 2]);
            V3 p2 = v3(m.p[3 * i2], m.p[3 * i2 + 1], m.p[3 * i2 + 2]);
            return 0.5 * norm(cross(p1 - p0, p2 - p0));
        }
        default: return (pr.v[2] - pr.v[0]) * (pr.v[3] - pr.v[1]);
    }
}
// util.rs:51-56
static V3 uniform_sample_sphere(double u0, double u1) {
    double z = 1.0 - 2.0 * u0;
    double r = dm_sqrt(rmax(0.0, 1.0 - z * z));
    double phi = 2.0 * PI * u1;
    return v3(r * dm_cos(phi), r * dm_sin(phi), z);
}
// primitive.rs:478-539 sample_area -> (p, n, pdf); FlipFace negates n
static void sample_area(const Scene& sc, const rt_primitive& pr, double u0, double u1, V3& p, V3& n, double& pdf) {
    switch (pr.kind) {
        case RT_PRIM_SPHERE: {  // Q8: centre offset omitted
            p = pr.v[3] * uniform_sample_sphere(u0, u1);
            n = normalize(p);
            break;
        }
        case RT_PRIM_TRIANGLE: {
            const Mesh& m = sc.meshes[pr.mesh_index];
            uint32_t i0 = m.ind[pr.tri_ind], i1 = m.ind[pr.tri_ind + 1], i2 = m.ind[pr.tri_ind + 2];
            double s0 = dm_sqrt(u0);  // util.rs:62-65
            double b0 = 1.0 - s0, b1 = u1 * s0;
            V3 p0 = v3(m.p[3 * i0], m.p[3 * i0 + 1], m.p[3 * i0 + 2]);
            V3 p1 = v3(m.p[3 * i1], m.p[3 * i1 + 1], m.p[3 * i1 + 2]);
            V3 p2 = v3(m.p[3 * i2], m.p[3 * i2 + 1], m.p[3 * i2 + 2]);
            p = (b0 * p0 + b1 * p1 + (1.0 - b0 - b1) * p2);
            if (!m.n.empty()) {
                V3 n0 = v3(m.n[3 * i0], m.n[3 * i0 + 1], m.n[3 * i0 + 2]);
                V3 n1 = v3(m.n[3 * i1], m.n[3 * i1 + 1], m.n[3 * i1 + 2]);
                V3 n2 = v3(m.n[3 * i2], m.n[3 * i2 + 1], m.n[3 * i2 + 2]);
                n = normalize(b0 * n0 + b1 * n1 + (1.0 - b0 - b1) * n2);
            } else {
                n = normalize(cross(p1 - p0, p2 - p0));
            }
            break;
        }
        case RT_PRIM_XY_RECT:
            n = v3(0, 0, 1);
            p = v3(pr.v[0] + u0 * (pr.v[2] - pr.v[0]), pr.v[1] + u1 * (pr.v[3] - pr.v[1]), pr.v[4]);
            break;
        case RT_PRIM_XZ_RECT:
            n = v3(0, 1, 0);
            p = v3(pr.v[0] + u0 * (pr.v[2] - pr.v[0]), pr.v[4], pr.v[1] + u1 * (pr.v[3] - pr.v[1]));
            break;
        default:
            n = v3(1, 0, 0);
            p = v3(pr.v[4], pr.v[0] + u0 * (pr.v[2] - pr.v[0]), pr.v[1] + u1 * (pr.v[3] - pr.v[1]));
            break;
    }
    pdf = 1.0 / prim_area(sc, pr);
    if (pr.flip) n = -n;
}
// light.rs:475-496 Light::l
static V3 light_l(const rt_light& lt, V3 n, V3 w) {
    if (dot(n, w) > 0.0 || lt.two_sided) return v3(lt.color[0], lt.color[1], lt.color[2]);
    return black();
}
// primitive.rs:462-473 Primitive::pdf: re-intersects the single light primitive, tmin 0
static double prim_pdf(const Scene& sc, const rt_primitive& pr, const Hit& rec, V3 dir) {
    Ray ray{rec.p, dir};
    Hit nh;
    if (!intersects_obj(sc, pr, ray, 0.0, INF, nh, nullptr)) return 0.0;
    V3 dist = rec.p - nh.p;
    return norm2(dist) / (prim_area(sc, pr) * std::fabs(dot(nh.n, -dir)));
}

// ------------------------------------------------- Distribution1D / 2D
// distribution.rs:27-55 make_distribution
static void dist1d_make(Dist1D& d, const double* f, size_t n) {
    d.func = f;
    d.n = n;
    d.cdf.assign(n + 1, 0.0);
    for (size_t i = 1; i < n + 1; i++) d.cdf[i] = d.cdf[i - 1] + f[i - 1] / (double)n;
    d.func_int = d.cdf[n];
    if (d.func_int == 0.0) {
        for (size_t i = 1; i < n + 1; i++) d.cdf[i] = (double)i / (double)n;
    } else {
        for (size_t i = 1; i < n + 1; i++) d.cdf[i] = d.cdf[i] / d.func_int;
    }
}
// distribution.rs:152-166 find_interval with pred = cdf[index] <= u
static size_t find_interval(const std::vector<double>& cdf, double u) {
    const size_t size = cdf.size();
    size_t first = 0, len = size;
    while (len > 0) {
        size_t half = len >> 1, middle = first + half;
        if (cdf[middle] <= u) {
            first = middle + 1;
            len = len - half - 1;
        } else {
            len = half;
        }
    }
    // clamp((first - 1) as f64, 0, size - 2) as usize; first == 0 (u < 0 or NaN) wraps in release builds
    double x = first == 0 ? 18446744073709551615.0 : (double)(first - 1);
    return sat_usize(clampd(x, 0.0, (double)(size - 2)));
}
// distribution.rs:64-78 sample_continuous
static void dist1d_sample(const Dist1D& d, double u, double& x, double& pdf, size_t& offset) {
    offset = find_interval(d.cdf, u);
    double du = u - d.cdf[offset];
    if (d.cdf[offset + 1] - d.cdf[offset] > 0.0) du = du / (d.cdf[offset + 1] - d.cdf[offset]);
    pdf = d.func_int > 0.0 ? d.func[offset] / d.func_int : 0.0;
    x = ((double)offset + du) / (double)d.n;
}
// light.rs:608-638 make_infinite_light's image + distribution.rs:115-128 make_distribution_2d.
// The reference slices row v as f[v .. v + nu] (not f[v * nu ..]): restated literally.
static void env_dist_make(Scene& sc, const rt_texture& t) {
    Dist2D& d = sc.env_dist;
    const size_t width = (size_t)t.width * 2, height = (size_t)t.height * 2;
    d.img.assign(width * height, 0.0);
    for (size_t v = 0; v < height; v++) {
        double vp = ((double)v + 0.5) / (double)height;
        double sin_theta = dm_sin(PI * ((double)v + 0.5) / (double)height);
        for (size_t u = 0; u < width; u++) {
            double up = (double)u / (double)width;
            V3 color = hdr_value(t, up, vp);
            double lum = 0.2126 * color.x + 0.7152 * color.y + 0.0722 * color.z;  // util.rs:169-171
            d.img[u + v * width] = lum * sin_theta;
        }
    }
    d.p_cond_v.resize(height);
    d.marginal_func.resize(height);
    for (size_t v = 0; v < height; v++) {
        dist1d_make(d.p_cond_v[v], d.img.data() + v, width);
        d.marginal_func[v] = d.p_cond_v[v].func_int;
    }
    dist1d_make(d.p_marginal, d.marginal_func.data(), height);
}
// distribution.rs:133-137
static void dist2d_sample(const Dist2D& d, double u0, double u1, double& x0, double& x1, double& pdf) {
    double pdf1, pdf0;
    size_t v, off;
    dist1d_sample(d.p_marginal, u1, x1, pdf1, v);
    dist1d_sample(d.p_cond_v[v], u0, x0, pdf0, off);
    pdf = pdf1 * pdf0;
}
// distribution.rs:139-145
static double dist2d_pdf(const Dist2D& d, double p0, double p1) {
    size_t iu = sat_usize(p0 * (double)d.p_cond_v[0].n);
    iu = std::min(iu, d.p_cond_v[0].n - 1);
    size_t iv = sat_usize(p1 * (double)d.p_marginal.n);
    iv = std::min(iv, d.p_marginal.n - 1);
    return d.p_cond_v[iv].func[iu] / d.p_marginal.func_int;
}

// ------------------------------------------------------ Light::Infinite
static const double INV_2PI = 1.0 / (2.0 * PI);  // consts.rs:42
static double spherical_theta(V3 v) { return dm_acos(clampd(v.y, -1.0, 1.0)); }  // util.rs:153-155
static double spherical_phi(V3 v) {                                              // util.rs:160-167
    double p = dm_atan2(v.z, v.x);
    return p < 0.0 ? p + 2.0 * PI : p;
}
static V3 light_to_world(const Scene& sc, const rt_light& lt, V3 v) {
    return lt.xform_index >= 0 ? xf_vector(sc.xforms[lt.xform_index].fwd, v) : v;
}
static V3 light_to_obj(const Scene& sc, const rt_light& lt, V3 v) {
    return lt.xform_index >= 0 ? xf_vector(sc.xforms[lt.xform_index].inv, v) : v;
}
// light.rs:499-512 Light::le (black for every other kind)
static V3 light_le(const Scene& sc, const rt_light& lt, V3 dir) {
    if (lt.kind != RT_LIGHT_INFINITE) return black();
    V3 w = normalize(light_to_obj(sc, lt, dir));
    return texture_value(sc, lt.tex_index, spherical_phi(w) * INV_2PI, spherical_theta(w) * INV_PI);
}
// light.rs:285-294 pdf_li (uses to_world, as the reference does)
static double infinite_pdf_li(const Scene& sc, const rt_light& lt, V3 wi) {
    V3 w = light_to_world(sc, lt, wi);
    double theta = spherical_theta(w), phi = spherical_phi(w);
    double sin_theta = dm_sin(theta);
    if (sin_theta == 0.0) return 0.0;
    return dist2d_pdf(sc.env_dist, phi * INV_2PI, theta * INV_PI) / (2.0 * PI * PI * sin_theta);
}

// ------------------------------------------------------------ integrator
struct Ctx {
    const Scene& sc;
    int mode;
    uint32_t max_depth;
    Counters c;
};

// hittable.rs:25-39 Visibility::unoccluded (Q13)
static bool unoccluded(Ctx& cx, V3 p0, V3 p1, int32_t light_prim) {
    V3 dir = p1 - p0;
    Ray ray{p0 + dir * SMALL, dir};
    Hit h;
    cx.c.r2++;
    if (!closest_hit(cx.sc, cx.mode, ray, 0.0, INF, h, &cx.c)) return false;  // infinite_light == false
    return h.prim == light_prim;
}
static bool unoccluded_infinite(Ctx& cx, V3 p0, V3 p1) {  // unoccluded(true): only a miss is unoccluded
    V3 dir = p1 - p0;
    Ray ray{p0 + dir * SMALL, dir};
    Hit h;
    cx.c.r2++;
    return !closest_hit(cx.sc, cx.mode, ray, 0.0, INF, h, &cx.c);
}

// integrator.rs:655-659
static double power_heuristic(int nf, double f_pdf, int ng, double g_pdf) {
    double f = (double)nf * f_pdf, g = (double)ng * g_pdf;
    return (f * f) / (f * f + g * g);
}

// integrator.rs:559-634 estimate_direct(.., specular = false) with light.rs:176-203, 284
static V3 estimate_direct(Ctx& cx, const Hit& rec, const Bsdf& bsdf, double us0, double us1, int light_idx,
                          double ul0, double ul1, Rng& rng) {
    const Scene& sc = cx.sc;
    const rt_light& lt = sc.lights[light_idx];
    const bool infinite = lt.kind == RT_LIGHT_INFINITE;
    static const rt_primitive no_prim{};
    const rt_primitive& lp = infinite ? no_prim : sc.prims[lt.prim_index];
    const uint8_t flags = RT_BSDF_ALL - RT_BSDF_SPECULAR;
    V3 ld = black();
    V3 sp, sn;
    double light_pdf;
    V3 wi, color;
    if (infinite) {
        // Light::sample_li, Infinite arm (light.rs:204-245)
        double uv0, uv1, map_pdf;
        dist2d_sample(sc.env_dist, ul0, ul1, uv0, uv1, map_pdf);
        if (map_pdf == 0.0) {
            wi = normalize(black());
            light_pdf = 0.0;
            color = black();
            sp = black();
        } else {
            double theta = uv1 * PI, phi = uv0 * 2.0 * PI;
            double cos_theta = dm_cos(theta), sin_theta = dm_sin(theta);
            double cos_phi = dm_cos(phi), sin_phi = dm_sin(phi);
            V3 v = v3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
            V3 wiv = light_to_world(sc, lt, v);
            light_pdf = map_pdf / (2.0 * PI * PI * sin_theta);
            if (sin_theta == 0.0) light_pdf = 0.0;
            sp = rec.p + wiv * (2.0 * lt.world_radius);  // Visibility's far end
            color = texture_value(sc, lt.tex_index, uv0, uv1);
            wi = normalize(wiv);
        }
    } else {
        // Light::sample_li (Diffuse) -> Primitive::sample (Q9)
        sample_area(sc, lp, ul0, ul1, sp, sn, light_pdf);
        V3 wi_raw = sp - rec.p;
        if (norm2(wi_raw) == 0.0) {
            light_pdf = 0.0;
        } else {
            V3 wn = normalize(wi_raw);
            light_pdf = light_pdf * norm2(rec.p - sp) / std::fabs(dot(sn, -wn));
        }
        if (light_pdf == 0.0 || norm2(rec.p - sp) == 0.0) {
            light_pdf = 0.0;
            wi = normalize(black());
            color = v3(lt.color[0], lt.color[1], lt.color[2]);
        } else {
            wi = normalize(sp - rec.p);
            color = light_l(lt, sn, -wi);
        }
    }
    double scattering_pdf;
    if (light_pdf > 0.0 && !is_black(color)) {
        V3 f = bsdf_f(bsdf, rec.wo, wi, flags) * std::fabs(dot(wi, rec.sh_n));
        scattering_pdf = bsdf_pdf(bsdf, rec.wo, wi, flags);
        if (!is_black(f)) {
            if (infinite) {
                if (!unoccluded_infinite(cx, rec.p, sp)) color = black();
            } else if (!unoccluded(cx, rec.p, sp, (int32_t)lt.prim_index)) {
                color = black();
            }
            if (!is_black(color)) {
                double weight = power_heuristic(1, light_pdf, 1, scattering_pdf);
                ld = ld + cmul(f, color) * (weight / light_pdf);
            }
        }
    }
    // BSDF-sampling half (area lights are never delta)
    {
        V3 f, wi2;
        double spdf;
        uint8_t sampled;
        bsdf_sample_f(bsdf, rec.wo, us0, us1, flags, rng, f, wi2, spdf, sampled);
        f = f * std::fabs(dot(wi2, rec.sh_n));
        bool sampled_specular = (sampled & RT_BSDF_SPECULAR) != 0;
        if (!is_black(f) && spdf > 0.0) {
            double weight = 1.0;
            if (!sampled_specular) {
                light_pdf = infinite ? infinite_pdf_li(sc, lt, wi2) : prim_pdf(sc, lp, rec, wi2);  // Light::pdf_li
                if (light_pdf == 0.0) return ld;
                weight = power_heuristic(1, spdf, 1, light_pdf);
            }
            Ray nr{rec.p, wi2};
            Hit nh;
            cx.c.r3++;
            V3 col = black();
            if (closest_hit(sc, cx.mode, nr, SMALL, INF, nh, &cx.c)) {
                int32_t li = sc.prims[nh.prim].light_index;
                if (li >= 0 && li == light_idx) col = light_l(sc.lights[li], nh.n, -wi2);  // new_record.le(-wi)
            } else {
                col = light_le(sc, lt, wi2);  // light.le(&new_ray): non-black only for Light::Infinite
            }
            if (!is_black(col)) ld = ld + cmul(f, col) * (weight / spdf);
        }
    }
    return ld;
}

// integrator.rs:530-557
static V3 uniform_sample_one_light(Ctx& cx, const Hit& rec, const Bsdf& bsdf, Rng& rng) {
    size_t n_lights = cx.sc.lights.size();
    if (n_lights == 0) return black();
    double pick = rng.next();
    size_t light_num = std::min(n_lights - 1, (size_t)(pick * (double)n_lights));
    double ul0 = rng.next(), ul1 = rng.next();
    double us0 = rng.next(), us1 = rng.next();
    return estimate_direct(cx, rec, bsdf, us0, us1, (int)light_num, ul0, ul1, rng) * (double)n_lights;
}

// integrator.rs:375-445 PathIntegrator::li (invisible_light = false), Q18
static V3 li(Ctx& cx, Ray ray, Rng& rng) {
    V3 beta = white(), l = black();
    bool specular_bounce = false;
    uint32_t bounces = 0;
    const Scene& sc = cx.sc;
    for (;;) {
        Hit rec;
        cx.c.r1++;
        bool is_some = closest_hit(sc, cx.mode, ray, SMALL, INF, rec, &cx.c);
        if (bounces == 0 || specular_bounce) {
            if (is_some) {
                int32_t li_ = sc.prims[rec.prim].light_index;
                if (li_ >= 0) l = l + cmul(light_l(sc.lights[li_], rec.n, -ray.d), beta);  // record.le(-ray.dir)
            }
            else {
                for (const rt_light& lt : sc.lights) l = l + cmul(light_le(sc, lt, ray.d), beta);  // integrator.rs:403-407
            }
        }
        if (!is_some || bounces >= cx.max_depth) break;
        Bsdf bsdf;
        compute_scattering(sc, rec, bsdf);
        cx.c.vertices++;
        l = l + cmul(uniform_sample_one_light(cx, rec, bsdf, rng), beta);
        V3 wo = -ray.d;
        double u0 = rng.next(), u1 = rng.next();
        V3 f, wi;
        double pdf;
        uint8_t flags;
        bsdf_sample_f(bsdf, wo, u0, u1, RT_BSDF_ALL, rng, f, wi, pdf, flags);
        if (is_black(f) || pdf == 0.0) break;
        beta = cmul(beta, f) * (std::fabs(dot(wi, rec.sh_n)) / pdf);
        specular_bounce = (flags & RT_BSDF_SPECULAR) != 0;
        ray = Ray{rec.p, wi};  // spawn_ray: no offset (Q4)
        if (bounces > 3) {
            double q = rmax(0.05, 1.0 - rmax(beta.x, rmax(beta.y, beta.z)));
            if (rng.next() < q) break;
            beta = beta * (1.0 / (1.0 - q));
        }
        bounces = bounces + 1;
    }
    return l;
}

// geometry.rs:177-190 Camera::get_ray with util.rs:105-113, 36-38
static Ray camera_get_ray(const rt_camera& cam, double u, double v, Rng& rng) {
    double dx, dy;
    for (;;) {
        dx = rng.next();
        dy = rng.next();
        if (dx * dx + dy * dy < 1.0) break;
    }
    V3 in_disk = v3(dx, dy, 0.0) * cam.lens_radius;
    V3 cu = v3(cam.u[0], cam.u[1], cam.u[2]), cv = v3(cam.v[0], cam.v[1], cam.v[2]);
    V3 offset = cu * in_disk.x + cv * in_disk.y;
    V3 origin = v3(cam.origin[0], cam.origin[1], cam.origin[2]);
    V3 ulc = v3(cam.upper_left_corner[0], cam.upper_left_corner[1], cam.upper_left_corner[2]);
    V3 ho = v3(cam.horizontal_offset[0], cam.horizontal_offset[1], cam.horizontal_offset[2]);
    V3 vo = v3(cam.vertical_offset[0], cam.vertical_offset[1], cam.vertical_offset[2]);
    V3 to = ulc + ho * u - vo * v;
    V3 dir = to - origin;
    (void)rng.next();  // rand_range(t0, t1): the time draw; no moving geometry reads it
    return Ray{origin + offset, dir - offset};
}

// integrator.rs:357-369 single_sample with sampler.rs:606-613
static V3 single_sample(Ctx& cx, const rt_camera& cam, uint32_t W, uint32_t H, uint32_t px, uint32_t py,
                        Rng& rng) {
    double ox = rng.next(), oy = rng.next();
    (void)rng.next();  // time
    (void)rng.next();  // lens.x
    (void)rng.next();  // lens.y
    double fx = (double)px + ox, fy = (double)py + oy;
    Ray ray = camera_get_ray(cam, fx / (double)W, fy / (double)H, rng);
    cx.c.paths++;
    return li(cx, ray, rng);
}

static uint32_t next_pow2(uint32_t v) {  // sampler.rs:633-642
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace orc

// ============================================================== C entry points
using namespace orc;

struct oracle_scene {
    Scene sc;
};

extern "C" {

int oracle_scene_create(const rt_scene_desc* d, oracle_scene** out) {
    if (!d || !out) return RT_ERR_INVALID_ARG;
    auto* h = new oracle_scene();
    Scene& sc = h->sc;
    for (uint64_t i = 0; i < d->n_meshes; i++) {
        const rt_mesh& m = d->meshes[i];
        Mesh mm;
        mm.p.assign(m.p, m.p + m.n_p * 3);
        if (m.n_n) mm.n.assign(m.n, m.n + m.n_n * 3);
        if (m.n_uv) mm.uv.assign(m.uv, m.uv + m.n_uv * 2);
        mm.ind.assign(m.ind, m.ind + m.n_ind);
        sc.meshes.push_back(std::move(mm));
    }
    sc.prims.assign(d->prims, d->prims + d->n_prims);
    if (d->n_xforms) sc.xforms.assign(d->xforms, d->xforms + d->n_xforms);
    sc.mats.assign(d->materials, d->materials + d->n_materials);
    sc.texs.assign(d->textures, d->textures + d->n_textures);
    if (d->n_lights) sc.lights.assign(d->lights, d->lights + d->n_lights);
    sc.hdr.resize(sc.texs.size());
    for (size_t i = 0; i < sc.texs.size(); i++) {
        rt_texture& t = sc.texs[i];
        if (t.kind != RT_TEX_HDR) continue;
        if (!t.rgbe || t.width == 0 || t.height == 0) { delete h; return RT_ERR_INVALID_ARG; }
        sc.hdr[i].assign(t.rgbe, t.rgbe + (size_t)t.width * t.height * 4);
        t.rgbe = sc.hdr[i].data();
    }
    for (size_t i = 0; i < sc.lights.size(); i++) {
        const rt_light& lt = sc.lights[i];
        if (lt.kind != RT_LIGHT_INFINITE) continue;
        if (sc.env_light >= 0 || lt.tex_index >= sc.texs.size() || sc.texs[lt.tex_index].kind != RT_TEX_HDR) {
            delete h;
            return RT_ERR_UNSUPPORTED;
        }
        sc.env_light = (int32_t)i;
        env_dist_make(sc, sc.texs[lt.tex_index]);
    }
    if (!sc.prims.empty()) {
        std::vector<int32_t> idx(sc.prims.size());
        for (size_t i = 0; i < idx.size(); i++) idx[i] = (int32_t)i;
        sc.nodes.reserve(2 * idx.size());
        sc.root = bvh_build(sc, idx, 0, idx.size());
    }
    *out = h;
    return RT_OK;
}

int oracle_scene_destroy(oracle_scene* s) {
    delete s;
    return RT_OK;
}

int oracle_render(const oracle_scene* s, const rt_camera* cam, const rt_render_cfg* cfg, int traversal_mode,
                  int n_threads, double* rgb_sum, uint32_t* n, rt_stats* stats) {
    if (!s || !cam || !cfg || !rgb_sum || !n) return RT_ERR_INVALID_ARG;
    const uint32_t W = cfg->width, H = cfg->height;
    if (W == 0 || H == 0 || cfg->spp == 0) return RT_ERR_INVALID_ARG;
    const uint32_t spp = next_pow2(cfg->spp);
    uint32_t x0 = cfg->x0, y0 = cfg->y0, x1 = cfg->x1, y1 = cfg->y1;
    if (x1 == 0 && y1 == 0) { x0 = 0; y0 = 0; x1 = W; y1 = H; }
    if (x1 > W || y1 > H || x0 > x1 || y0 > y1) return RT_ERR_INVALID_ARG;
    const uint32_t ts = cfg->tile_size ? cfg->tile_size : 16;
    const uint32_t world = cfg->tile_world ? cfg->tile_world : 1;
    const uint32_t rank = cfg->tile_rank;
    if (rank >= world) return RT_ERR_INVALID_ARG;
    const uint32_t tw = (W + ts - 1) / ts, th = (H + ts - 1) / ts;
    std::memset(rgb_sum, 0, sizeof(double) * 3 * (size_t)W * H);
    std::memset(n, 0, sizeof(uint32_t) * (size_t)W * H);
    if (n_threads < 1) n_threads = 1;
    std::atomic<uint32_t> next_tile{0};
    std::vector<Counters> per(n_threads);
    auto worker = [&](int tid) {
        Ctx cx{s->sc, traversal_mode, cfg->max_depth, Counters()};
        for (;;) {  // render.rs:49-71: workers claim 16x16 tiles in row-major order
            uint32_t k = next_tile.fetch_add(1);
            if (k >= tw * th) break;
            uint32_t tx = k % tw, ty = k / tw;
            if (rtabi_tile_owner(tx, ty, world) != rank) continue;  // include/rt_abi.h: the interface's tile -> rank rule
            for (uint32_t y = 0; y < ts; y++)
                for (uint32_t x = 0; x < ts; x++) {
                    uint32_t px = tx * ts + x, py = ty * ts + y;
                    if (px < x0 || px >= x1 || py < y0 || py >= y1) continue;
                    size_t pix = (size_t)py * W + px;
                    for (uint32_t sidx = 0; sidx < spp; sidx++) {  // integrator.rs:341-347
                        Rng rng(cfg->seed, pix, sidx);
                        V3 c = single_sample(cx, *cam, W, H, px, py, rng);
                        rgb_sum[pix * 3 + 0] += c.x;  // util.rs:208-232 (Q17: NaNs propagate)
                        rgb_sum[pix * 3 + 1] += c.y;
                        rgb_sum[pix * 3 + 2] += c.z;
                        n[pix] += 1;
                    }
                }
        }
        per[tid] = cx.c;
    };
    std::vector<std::thread> th_;
    for (int t = 0; t < n_threads; t++) th_.emplace_back(worker, t);
    for (auto& t : th_) t.join();
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        Counters tot;
        for (auto& c : per) tot.add(c);
        stats->paths = tot.paths;
        stats->rays_extension = tot.r1;
        stats->rays_shadow = tot.r2;
        stats->rays_probe = tot.r3;
        stats->vertices_shaded = tot.vertices;
        stats->nodes_fetched = tot.nodes;
        stats->tris_tested = tot.tris;
        stats->others_tested = tot.others;
    }
    return RT_OK;
}

int oracle_intersect_batch(const oracle_scene* s, const rt_ray* rays, uint64_t n, int traversal_mode,
                           rt_hit* hits) {
    if (!s || !rays || !hits) return RT_ERR_INVALID_ARG;
    for (uint64_t i = 0; i < n; i++) {
        Ray r{v3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]),
              v3(rays[i].dir[0], rays[i].dir[1], rays[i].dir[2])};
        Hit h;
        if (closest_hit(s->sc, traversal_mode, r, rays[i].tmin, rays[i].tmax, h, nullptr)) {
            hits[i].t = h.t;
            hits[i].prim = h.prim;
        } else {
            hits[i].t = RT_INFINITY;
            hits[i].prim = -1;
        }
        hits[i].reserved = 0;
    }
    return RT_OK;
}

// Full record of one primitive test (Primitive::intersects), for per-stage vectors.
int oracle_prim_intersect(const oracle_scene* s, int32_t prim, const rt_ray* ray, oracle_hit_record* out) {
    if (!s || !ray || !out || prim < 0 || prim >= (int32_t)s->sc.prims.size()) return RT_ERR_INVALID_ARG;
    Ray r{v3(ray->origin[0], ray->origin[1], ray->origin[2]), v3(ray->dir[0], ray->dir[1], ray->dir[2])};
    Hit h;
    std::memset(out, 0, sizeof(*out));
    if (!prim_intersects(s->sc, prim, r, ray->tmin, ray->tmax, h, nullptr)) return RT_OK;
    out->hit = 1;
    out->t = h.t;
    out->front = h.front;
    out->uv[0] = h.u; out->uv[1] = h.v;
    const V3* src[] = {&h.p, &h.n, &h.sh_n, &h.sh_dpdu, &h.sh_dpdv};
    double* dst[] = {out->p, out->n, out->sh_n, out->sh_dpdu, out->sh_dpdv};
    for (int i = 0; i < 5; i++) {
        dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z;
    }
    return RT_OK;
}

// One camera sample, returning radiance and the number of RNG draws consumed.
int oracle_sample(const oracle_scene* s, const rt_camera* cam, const rt_render_cfg* cfg, uint32_t px, uint32_t py,
                  uint32_t sample, int traversal_mode, double* rgb, rt_stats* stats) {
    if (!s || !cam || !cfg || !rgb) return RT_ERR_INVALID_ARG;
    Ctx cx{s->sc, traversal_mode, cfg->max_depth, Counters()};
    Rng rng(cfg->seed, (uint64_t)py * cfg->width + px, sample);
    V3 c = single_sample(cx, *cam, cfg->width, cfg->height, px, py, rng);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->paths = cx.c.paths;
        stats->rays_extension = cx.c.r1;
        stats->rays_shadow = cx.c.r2;
        stats->rays_probe = cx.c.r3;
        stats->vertices_shaded = cx.c.vertices;
    }
    return RT_OK;
}

// Every root closest-hit query (R1/R2/R3) of one camera sample, in program order, with its result.
int64_t oracle_sample_rays(const oracle_scene* s, const rt_camera* cam, const rt_render_cfg* cfg, uint32_t px,
                           uint32_t py, uint32_t sample, int traversal_mode, rt_ray* rays, rt_hit* hits,
                           uint64_t capacity) {
    if (!s || !cam || !cfg) return RT_ERR_INVALID_ARG;
    RayLog log;
    g_ray_log = &log;
    Ctx cx{s->sc, traversal_mode, cfg->max_depth, Counters()};
    Rng rng(cfg->seed, (uint64_t)py * cfg->width + px, sample);
    (void)single_sample(cx, *cam, cfg->width, cfg->height, px, py, rng);
    g_ray_log = nullptr;
    const uint64_t n = std::min<uint64_t>(capacity, log.rays.size());
    for (uint64_t i = 0; i < n; i++) {
        if (rays) rays[i] = log.rays[i];
        if (hits) hits[i] = log.hits[i];
    }
    return (int64_t)log.rays.size();
}

// ---- known-answer entry points
double oracle_fr_dielectric(double c, double ei, double et) { return fr_dielectric(c, ei, et); }
void oracle_fr_conductor(double c, const double* eta, const double* k, double* out) {
    V3 r = fr_conductor(c, v3(eta[0], eta[1], eta[2]), v3(k[0], k[1], k[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
double oracle_power_heuristic(int nf, double f, int ng, double g) { return power_heuristic(nf, f, ng, g); }
double oracle_tr_d(double ax, double ay, const double* wh) { return tr_d(ax, ay, v3(wh[0], wh[1], wh[2])); }
double oracle_tr_lambda(double ax, double ay, const double* w) { return tr_lambda(ax, ay, v3(w[0], w[1], w[2])); }
double oracle_tr_g(double ax, double ay, const double* wo, const double* wi) {
    return tr_g(ax, ay, v3(wo[0], wo[1], wo[2]), v3(wi[0], wi[1], wi[2]));
}
double oracle_tr_pdf(double ax, double ay, const double* wo, const double* wh) {
    return tr_pdf(ax, ay, v3(wo[0], wo[1], wo[2]), v3(wh[0], wh[1], wh[2]));
}
void oracle_tr_sample_wh(double ax, double ay, const double* wo, double u0, double u1, double* out) {
    V3 r = tr_sample_wh(ax, ay, v3(wo[0], wo[1], wo[2]), u0, u1);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
double oracle_tr_roughness_to_alpha(double r) { return tr_roughness_to_alpha(r); }
void oracle_concentric_sample_disk(double u0, double u1, double* out) { concentric_sample_disk(u0, u1, out[0], out[1]); }
void oracle_rand_cosine_dir(double r1, double r2, double* out) {
    V3 r = rand_cosine_dir(r1, r2);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void oracle_rng_draws(uint64_t seed, uint64_t pixel, uint64_t sample, uint32_t n, double* out) {
    Rng r(seed, pixel, sample);
    for (uint32_t i = 0; i < n; i++) out[i] = r.next();
}
int oracle_box_intersects(const double* bmin, const double* bmax, const rt_ray* ray) {
    Ray r{v3(ray->origin[0], ray->origin[1], ray->origin[2]), v3(ray->dir[0], ray->dir[1], ray->dir[2])};
    return box_intersects(bmin, bmax, r, ray->tmin, ray->tmax) ? 1 : 0;
}
int oracle_refract(const double* v, const double* n, double eta, double* out) {
    V3 o;
    if (!refract(v3(v[0], v[1], v[2]), v3(n[0], n[1], n[2]), eta, o)) return 0;
    out[0] = o.x; out[1] = o.y; out[2] = o.z;
    return 1;
}
// Lambertian lobe: f, pdf for local wo, wi (white furnace test)
void oracle_lambert_f_pdf(const double* color, const double* wo, const double* wi, double* f, double* pdf) {
    Lobe l = make_lambert(v3(color[0], color[1], color[2]));
    V3 r = bxdf_f(l, v3(wo[0], wo[1], wo[2]), v3(wi[0], wi[1], wi[2]));
    f[0] = r.x; f[1] = r.y; f[2] = r.z;
    *pdf = bxdf_pdf(l, v3(wo[0], wo[1], wo[2]), v3(wi[0], wi[1], wi[2]));
}
// Microfacet conductor lobe f / pdf (local frame)
void oracle_microfacet_f_pdf(double ax, double ay, const double* eta, const double* k, const double* wo,
                             const double* wi, double* f, double* pdf) {
    Lobe l = make_microfacet(white(), ax, ay);
    l.fresnel = FR_CONDUCTOR;
    l.eta = v3(eta[0], eta[1], eta[2]);
    l.k = v3(k[0], k[1], k[2]);
    V3 r = bxdf_f(l, v3(wo[0], wo[1], wo[2]), v3(wi[0], wi[1], wi[2]));
    f[0] = r.x; f[1] = r.y; f[2] = r.z;
    *pdf = bxdf_pdf(l, v3(wo[0], wo[1], wo[2]), v3(wi[0], wi[1], wi[2]));
}
// MicrofacetTransmission lobe (next-row f4), local frame: f, pdf for (wo, wi) and one sample_f draw
void oracle_micro_trans(double ax, double ay, double eta, const double* color, const double* wo, const double* wi,
                        double u0, double u1, double* f, double* pdf, double* s_wi, double* s_f, double* s_pdf) {
    Lobe l = make_microfacet(v3(color[0], color[1], color[2]), ax, ay);
    l.kind = LOBE_MICRO_TRANS;
    l.type = RT_BSDF_TRANSMISSION | RT_BSDF_GLOSSY;
    l.fresnel = FR_DIELECTRIC;
    l.eta_i = eta; l.eta_t = 1.0;
    l.eta_a = eta; l.eta_b = 1.0;
    V3 o = v3(wo[0], wo[1], wo[2]), i = v3(wi[0], wi[1], wi[2]);
    V3 r = bxdf_f(l, o, i);
    f[0] = r.x; f[1] = r.y; f[2] = r.z;
    *pdf = bxdf_pdf(l, o, i);
    Rng rng(0, 0, 0);
    V3 sf, swi;
    bxdf_sample_f(l, o, u0, u1, rng, sf, swi, *s_pdf);
    s_wi[0] = swi.x; s_wi[1] = swi.y; s_wi[2] = swi.z;
    s_f[0] = sf.x; s_f[1] = sf.y; s_f[2] = sf.z;
}
// Generic single-lobe entry points for the independent value tables (tests/golden/lobe_tables.json, made by
// tools/make_lobe_tables.py with mpmath straight from bxdf.rs / microfacet.rs): local-frame f + pdf, and one
// sample_f draw.  `rng_key` = (seed, pixel, sample) of the counter stream that feeds default_sample_f's two
// entropy draws (Lambert only).
static Lobe lobe_from_desc(const oracle_lobe* d) {
    Lobe l;
    l.kind = d->kind;
    l.fresnel = d->fresnel;
    l.color = v3(d->color[0], d->color[1], d->color[2]);
    l.t = v3(d->t[0], d->t[1], d->t[2]);
    l.eta_i = d->eta_i; l.eta_t = d->eta_t;
    l.eta = v3(d->eta[0], d->eta[1], d->eta[2]);
    l.k = v3(d->k[0], d->k[1], d->k[2]);
    l.alpha_x = d->alpha_x; l.alpha_y = d->alpha_y;
    l.eta_a = d->eta_a; l.eta_b = d->eta_b;
    switch (d->kind) {
        case LOBE_LAMBERT: l.type = RT_BSDF_REFLECTION | RT_BSDF_DIFFUSE; break;
        case LOBE_MICROFACET: l.type = RT_BSDF_REFLECTION | RT_BSDF_GLOSSY; break;
        case LOBE_FRESNEL_SPECULAR: l.type = RT_BSDF_REFLECTION | RT_BSDF_TRANSMISSION | RT_BSDF_SPECULAR; break;
        case LOBE_SPECULAR_REFL: l.type = RT_BSDF_REFLECTION | RT_BSDF_SPECULAR; break;
        default: l.type = RT_BSDF_TRANSMISSION | RT_BSDF_GLOSSY; break;
    }
    return l;
}
void oracle_lobe_eval(const oracle_lobe* d, const double* wo, const double* wi, double* f, double* pdf) {
    const Lobe l = lobe_from_desc(d);
    const V3 o = v3(wo[0], wo[1], wo[2]), i = v3(wi[0], wi[1], wi[2]);
    const V3 r = bxdf_f(l, o, i);
    f[0] = r.x; f[1] = r.y; f[2] = r.z;
    *pdf = bxdf_pdf(l, o, i);
}
void oracle_lobe_sample(const oracle_lobe* d, const double* wo, double u0, double u1, const uint64_t* rng_key,
                        double* f, double* wi, double* pdf) {
    const Lobe l = lobe_from_desc(d);
    Rng rng(rng_key[0], rng_key[1], rng_key[2]);
    V3 sf, swi;
    bxdf_sample_f(l, v3(wo[0], wo[1], wo[2]), u0, u1, rng, sf, swi, *pdf);
    f[0] = sf.x; f[1] = sf.y; f[2] = sf.z;
    wi[0] = swi.x; wi[1] = swi.y; wi[2] = swi.z;
}
// Light::Infinite of the scene (next-row f4).  what: 0 sample (in = u0,u1; out = uv0, uv1, map_pdf),
// 1 map pdf at (in = p0,p1), 2 le(dir), 3 pdf_li(dir), 4 table sizes (out = nu, nv, marg_int)
int oracle_env(const oracle_scene* s, int what, const double* in, double* out) {
    const Scene& sc = s->sc;
    if (sc.env_light < 0) return RT_ERR_STATE;
    const rt_light& lt = sc.lights[sc.env_light];
    switch (what) {
        case 0: dist2d_sample(sc.env_dist, in[0], in[1], out[0], out[1], out[2]); break;
        case 1: out[0] = dist2d_pdf(sc.env_dist, in[0], in[1]); break;
        case 2: {
            V3 c = light_le(sc, lt, v3(in[0], in[1], in[2]));
            out[0] = c.x; out[1] = c.y; out[2] = c.z;
            break;
        }
        case 3: out[0] = infinite_pdf_li(sc, lt, v3(in[0], in[1], in[2])); break;
        default:
            out[0] = (double)sc.env_dist.p_cond_v[0].n;
            out[1] = (double)sc.env_dist.p_marginal.n;
            out[2] = sc.env_dist.p_marginal.func_int;
    }
    return RT_OK;
}
double oracle_prim_area(const oracle_scene* s, int32_t prim) { return prim_area(s->sc, s->sc.prims[prim]); }
double oracle_prim_pdf(const oracle_scene* s, int32_t prim, const double* p, const double* dir) {
    Hit rec{};
    rec.p = v3(p[0], p[1], p[2]);
    return prim_pdf(s->sc, s->sc.prims[prim], rec, v3(dir[0], dir[1], dir[2]));
}
void oracle_texture_value(const oracle_scene* s, uint32_t tex, double u, double v, double* out) {
    V3 r = texture_value(s->sc, tex, u, v);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
// detmath array helper: fn 0 sin,1 cos,2 log,3 acos,4 atan2(x,y),5 exp,6 pow(x,y),7 sqrt
void oracle_detmath(int fn, const double* x, const double* y, uint64_t n, double* out) {
    for (uint64_t i = 0; i < n; i++) {
        switch (fn) {
            case 0: out[i] = dm_sin(x[i]); break;
            case 1: out[i] = dm_cos(x[i]); break;
            case 2: out[i] = dm_log(x[i]); break;
            case 3: out[i] = dm_acos(x[i]); break;
            case 4: out[i] = dm_atan2(x[i], y[i]); break;
            case 5: out[i] = dm_exp(x[i]); break;
            case 6: out[i] = dm_pow(x[i], y[i]); break;
            default: out[i] = dm_sqrt(x[i]); break;
        }
    }
}
// util.rs:400-408, 441-471 point_to_color (tone-map row f1)
void oracle_resolve_rgb8(const double* rgb_sum, const uint32_t* n, uint64_t npix, uint8_t* out) {
    for (uint64_t i = 0; i < npix; i++) {
        double scale = 1.0 / (double)n[i];
        for (int c = 0; c < 3; c++) {
            double x = rgb_sum[i * 3 + c] * scale;
            x = x * 0.6;  // aces_comp
            x = clampd((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14), 0.0, 1.0);
            double v = std::round(dm_pow(x, 1.0 / 2.2) * 256.0);
            // Rust `as u8` saturates
            out[i * 3 + c] = (uint8_t)(v != v ? 0 : (v < 0.0 ? 0 : (v > 255.0 ? 255 : v)));
        }
    }
}

}  // extern "C"

// geom.h -- ray/primitive tests and the closest-hit BVH traversal for gfx950.
//
// Replaces, with identical results (tests/test_gpu_parity.py::test_intersect_batch_*, tests/test_gpu_arms.py):
//   BoundingBox::intersects        src/hittable.rs:494-508   -> slab()
//   Mesh::intersects_triangle      src/hittable.rs:292-452   -> tri_core() + tri_record()
//   sphere_intersect / record      src/intersects.rs:177-258 -> sphere_core() + sphere_record()
//   {xy,xz,yz}_rect_intersect      src/intersects.rs:10-175  -> rect_core() + rect_record()
//   Primitive::intersects          src/primitive.rs:247-316  -> prim_intersects()
//   BvhNode::intersects            src/hittable.rs:591-634   -> closest_hit()
//
// The reference's traversal is exhaustive (both children always visited, no tmax
// shrinking), so its result is "closest hit over every primitive whose own AABB
// passes the slab test on [tmin, 1e308]".  closest_hit() returns exactly that with
// an ordered, pruned walk over our own BVH: the f64 slab test is monotone in the
// box, so a parent box (a superset, rounded outward to f32) passes whenever the
// leaf's f64 box does; pruning uses [tmin, max(best_t, tmin)*(1+1e-9)] so rounding between
// the box entry and the primitive's own t can never drop a closer hit (prune_limit()).  The tree is
// 4-wide (scene_dev.h); the order in which equally valid subtrees are visited does
// not matter for the result (closest t, ties by primitive index).
#pragma once
#include "dvec.h"
#include "scene_dev.h"

namespace rtd {

struct HitRec {  // hittable.rs:50-72, the fields the path reads
    double t;
    D3 n, p;
    bool front;
    double u, v;
    uint32_t mat;
    int32_t prim;
    int32_t light;  // light_index of the primitive
    D3 sh_n, sh_dpdu;
    D3 wo;
};

// util.rs:567-576
RTD void make_coordinate_system(D3 v1, D3& v2, D3& v3) {
    if (absd(v1.x) > absd(v1.y))
        v2 = d3(-v1.z, 0.0, v1.x) * (1.0 / dm_sqrt(v1.x * v1.x + v1.z * v1.z));
    else
        v2 = d3(0.0, v1.z, -v1.y) * (1.0 / dm_sqrt(v1.y * v1.y + v1.z * v1.z));
    v3 = cross(v1, v2);
}
RTD D3 face_forward(D3 n, D3 v) { return dot(n, v) < 0.0 ? -n : n; }  // util.rs:578-581

RTD D3 xf_point(const f64_t* m, D3 p) {
    return d3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
RTD D3 xf_vector(const f64_t* m, D3 v) {
    return d3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
              m[8] * v.x + m[9] * v.y + m[10] * v.z);
}

// hittable.rs:494-508 with 1/dir hoisted per ray (same IEEE value as the per-node divide).
// Returns pass/fail; `entry` = final tmin (the box entry parameter).
// hmin/hmax: v_min_f64 / v_max_f64 (IEEE minNum/maxNum: the non-NaN operand wins, like f64::min).
// They can differ from rmin/rmax only in the sign of a zero result, which the slab test never observes
// (its outputs are only compared), so the pass/fail decision and the ordering are unchanged.
RTD double hmin(double a, double b) { return __builtin_fmin(a, b); }
RTD double hmax(double a, double b) { return __builtin_fmax(a, b); }
RTD bool slab(double b0x, double b0y, double b0z, double b1x, double b1y, double b1z, D3 o, D3 inv, double tmin,
              double tmax, double& entry) {
    // The reference rejects after each axis (tmax <= tmin); tmin only grows and tmax only shrinks, so one
    // final comparison decides the same thing without three divergent exits.
    double v1 = (b0x - o.x) * inv.x, v2 = (b1x - o.x) * inv.x;
    tmin = hmax(tmin, hmin(v1, v2));
    tmax = hmin(tmax, hmax(v1, v2));
    v1 = (b0y - o.y) * inv.y;
    v2 = (b1y - o.y) * inv.y;
    tmin = hmax(tmin, hmin(v1, v2));
    tmax = hmin(tmax, hmax(v1, v2));
    v1 = (b0z - o.z) * inv.z;
    v2 = (b1z - o.z) * inv.z;
    tmin = hmax(tmin, hmin(v1, v2));
    tmax = hmin(tmax, hmax(v1, v2));
    entry = tmin;
    return !(tmax <= tmin);
}

// ------------------------------------------------------------------ triangle
// hittable.rs:300-362: sheared edge-function test (Q4, Q6).  tmin is ignored by the reference.
// Per-ray constants of the triangle test (the reference recomputes them per triangle from the same
// inputs, hittable.rs:313-323, so hoisting them is value-identical).
struct TriRay {
    int kz;
    double s_x, s_y, s_z;
    double dir_x, dir_y, dir_z;  // the ray direction itself (sphere / rect tests)
};
RTD TriRay tri_ray(D3 dir) {
    TriRay r;
    double ax = absd(dir.x), ay = absd(dir.y), az = absd(dir.z);
    r.kz = 0;
    double best = ax;
    if (ay > best) { best = ay; r.kz = 1; }
    if (az > best) { best = az; r.kz = 2; }
    const int kx = (r.kz + 1) % 3, ky = (kx + 1) % 3;
    D3 d = d3(comp(dir, kx), comp(dir, ky), comp(dir, r.kz));
    r.dir_x = dir.x;
    r.dir_y = dir.y;
    r.dir_z = dir.z;
    r.s_x = -d.x / d.z;
    r.s_y = -d.y / d.z;
    r.s_z = 1.0 / d.z;
    return r;
}
RTD D3 permute(D3 v, const TriRay& r) {  // util.rs:195-201 with (kx,ky,kz) a cyclic shift of (0,1,2)
    if (r.kz == 2) return v;
    if (r.kz == 0) return d3(v.y, v.z, v.x);
    return d3(v.z, v.x, v.y);
}
RTD D3 unpermute(D3 v, int kz) {  // inverse of permute()
    if (kz == 2) return v;
    if (kz == 0) return d3(v.z, v.x, v.y);
    return d3(v.y, v.z, v.x);
}
// the test on vertices already translated to the ray origin and permuted (p - o is componentwise, so it
// commutes with the permutation)
RTD bool tri_core_t(D3 p0t, D3 p1t, D3 p2t, const TriRay& tr, double tmax, double& t, double& b0, double& b1,
                    double& b2);
RTD bool tri_core(D3 p0, D3 p1, D3 p2, D3 o, const TriRay& tr, double tmax, double& t, double& b0, double& b1,
                  double& b2) {
    return tri_core_t(permute(p0 - o, tr), permute(p1 - o, tr), permute(p2 - o, tr), tr, tmax, t, b0, b1, b2);
}
RTD bool tri_core_t(D3 p0t, D3 p1t, D3 p2t, const TriRay& tr, double tmax, double& t, double& b0, double& b1,
                    double& b2) {
    const double s_x = tr.s_x, s_y = tr.s_y, s_z = tr.s_z;
    p0t.x += s_x * p0t.z; p0t.y += s_y * p0t.z;
    p1t.x += s_x * p1t.z; p1t.y += s_y * p1t.z;
    p2t.x += s_x * p2t.z; p2t.y += s_y * p2t.z;
    double e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    double e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    double e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if ((e0 < 0.0 || e1 < 0.0 || e2 < 0.0) && (e0 > 0.0 || e1 > 0.0 || e2 > 0.0)) return false;
    double det = e0 + e1 + e2;
    if (absd(det) < kSmall / 10000.0) return false;
    p0t.z *= s_z; p1t.z *= s_z; p2t.z *= s_z;
    double t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0.0 && (t_scaled >= 0.0 || t_scaled < tmax * det))
        return false;
    else if (det > 0.0 && (t_scaled <= 0.0 || t_scaled > tmax * det))
        return false;
    double inv_det = 1.0 / det;
    b0 = e0 * inv_det;
    b1 = e1 * inv_det;
    b2 = e2 * inv_det;
    t = t_scaled * inv_det;
    if (t < kSmall / 10.0) return false;
    return true;
}

struct TriUv {
    double u0, v0, u1, v1, u2, v2;
};
RTD TriUv tri_uvs(const DevMesh& m, uint32_t i1, uint32_t i2, uint32_t i3) {  // hittable.rs:454-468
    TriUv r{0.0, 0.0, 1.0, 0.0, 1.0, 1.0};
    if (m.uv) {
        r.u0 = m.uv[2 * i1]; r.v0 = m.uv[2 * i1 + 1];
        r.u1 = m.uv[2 * i2]; r.v1 = m.uv[2 * i2 + 1];
        r.u2 = m.uv[2 * i3]; r.v2 = m.uv[2 * i3 + 1];
    }
    return r;
}
// hittable.rs:367-386: the only rejection after t is accepted (degenerate triangle
// under a degenerate uv map).  Also yields dpdu/dpdv.
RTD bool tri_dpdu(D3 p0, D3 p1, D3 p2, const TriUv& uv, D3& dpdu, D3& dpdv) {
    double duv02x = uv.u0 - uv.u2, duv02y = uv.v0 - uv.v2;
    double duv12x = uv.u1 - uv.u2, duv12y = uv.v1 - uv.v2;
    D3 dp02 = p0 - p2, dp12 = p1 - p2;
    double determinant = duv02x * duv12y - duv02y * duv12x;
    if (absd(determinant) < kSmall / 10000.0) {
        D3 n = cross(p2 - p0, p1 - p0);
        if (norm2(n) == 0.0) return false;
        make_coordinate_system(n, dpdu, dpdv);
    } else {
        double invd = 1.0 / determinant;
        dpdu = (duv12y * dp02 - duv02y * dp12) * invd;
        dpdv = (-duv12x * dp02 + duv02x * dp12) * invd;
    }
    return true;
}

RTD void load_tri(const DevScene& sc, const rt_primitive& pr, D3& p0, D3& p1, D3& p2, uint32_t& i1, uint32_t& i2,
                  uint32_t& i3) {
    const DevMesh& m = sc.meshes[pr.mesh_index];
    i1 = m.ind[pr.tri_ind];
    i2 = m.ind[pr.tri_ind + 1];
    i3 = m.ind[pr.tri_ind + 2];
    p0 = d3(m.p[3 * i1], m.p[3 * i1 + 1], m.p[3 * i1 + 2]);
    p1 = d3(m.p[3 * i2], m.p[3 * i2 + 1], m.p[3 * i2 + 2]);
    p2 = d3(m.p[3 * i3], m.p[3 * i3 + 1], m.p[3 * i3 + 2]);
}

// HitRecord::new + set_front pieces (hittable.rs:75-117, 186-189)
RTD void hit_new(HitRec& h, D3 p, double u, double v, D3 wo, D3 dpdu, D3 dpdv, double t, uint32_t mat) {
    D3 n = normalize(cross(dpdu, dpdv));
    h.p = p;
    h.n = n;
    h.t = t;
    h.front = false;
    h.u = u;
    h.v = v;
    h.mat = mat;
    h.wo = wo;
    h.sh_n = n;
    h.sh_dpdu = normalize(dpdu);
    h.prim = 0;
}
RTD void set_front(HitRec& h, D3 dir) {
    h.front = dot(dir, h.n) < 0.0;
    if (!h.front) h.n = -h.n;
}

// hittable.rs:363-451: the differential-geometry block, run for the winning hit only.
RTD bool tri_record_core(D3 p0, D3 p1, D3 p2, const TriUv& uv, bool has_n, D3 n1, D3 n2, D3 n3, uint32_t mat, D3 o,
                         D3 dir, double tmax, HitRec& h) {
    double t, b0, b1, b2;
    if (!tri_core(p0, p1, p2, o, tri_ray(dir), tmax, t, b0, b1, b2)) return false;
    D3 dpdu, dpdv;
    if (!tri_dpdu(p0, p1, p2, uv, dpdu, dpdv)) return false;
    D3 dp02 = p0 - p2, dp12 = p1 - p2;
    D3 p_hit = b0 * p0 + b1 * p1 + b2 * p2;
    double u_hit = b0 * uv.u0 + b1 * uv.u1 + b2 * uv.u2;
    double v_hit = b0 * uv.v0 + b1 * uv.v1 + b2 * uv.v2;
    D3 normal;
    if (!has_n)
        normal = cross(dp02, dp12);
    else
        normal = b0 * n1 + b1 * n2 + b2 * n3;
    hit_new(h, p_hit, u_hit, v_hit, -dir, dpdu, dpdv, t, mat);
    h.n = normalize(cross(dp02, dp12));
    h.sh_n = normalize(normal);
    D3 ss = normalize(dpdu);
    D3 ts = normalize(cross(h.sh_n, ss));
    if (norm2(ts) > 0.0) {
        ss = normalize(cross(ts, h.sh_n));
    } else {
        D3 a, b;
        make_coordinate_system(h.sh_n, a, b);
        ss = normalize(a);
        ts = normalize(b);
    }
    D3 n = normalize(cross(ss, ts));  // set_shading_geometry(.., is_auth = true)
    h.sh_n = n;
    h.n = face_forward(h.n, h.sh_n);
    h.sh_dpdu = ss;
    set_front(h, dir);
    h.u = u_hit;
    h.v = v_hit;
    return true;
}
RTD bool tri_record(const DevScene& sc, const rt_primitive& pr, D3 o, D3 dir, double tmax, HitRec& h) {
    D3 p0, p1, p2;
    uint32_t i1, i2, i3;
    load_tri(sc, pr, p0, p1, p2, i1, i2, i3);
    const DevMesh& m = sc.meshes[pr.mesh_index];
    const TriUv uv = tri_uvs(m, i1, i2, i3);
    D3 n1 = d3(0, 0, 0), n2 = n1, n3 = n1;
    if (m.n) {
        n1 = d3(m.n[3 * i1], m.n[3 * i1 + 1], m.n[3 * i1 + 2]);
        n2 = d3(m.n[3 * i2], m.n[3 * i2 + 1], m.n[3 * i2 + 2]);
        n3 = d3(m.n[3 * i3], m.n[3 * i3 + 1], m.n[3 * i3 + 2]);
    }
    return tri_record_core(p0, p1, p2, uv, m.n != nullptr, n1, n2, n3, pr.mat_index, o, dir, tmax, h);
}
// The same record from the triangle's leaf slot (meshes without uvs: default uvs of hittable.rs:455-460)
RTD bool tri_record_slot(const DevScene& sc, uint32_t ls, int32_t pi, D3 o, D3 dir, double tmax, HitRec& h) {
#ifdef RT_F32
    const float* tp = sc.leaf_tri32 + (size_t)ls * 9;  // RT_KEEP_F64 (the line is already in its fast-mode form)
#else
    const f64_t* tp = sc.leaf_tri + (size_t)ls * 9;
#endif
    const LeafMeta meta = sc.leaf_meta[ls];
    const D3 p0 = d3(tp[0], tp[1], tp[2]), p1 = d3(tp[3], tp[4], tp[5]), p2 = d3(tp[6], tp[7], tp[8]);
    const bool has_n = (meta.mat_flags & kMetaHasNormals) != 0u;
    D3 n1 = d3(0, 0, 0), n2 = n1, n3 = n1;
    if (has_n) {
#ifdef RT_F32
        const float* np = sc.leaf_nrm32 + (size_t)ls * 9;  // RT_KEEP_F64
#else
        const f64_t* np = sc.leaf_nrm + (size_t)ls * 9;
#endif
        n1 = d3(np[0], np[1], np[2]);
        n2 = d3(np[3], np[4], np[5]);
        n3 = d3(np[6], np[7], np[8]);
    }
    const TriUv uv{0.0, 0.0, 1.0, 0.0, 1.0, 1.0};
    if (!tri_record_core(p0, p1, p2, uv, has_n, n1, n2, n3, meta.mat_flags & kMetaMatMask, o, dir, tmax, h)) return false;
    if (meta.mat_flags & kMetaFlip) h.front = !h.front;
    h.prim = pi;
    h.light = meta.light;
    return true;
}

// -------------------------------------------------------------------- rects
// intersects.rs:10-175.  Outputs t and the in-plane coordinates.  Takes the raw parameters
// (kind, v = a0,b0,a1,b1,k, transform index) so that the traversal can feed it from the leaf slot.
RTD bool rect_core_v(const DevScene& sc, uint32_t kind, double a0, double b0, double a1, double b1, double k,
                     int32_t xform_index, D3 o, D3 dir, double t0, double t1, double& t, double& a, double& b, D3& to,
                     D3& td) {
    to = o;
    td = dir;
    if (xform_index >= 0) {  // Ray::transform (geometry.rs:231-235) with the stored inverse
        const rt_xform& xf = sc.xforms[xform_index];
        td = xf_vector(xf.inv, dir);
        to = xf_point(xf.inv, o);
    }
    if (kind == RT_PRIM_XY_RECT) {
        t = (k - to.z) / td.z;
        if (t < t0 || t > t1) return false;
        a = to.x + t * td.x;
        b = to.y + t * td.y;
    } else if (kind == RT_PRIM_XZ_RECT) {
        t = (k - to.y) / td.y;
        if (t < t0 || t > t1) return false;
        a = to.x + t * td.x;
        b = to.z + t * td.z;
    } else {
        t = (k - to.x) / td.x;
        if (t < t0 || t > t1) return false;
        a = to.y + t * td.y;
        b = to.z + t * td.z;
    }
    if (a < a0 || b < b0 || a > a1 || b > b1) return false;
#ifdef RT_F32
    if (t != t || a != a || b != b) return false;  // fast mode: a ray in the rect's plane (0 / 0) is a miss, not a NaN hit
#endif
    return true;
}
RTD bool rect_core(const DevScene& sc, const rt_primitive& pr, D3 o, D3 dir, double t0, double t1, double& t,
                   double& a, double& b, D3& to, D3& td) {
    return rect_core_v(sc, pr.kind, pr.v[0], pr.v[1], pr.v[2], pr.v[3], pr.v[4], pr.xform_index, o, dir, t0, t1, t, a,
                       b, to, td);
}
RTD bool rect_record(const DevScene& sc, const rt_primitive& pr, D3 o, D3 dir, double t0, double t1, HitRec& h) {
    double t, a, b;
    D3 to, td;
    if (!rect_core(sc, pr, o, dir, t0, t1, t, a, b, to, td)) return false;
    D3 dpdu, dpdv;
    if (pr.kind == RT_PRIM_XY_RECT) {
        dpdu = d3(1, 0, 0);
        dpdv = d3(0, 1, 0);
    } else if (pr.kind == RT_PRIM_XZ_RECT) {
        dpdu = d3(1, 0, 0);
        dpdv = d3(0, 0, 1);
    } else {
        dpdu = d3(0, 1, 0);
        dpdv = d3(0, 0, 1);
    }
    double u = (a - pr.v[0]) / (pr.v[2] - pr.v[0]), v = (b - pr.v[1]) / (pr.v[3] - pr.v[1]);
    D3 p = to + td * t;
    if (pr.xform_index >= 0) {
        const rt_xform& xf = sc.xforms[pr.xform_index];
        p = xf_point(xf.fwd, p);
        dpdu = xf_vector(xf.fwd, dpdu);
        dpdv = xf_vector(xf.fwd, dpdv);
        hit_new(h, p, u, v, -dir, dpdu, dpdv, t, pr.mat_index);
    } else {
        // An axis-aligned rect: hit_new's normalize(cross(dpdu, dpdv)) and normalize(dpdu) are exact -- the cross
        // product of two unit axes is (+0, +0, 1), (+0, -1, +0) or (1, +0, +0) (xy / xz / yz, signed zeros as the
        // formula yields them), its length is 1 and x / 1 = x -- so the record is written directly: the same bits
        // without two square roots and six divisions per wall hit.
        h.p = p;
        h.n = pr.kind == RT_PRIM_XY_RECT ? d3(0.0, 0.0, 1.0) : (pr.kind == RT_PRIM_XZ_RECT ? d3(0.0, -1.0, 0.0) : d3(1.0, 0.0, 0.0));
        h.t = t;
        h.front = false;
        h.u = u;
        h.v = v;
        h.mat = pr.mat_index;
        h.wo = -dir;
        h.sh_n = h.n;
        h.sh_dpdu = dpdu;
        h.prim = 0;
    }
    set_front(h, dir);
    return true;
}
// Which component of a direction an axis-aligned rect's plane normal picks: xy -> z, xz -> y, yz -> x.
RTD double rect_axis_comp(uint32_t kind, D3 d) { return kind == RT_PRIM_XY_RECT ? d.z : (kind == RT_PRIM_XZ_RECT ? d.y : d.x); }

// ------------------------------------------------------------------- sphere
// intersects.rs:177-213
RTD bool sphere_core_v(D3 center, double r, D3 o, D3 dir, double tmin, double tmax, double& t) {
    D3 diff = o - center;
    double a = dot(dir, dir);
    double b = dot(diff, dir);
    double c = dot(diff, diff) - r * r;
    double disc = b * b - a * c;
    if (disc < 0.0) return false;
    double inv_a = 1.0 / a;
    double root = dm_sqrt(disc);
    double ans = (-b - root) * inv_a;
    if (ans < tmax && ans > tmin) {
        t = ans;
        return true;
    }
    ans = (-b + root) * inv_a;
    if (ans < tmax && ans > tmin) {
        t = ans;
        return true;
    }
    return false;
}
RTD bool sphere_core(const rt_primitive& pr, D3 o, D3 dir, double tmin, double tmax, double& t) {
    return sphere_core_v(d3(pr.v[0], pr.v[1], pr.v[2]), pr.v[3], o, dir, tmin, tmax, t);
}
// intersects.rs:216-258 (Q8)
RTD bool sphere_record(const rt_primitive& pr, D3 o, D3 dir, double tmin, double tmax, HitRec& h) {
    double t;
    if (!sphere_core(pr, o, dir, tmin, tmax, t)) return false;
    D3 center = d3(pr.v[0], pr.v[1], pr.v[2]);
    double r = pr.v[3];
    D3 to = o - center;
    D3 p = to + dir * t;
    p = p * r / norm(p);
    if (p.x == 0.0 && p.y == 0.0) p.x = 1e-5 * r;
    double phi = dm_atan2(p.y, p.x);
    if (phi < 0.0) phi = phi + 2.0 * kPi;
    double phi_max = 2.0 * kPi;
    double theta_min = 0.0, theta_max = kPi;
    double u = phi / phi_max;
    double theta = dm_acos(clampd(p.z / r, -1.0, 1.0));
    double v = (theta - theta_min) / (theta_max - theta_min);
    double z_r = dm_sqrt(p.x * p.x + p.y * p.y);
    double inv_z_r = 1.0 / z_r;
    double cos_phi = p.x * inv_z_r;
    double sin_phi = p.y * inv_z_r;
    D3 dpdu = d3(-phi_max * p.y, phi_max * p.x, 0.0);
    D3 dpdv = (theta_max - theta_min) * d3(p.z * cos_phi, p.z * sin_phi, -r * dm_sin(theta));
    hit_new(h, p, u, v, -dir, dpdu, dpdv, t, pr.mat_index);
    set_front(h, dir);
    h.p = h.p + center;
    return true;
}

// primitive.rs:372-425 intersects_obj: full record of ONE primitive, no prim_index stamp
RTD bool intersects_obj(const DevScene& sc, const rt_primitive& pr, D3 o, D3 dir, double tmin, double tmax,
                        HitRec& h) {
    if (pr.kind == RT_PRIM_TRIANGLE) return tri_record(sc, pr, o, dir, tmax, h);
    if (pr.kind == RT_PRIM_SPHERE) return sphere_record(pr, o, dir, tmin, tmax, h);
    return rect_record(sc, pr, o, dir, tmin, tmax, h);
}
// primitive.rs:247-316
RTD bool prim_intersects(const DevScene& sc, int32_t index, D3 o, D3 dir, double tmin, double tmax, HitRec& h) {
    const rt_primitive& pr = sc.prims[index];
    if (!intersects_obj(sc, pr, o, dir, tmin, tmax, h)) return false;
    if (pr.flip) h.front = !h.front;
    h.prim = index;
    h.light = pr.light_index;
    return true;
}
// The extension ray's result as ONE word (scene_dev.h: the hit word of a list entry): class bits and kLeafOther of the
// leaf, and what the winner's record is rebuilt from -- the LEAF SLOT of a mesh hit, the PRIMITIVE INDEX of a sphere / rect.
RTD uint32_t hit_word(int32_t prim, uint32_t best_slot) {
    return (best_slot & kLeafOther) ? ((uint32_t)prim | (best_slot & ~kIdxMask)) : best_slot;
}
// Record of the extension ray's winner.  KIND (scene_dev.h: kKind*) = what the caller knows about its hits; a mesh whose
// record needs the primitive (texture coordinates) finds it through the leaf slot.
template <int KIND>
RTD bool hit_record(const DevScene& sc, uint32_t hit, D3 o, D3 dir, double tmin, double tmax, HitRec& h) {
    const uint32_t idx = hit & kIdxMask;
    if (KIND == kKindMesh) return tri_record_slot(sc, idx, 0, o, dir, tmax, h);
    if (KIND == kKindOther || (hit & kLeafOther)) return prim_intersects(sc, (int32_t)idx, o, dir, tmin, tmax, h);
    if (!sc.mesh_has_uv) return tri_record_slot(sc, idx, 0, o, dir, tmax, h);
    return prim_intersects(sc, (int32_t)(sc.leaf_prim[idx] & kIdxMask), o, dir, tmin, tmax, h);
}

// ---------------------------------------------------------------- traversal
RTD float float_lower(double t) {  // largest float <= t, for t >= 0
    float f = (float)t;
    if ((double)f > t) f = __uint_as_float(__float_as_uint(f) - 1u);
    return f;
}

struct TravCount {
    uint32_t nodes, tris, others;
};

// Per-lane traversal stack: the first kLdsStack entries live in LDS ([entry][thread] layout, one
// 8-byte {node, entry_t} pair per lane -> conflict-free ds_read/write_b64), deeper entries spill to
// a small private array that only keeps the node (popped unconditionally; its children are then
// culled by their own slab tests).  Keeping the private part under ~256 B per lane matters: a larger
// scratch frame makes every dispatch of the kernel pay a use-once scratch allocation (~0.1 ms).
// Round 3: five waves per SIMD for k_trace (96 VGPRs, a handful spilled) with a 12-entry LDS stack (24 KB per block, five
// blocks per CU): C4 k_trace 577 -> 561 ms, C3 105 -> 100.  With 16 entries five blocks do not fit the CU's LDS.
#ifndef RT_LDS_STACK
#define RT_LDS_STACK 12
#endif
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 5
#endif
constexpr int kLdsStack = RT_LDS_STACK;
constexpr int kOvfStack = 3 * kMaxBvhDepth + 2 - kLdsStack;  // a 4-wide node pushes up to three children
// BVH-node tiles staged in LDS (BASELINE north_star): every block of k_trace copies the first RT_LDS_NODES nodes -- the
// top of the tree, which rt_scene_commit lays out breadth-first (levels 0-2 are 21 nodes) and which every ray walks --
// into LDS once; node_step reads them with ds_read and everything else with global loads.  Round 2 measured a version
// that went through a generic pointer (flat loads) as 11-16 % slower; that measurement was dominated by the traversal
// stack's scratch / flat traffic (fixed in round 3).  Now: C4 k_trace -1.0 %, C3 +-0, C2 -0.3 % (profiles/
// r03_exp_lds_nodes_two_path*.txt) -- the kernel is bound by VALU issue, not by the node fetches of its top levels.
#ifndef RT_LDS_NODES
#define RT_LDS_NODES 32
#endif
// (Round 3) The struct holds two POINTERS and nothing else, so that it lives in registers: with the overflow array inside it
// the whole struct sat in scratch, and every push / pop first re-read the LDS base and the stride from scratch
// (scratch_load + s_waitcnt vmcnt(0) + a quarter-rate v_mul_lo_u32 per stack access), and the pop loop read its entries
// through a generic pointer (flat_load + vmcnt(0) lgkmcnt(0)).  The stride is the block size of every traversing kernel.
constexpr int kStackStride = 256;
struct TravStack {
    int2* lds;       // &lds_stack[threadIdx.x]; entry i at lds[i * kStackStride]
    int32_t* ovf;    // the kernel's private overflow array, kOvfStack entries
#if RT_LDS_NODES > 0
    // BVH-node tiles staged in LDS (north_star): the block's copy of nodes[0, n_top), the top of the tree in breadth-first
    // order (k_trace only; n_top = 0 elsewhere).  top_lds = its byte address in LDS.
    uint32_t top_lds = 0;
    uint32_t n_top = 0;
#endif
};
typedef const __attribute__((address_space(3))) char* LdsBytes;
typedef float rt_v4f __attribute__((ext_vector_type(4)));
typedef int rt_v4i __attribute__((ext_vector_type(4)));
RTD float4 lds_ld4f(LdsBytes p) {
    const rt_v4f v = *reinterpret_cast<const __attribute__((address_space(3))) rt_v4f*>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
RTD int4 lds_ld4i(LdsBytes p) {
    const rt_v4i v = *reinterpret_cast<const __attribute__((address_space(3))) rt_v4i*>(p);
    return make_int4(v.x, v.y, v.z, v.w);
}
#define RT_TRAV_STACK(ts, lds_array)            \
    int32_t ts##_ovf_store[kOvfStack];          \
    TravStack ts;                               \
    ts.lds = &(lds_array)[threadIdx.x];         \
    ts.ovf = ts##_ovf_store;
RTD void stack_push(TravStack& ts, int sp, int32_t node, float t) {
    if (sp < kLdsStack)
        ts.lds[sp * kStackStride] = make_int2(node, __float_as_int(t));
    else
        ts.ovf[sp - kLdsStack] = node;
}
RTD void stack_get(const TravStack& ts, int sp, int32_t& node, float& t) {
    // The LDS read is unconditional (index clamped), the rare overflow entry overrides it: two loads of different address
    // spaces in the two arms of one `if` were merged into a flat load by the compiler.
    const int2 v = ts.lds[(sp < kLdsStack ? sp : kLdsStack - 1) * kStackStride];
    node = v.x;
    t = __int_as_float(v.y);
    asm volatile("" : "+v"(node));  // (opaque copy: keeps this a ds_read -- see above)
    if (sp >= kLdsStack) {
        node = ts.ovf[sp - kLdsStack];
        asm volatile("" : "+v"(node));  // the scratch load is awaited HERE, in the rare branch, not by an s_waitcnt vmcnt(0)
        t = 0.0f;                       // at the merge point that every pop would pay (stores may be in flight)
    }
}

// Traversal state of one ray.  It advances in two kinds of steps so that a wave can run tight
// node-only rounds and then primitive-only rounds (while-while traversal) and refill finished lanes
// in between (k_trace), or simply loop to completion (closest_hit):
//   cur >= 0            at an internal node          -> node_step()
//   cur <  0 (not done) at a leaf, its first remaining primitive next -> leaf_step()
struct Trav {
    D3 op, ip;  // origin and 1/dir, PERMUTED to the triangle test's (kx, ky, kz) axis order
    // f32 constants of the conservative interior-node test (node_step): t = fma(plane, inv32, c),
    // c = -(o * inv32) -/+ slack for the near / far plane
    float ix, iy, iz, cnx, cny, cnz, cfx, cfy, cfz;
    TriRay trr;
    double tmin, tmax, best_t;
    int32_t best_prim, cur;
    uint32_t best_slot;  // leaf slot of best_prim | the leaf_prim word's class bits and kLeafOther
    int sp;
    bool done;
};


// Interior-node boxes only cull: a primitive is gated by the f64 slab test of its OWN box in leaf_step
// (the reference's semantics, hittable.rs:625), and every ancestor box contains that box, so an ancestor
// test may be evaluated in any arithmetic as long as it never rejects an interval the f64 test of a
// contained box accepts.  node_step does it in f32 with one fma per plane,
//     t32 = fma(plane, inv32, c),   inv32 = f32(1/d),   c = f32(-(o * inv32) -/+ s)
// against the f64 value t = (plane - o) * (1/d):
//     |t32 - t| <= 2^-24 |t|   (inv32 vs 1/d, both terms use the same inv32)
//                + 2^-24 |o inv32|   (rounding c)  + 2^-24 |t32|   (the fma's single rounding)
// The absolute part is folded into c as s = 2^-21 |o inv32| (lower bound for the near plane, upper bound
// for the far plane); the relative part is applied after the 3-axis max/min as 2^-21 |t| (the axis that
// attains the max/min is the one whose error matters).  Both carry a >= 4x margin over the bound above
// and over the f64 roundings of the reference expression.  NaNs (0 * inf when a direction component is
// 0 or below f32 range) are dropped by min/max, which only widens the accepted interval.
constexpr float kNodeSlack = 4.76837158203125e-07f;  // 2^-21
RTD void node_consts(Trav& tv, D3 o, D3 inv) {
    tv.ix = (float)inv.x;
    tv.iy = (float)inv.y;
    tv.iz = (float)inv.z;
    const double ox = o.x * (double)tv.ix, oy = o.y * (double)tv.iy, oz = o.z * (double)tv.iz;
    const double sx = absd(ox) * (double)kNodeSlack, sy = absd(oy) * (double)kNodeSlack,
                 sz = absd(oz) * (double)kNodeSlack;
    tv.cnx = (float)(-ox - sx);
    tv.cny = (float)(-oy - sy);
    tv.cnz = (float)(-oz - sz);
    tv.cfx = (float)(-ox + sx);
    tv.cfy = (float)(-oy + sy);
    tv.cfz = (float)(-oz + sz);
}

RTD void trav_init(Trav& tv, const DevScene& sc, D3 o, D3 dir, double tmin, double tmax) {
#ifdef RT_F32
    // Fast mode only.  In binary32 a direction component is EXACTLY zero every few hundred thousand rays (e.g.
    // to.x - origin.x of the camera rays near the image's centre column cancels to 0): 1/0 = inf makes the slab
    // tests of that axis NaN, i.e. unconstrained (the reference's Q5 behaviour, practically unreachable in f64), and
    // such a ray then walks a whole slice of the tree -- thousands of steps that the launch's other waves wait
    // for.  The reciprocal is therefore taken of a component pushed away from zero (relative 1e-12).
    const double dmax = rmax(absd(dir.x), rmax(absd(dir.y), absd(dir.z))), deps = rmax(dmax * 1e-12, 1e-30);
    const D3 dsafe = d3(absd(dir.x) < deps ? __builtin_copysign(deps, dir.x) : dir.x,
                        absd(dir.y) < deps ? __builtin_copysign(deps, dir.y) : dir.y,
                        absd(dir.z) < deps ? __builtin_copysign(deps, dir.z) : dir.z);
    const D3 inv = d3(1.0 / dsafe.x, 1.0 / dsafe.y, 1.0 / dsafe.z);
#else
    const D3 inv = d3(1.0 / dir.x, 1.0 / dir.y, 1.0 / dir.z);
#endif
    tv.trr = tri_ray(dir);
    tv.op = permute(o, tv.trr);
    tv.ip = permute(inv, tv.trr);
    node_consts(tv, o, inv);
    tv.tmin = tmin;
    tv.tmax = tmax;
    tv.best_t = tmax;
    tv.best_prim = -1;
    tv.best_slot = 0;
    tv.cur = 0;
    tv.sp = 0;
    // A ray with a NaN component (Q17: NaNs are never filtered) passes every slab test and "hits" whatever
    // triangle is visited first, i.e. the reference's answer depends on its random tree.  The ABI pins it:
    // such a ray misses.
    tv.done = (o.x != o.x) || (o.y != o.y) || (o.z != o.z) || (dir.x != dir.x) || (dir.y != dir.y) || (dir.z != dir.z);
}

// Upper end of the interval a subtree's box must overlap to be worth visiting.  A triangle hit may lie
// BELOW tmin (the reference's triangle test ignores tmin and accepts t >= 1e-4, hittable.rs:360, while
// its boxes are tested on [tmin = 1e-3, ..), Q4), so the limit must never fall to tmin or below: the
// interval [tmin, limit] has to stay non-empty or every remaining subtree -- including one holding a
// still closer such hit -- would be culled.
RTD double prune_limit(const Trav& tv) {
    if (tv.best_prim < 0) return tv.tmax;
    return hmax(tv.best_t * (1.0 + 1e-9), tv.tmin * (1.0 + 1e-9) + 1e-300);
}

// pop the next subtree that can still contain a closer hit; marks the traversal done when none is left
RTD void trav_pop(Trav& tv, TravStack& ts) {
    const double lim2 = prune_limit(tv);
    while (tv.sp > 0) {
        tv.sp--;
        int32_t node;
        float et;
        stack_get(ts, tv.sp, node, et);
        if ((double)et <= lim2) {
            tv.cur = node;
                    return;
        }
    }
    tv.done = true;
}

// One internal node: test its four children (one 128-B fetch), descend into the nearest hit and push
// the others so that they pop nearest-first.
template <bool COUNT>
RTD void node_step(Trav& tv, const DevScene& sc, TravStack& ts, TravCount* tc) {
    // near / far plane rows of the node picked by the direction signs (rows: lo_x lo_y lo_z hi_x hi_y hi_z,
    // 16 B each), so no per-value select is needed
    const uint32_t kx = (__float_as_uint(tv.ix) >> 31) * 48u, ky = (__float_as_uint(tv.iy) >> 31) * 48u,
                   kz = (__float_as_uint(tv.iz) >> 31) * 48u;
    float4 nx, ny, nz, fx, fy, fz;
    int4 ch;
#if RT_LDS_NODES > 0
    // Two separately addressed paths -- ds_read for the lanes in the staged top of the tree, global loads for the
    // others -- into the same registers (a generic pointer would make all of them flat loads, which occupy both the
    // LDS and the vector-memory path and both wait counters).
    if ((uint32_t)tv.cur < ts.n_top) {
        const LdsBytes lb = (LdsBytes)(uintptr_t)(ts.top_lds + (uint32_t)tv.cur * 128u);
        nx = lds_ld4f(lb + kx);
        ny = lds_ld4f(lb + 16u + ky);
        nz = lds_ld4f(lb + 32u + kz);
        fx = lds_ld4f(lb + 48u - kx);
        fy = lds_ld4f(lb + 64u - ky);
        fz = lds_ld4f(lb + 80u - kz);
        ch = lds_ld4i(lb + 96u);
    } else
#endif
    {
        const char* nb = reinterpret_cast<const char*>(&sc.nodes[tv.cur]);
        nx = *reinterpret_cast<const float4*>(nb + kx);
        ny = *reinterpret_cast<const float4*>(nb + 16u + ky);
        nz = *reinterpret_cast<const float4*>(nb + 32u + kz);
        fx = *reinterpret_cast<const float4*>(nb + 48u - kx);
        fy = *reinterpret_cast<const float4*>(nb + 64u - ky);
        fz = *reinterpret_cast<const float4*>(nb + 80u - kz);
        ch = *reinterpret_cast<const int4*>(nb + 96u);
    }
    if (COUNT) tc->nodes++;
    const float tmin32 = float_lower(tv.tmin);
    const float tmax32 = (float)tv.tmax * (1.0f + kNodeSlack);
    const float lim32 = (float)prune_limit(tv) * (1.0f + kNodeSlack);
#define RT_NODE_CHILD(c, e, h, cid)                                                                    \
    float e;                                                                                           \
    bool h;                                                                                            \
    {                                                                                                  \
        const float m_ = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(nx.c, tv.ix, tv.cnx),          \
                                                         __builtin_fmaf(ny.c, tv.iy, tv.cny)),         \
                                         __builtin_fmaf(nz.c, tv.iz, tv.cnz));                         \
        const float f_ = __builtin_fminf(__builtin_fminf(__builtin_fmaf(fx.c, tv.ix, tv.cfx),          \
                                                         __builtin_fmaf(fy.c, tv.iy, tv.cfy)),         \
                                         __builtin_fmaf(fz.c, tv.iz, tv.cfz));                         \
        e = __builtin_fmaxf(__builtin_fmaf(-__builtin_fabsf(m_), kNodeSlack, m_), tmin32);             \
        /* (the min with tmax32 also turns an all-NaN far side -- a ray along an axis -- into "unbounded") */ \
        const float x_ = __builtin_fminf(__builtin_fmaf(__builtin_fabsf(f_), kNodeSlack, f_), tmax32); \
        h = e <= x_ && e <= lim32 && cid != kNoChild;                                                  \
    }
    RT_NODE_CHILD(x, e0, h0, ch.x)
    RT_NODE_CHILD(y, e1, h1, ch.y)
    RT_NODE_CHILD(z, e2, h2, ch.z)
    RT_NODE_CHILD(w, e3, h3, ch.w)
#undef RT_NODE_CHILD
    // sort the hit children by entry distance (misses sink to the end as +inf): 5-comparator network
    const float kMiss = __builtin_huge_valf();
    float d0 = h0 ? e0 : kMiss, d1 = h1 ? e1 : kMiss;
    float d2 = h2 ? e2 : kMiss, d3 = h3 ? e3 : kMiss;
    int32_t c0 = ch.x, c1 = ch.y, c2 = ch.z, c3 = ch.w;
#define RT_CSWAP(da, ca, db, cb)          \
    if (db < da) {                         \
        const float td_ = da; da = db; db = td_; \
        const int32_t tc_ = ca; ca = cb; cb = tc_; \
    }
    RT_CSWAP(d0, c0, d1, c1)
    RT_CSWAP(d2, c2, d3, c3)
    RT_CSWAP(d0, c0, d2, c2)
    RT_CSWAP(d1, c1, d3, c3)
    RT_CSWAP(d1, c1, d2, c2)
#undef RT_CSWAP
    const int nh = (int)h0 + (int)h1 + (int)h2 + (int)h3;
    if (nh == 0) {
        trav_pop(tv, ts);
        return;
    }
    // farthest first, so that the nearest pending child is on top of the stack
    if (nh > 3) { stack_push(ts, tv.sp, c3, d3); tv.sp++; }
    if (nh > 2) { stack_push(ts, tv.sp, c2, d2); tv.sp++; }
    if (nh > 1) { stack_push(ts, tv.sp, c1, d1); tv.sp++; }
    tv.cur = c0;
}

// One primitive of the current leaf (leaf code: -1 - ((first*8 + count-1) | kLeafCodeOther?)); pops after
// the last one.  Ties in t go to the larger prim index (ABI tie rule).
// SIMPLE: the scene has no sphere and no transformed rect (DevScene::simple_others), so the sphere / rect arm
// of a leaf is a plain axis-aligned rect test -- an instance of the traversal kernel without that code needs
// 11 VGPRs less (113 instead of 124) and is 2-3 % faster.
struct __attribute__((packed, aligned(8))) LeafNine {
    f64_t v[9];  // RT_KEEP_F64
};
template <bool COUNT, bool SIMPLE = false>
RTD void leaf_step(Trav& tv, const DevScene& sc, TravStack& ts, TravCount* tc) {
    const double tmin = tv.tmin, tmax = tv.tmax;
    const uint32_t code = (uint32_t)(-1 - tv.cur);
    const uint32_t first = (code & ~kLeafCodeOther) >> 3, count = (code & 7u) + 1u;
    const uint32_t slot = first;
#ifdef RT_F32
    const uint32_t e = sc.leaf_prim[slot];
    const float* tvp = sc.leaf_tri32 + (size_t)slot * 9;  // RT_KEEP_F64
#else
    // one aligned 128-B line per slot (scene_dev.h: leaf_trav): geometry and the primitive index word
    const f64_t* rec = sc.leaf_trav + (size_t)slot * 16;
    const uint32_t e = reinterpret_cast<const uint32_t*>(rec)[30];
#endif
    double t = 0.0;
    int32_t pi = -1;
    bool hit = false;
    if (!(code & kLeafCodeOther)) {
        // The nine coordinates are fetched in the ray's permuted axis order (kx, ky, kz), so the sheared
        // test needs no per-triangle shuffle; the addresses do not depend on `e` (loads issue together).
        const int kz = tv.trr.kz, kx = kz == 2 ? 0 : kz + 1, ky = kx == 2 ? 0 : kx + 1;
        const D3 op = tv.op, ip = tv.ip;
        if (COUNT) tc->tris++;
#ifdef RT_F32
        const float *px = tvp + kx, *py = tvp + ky, *pz = tvp + kz;  // RT_KEEP_F64
        const D3 p0t = d3(px[0] - op.x, py[0] - op.y, pz[0] - op.z);
        const D3 p1t = d3(px[3] - op.x, py[3] - op.y, pz[3] - op.z);
        const D3 p2t = d3(px[6] - op.x, py[6] - op.y, pz[6] - op.z);
#else
        // the 72 contiguous bytes from double 3 * kx: components kx, ky, kz of the three vertices (four 16-B loads
        // and an 8-B one on the slot's line; the window never leaves it)
        LeafNine w;
        __builtin_memcpy(&w, rec + kx * 3, sizeof(w));
        (void)ky;
        const D3 p0t = d3(w.v[0] - op.x, w.v[3] - op.y, w.v[6] - op.z);
        const D3 p1t = d3(w.v[1] - op.x, w.v[4] - op.y, w.v[7] - op.z);
        const D3 p2t = d3(w.v[2] - op.x, w.v[5] - op.y, w.v[8] - op.z);
#endif
        // The reference tests the leaf's own f64 box on the ORIGINAL interval (hittable.rs:625):
        // ((min_i p_i) - o) * inv.  fl(a - o) is monotone in a, so min/max commute with the subtraction
        // and the corner offsets are the min/max of the translated vertices (bit-identical); the axis order
        // of the three slabs does not matter (max/min of the same six products).  hmin/hmax: see slab().
        double lo = hmin(p0t.x, hmin(p1t.x, p2t.x)) * ip.x, hi = hmax(p0t.x, hmax(p1t.x, p2t.x)) * ip.x;
        double tn = hmax(tmin, hmin(lo, hi)), tf = hmin(tmax, hmax(lo, hi));
        lo = hmin(p0t.y, hmin(p1t.y, p2t.y)) * ip.y;
        hi = hmax(p0t.y, hmax(p1t.y, p2t.y)) * ip.y;
        tn = hmax(tn, hmin(lo, hi));
        tf = hmin(tf, hmax(lo, hi));
        lo = hmin(p0t.z, hmin(p1t.z, p2t.z)) * ip.z;
        hi = hmax(p0t.z, hmax(p1t.z, p2t.z)) * ip.z;
        tn = hmax(tn, hmin(lo, hi));
        tf = hmin(tf, hmax(lo, hi));
        double b0, b1, b2;
        TriRay trz = tv.trr;
        trz.s_z = ip.z;  // 1 / dir[kz]: the same value as tri_ray()'s s_z, so that need not stay in a register
        hit = !(tf <= tn) && tri_core_t(p0t, p1t, p2t, trz, tmax, t, b0, b1, b2);
        pi = (int32_t)(e & kIdxMask);
        if (hit && sc.mesh_has_uv) {  // rare path: uv-degenerate rejection (hittable.rs:373-378)
            const rt_primitive& pr = sc.prims[pi];
            const DevMesh& m = sc.meshes[pr.mesh_index];
            if (m.uv) {
#ifdef RT_F32
                const D3 p0 = d3(tvp[0], tvp[1], tvp[2]), p1 = d3(tvp[3], tvp[4], tvp[5]), p2 = d3(tvp[6], tvp[7], tvp[8]);
#else
                const D3 p0 = d3(rec[0], rec[3], rec[6]), p1 = d3(rec[1], rec[4], rec[7]), p2 = d3(rec[2], rec[5], rec[8]);
#endif
                TriUv uv = tri_uvs(m, m.ind[pr.tri_ind], m.ind[pr.tri_ind + 1], m.ind[pr.tri_ind + 2]);
                D3 du, dv;
                hit = tri_dpdu(p0, p1, p2, uv, du, dv);
            }
        }
    } else {
        // sphere / rect: the leaf slot carries v[0..4] and {kind, transform index} (abi.hip, commit), so an
        // untransformed rect or a sphere needs no second, dependent fetch.  Their own AABB
        // (Primitive::get_bounding_box) is re-derived with the constructor's arithmetic
        // (primitive.rs:66-68, 110-113, 161-164, 212-215): k -/+ SMALL, centre -/+ r.
        const D3 o = unpermute(tv.op, tv.trr.kz), inv = unpermute(tv.ip, tv.trr.kz);
        pi = (int32_t)(e & kIdxMask);
        if (COUNT) tc->others++;
#ifdef RT_F32
        const double v0 = tvp[0], v1 = tvp[1], v2 = tvp[2], v3 = tvp[3], v4 = tvp[4];
        const uint64_t meta = (uint64_t)__float_as_uint(tvp[5]) | ((uint64_t)__float_as_uint(tvp[6]) << 32);
#else
        const f64x2_t q0 = *reinterpret_cast<const f64x2_t*>(rec);
        const f64x2_t q1 = *reinterpret_cast<const f64x2_t*>(rec + 2);
        const f64x2_t q2 = *reinterpret_cast<const f64x2_t*>(rec + 4);
        const double v0 = q0.x, v1 = q0.y, v2 = q1.x, v3 = q1.y, v4 = q2.x;
        const uint64_t meta = dm_bits(q2.y);
#endif
        const uint32_t kind = (uint32_t)(meta & 0xffu);
        const int32_t xform_index = (int32_t)(uint32_t)(meta >> 32) - 1;
        double bx0, by0, bz0, bx1, by1, bz1;
        if (!SIMPLE && xform_index >= 0) {  // transformed rect: its box is util::get_new_box's (util.rs:493-517), stored
            const rt_primitive& pr = sc.prims[pi];
            bx0 = pr.bbox_min[0]; by0 = pr.bbox_min[1]; bz0 = pr.bbox_min[2];
            bx1 = pr.bbox_max[0]; by1 = pr.bbox_max[1]; bz1 = pr.bbox_max[2];
        } else if (!SIMPLE && kind == RT_PRIM_SPHERE) {
            bx0 = v0 - v3; by0 = v1 - v3; bz0 = v2 - v3;
            bx1 = v0 + v3; by1 = v1 + v3; bz1 = v2 + v3;
        } else {
            // xy: (v0, v1, k -/+ S), (v2, v3)   xz: (v0, k -/+ S, v1), (v2, .., v3)   yz: (k -/+ S, v0, v1), (.., v2, v3)
            // Written with selects, not as an if-chain over the three kinds: hipcc (ROCm 7.2) compiled the chain's
            // yz arm of the binary32 instance with by0 / bz0 left undefined (every yz rect was missed).
            const double klo = v4 - kSmall, khi = v4 + kSmall;
            const bool xy = kind == RT_PRIM_XY_RECT, xz = kind == RT_PRIM_XZ_RECT;
            bx0 = (xy || xz) ? v0 : klo;
            bx1 = (xy || xz) ? v2 : khi;
            by0 = xy ? v1 : (xz ? klo : v0);
            by1 = xy ? v3 : (xz ? khi : v2);
            bz0 = xy ? klo : v1;
            bz1 = xy ? khi : v3;
        }
        double en;
        if (slab(bx0, by0, bz0, bx1, by1, bz1, o, inv, tmin, tmax, en)) {
            const D3 dir = d3(tv.trr.dir_x, tv.trr.dir_y, tv.trr.dir_z);
            if (!SIMPLE && kind == RT_PRIM_SPHERE) {
                hit = sphere_core_v(d3(v0, v1, v2), v3, o, dir, tmin, tmax, t);
            } else {
                double a, b;
                D3 to, td;
                hit = rect_core_v(sc, kind, v0, v1, v2, v3, v4, SIMPLE ? -1 : xform_index, o, dir, tmin, tmax, t, a, b, to,
                                  td);
            }
        }
    }
    if (hit && (tv.best_prim < 0 || t < tv.best_t || (t == tv.best_t && pi > tv.best_prim))) {
        tv.best_t = t;
        tv.best_prim = pi;
        tv.best_slot = slot | (e & ~kIdxMask);  // + the leaf's vertex class and kLeafOther
    }
    // the rest of the leaf is the leaf (first + 1, count - 1): no separate cursor to keep in a register
    if (count > 1u)
        tv.cur = -1 - (int32_t)((((first + 1u) << 3) | (count - 2u)) | (code & kLeafCodeOther));
    else
        trav_pop(tv, ts);
}

// Closest hit of one ray, run to completion.  Returns prim index or -1; t_out = hit parameter.
template <bool COUNT>
RTD int32_t closest_hit(const DevScene& sc, D3 o, D3 dir, double tmin, double tmax, double& t_out, TravStack& ts,
                        TravCount* tc, uint32_t* slot_out = nullptr) {
    if (sc.n_nodes == 0) {
        t_out = tmax;
        return -1;
    }
    Trav tv;
    trav_init(tv, sc, o, dir, tmin, tmax);
    while (!tv.done) {
        if (tv.cur >= 0)
            node_step<COUNT>(tv, sc, ts, tc);
        else
            leaf_step<COUNT>(tv, sc, ts, tc);
    }
    t_out = tv.best_t;
    if (slot_out) *slot_out = tv.best_slot;
    return tv.best_prim;
}

}  // namespace rtd

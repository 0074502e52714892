// rr_host.cpp -- constructors of the host-side mirror (see rr_host.hpp).
// Compile with -ffp-contract=off: the values computed here are inputs of the
// numerical contract (both the GPU path and the CPU oracle consume them).
#include "rr_host.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../../include/rt_detmath.h"

namespace rr {

static const double PI = RT_PI;        // consts.rs:31
static const double SMALL = RT_SMALL;  // consts.rs:32
static const double INF = RT_INFINITY;

static inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
static inline Vec3 operator*(Vec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static inline Vec3 cross(Vec3 a, Vec3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline double norm(Vec3 a) { return dm_sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline Vec3 normalize(Vec3 a) {
    double n = norm(a);
    return {a.x / n, a.y / n, a.z / n};
}

// ------------------------------------------------------------------ Mat4
Mat4 Mat4::identity() {
    Mat4 r;
    std::memset(r.m, 0, sizeof(r.m));
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0;
    return r;
}
Mat4 Mat4::translation(double x, double y, double z) {
    Mat4 r = identity();
    r.m[3] = x;
    r.m[7] = y;
    r.m[11] = z;
    return r;
}
Mat4 Mat4::from_euler_angles(double roll, double pitch, double yaw) {
    double sr = dm_sin(roll), cr = dm_cos(roll), sp = dm_sin(pitch), cp = dm_cos(pitch), sy = dm_sin(yaw),
           cy = dm_cos(yaw);
    Mat4 r = identity();
    r.m[0] = cy * cp; r.m[1] = cy * sp * sr - sy * cr; r.m[2] = cy * sp * cr + sy * sr;
    r.m[4] = sy * cp; r.m[5] = sy * sp * sr + cy * cr; r.m[6] = sy * sp * cr - cy * sr;
    r.m[8] = -sp;     r.m[9] = cp * sr;                r.m[10] = cp * cr;
    return r;
}
Mat4 Mat4::from_scaling(double s) {
    Mat4 r = identity();
    r.m[0] = r.m[5] = r.m[10] = s;
    return r;
}
Mat4 Mat4::similarity(Vec3 t, double s) {
    Mat4 r = from_scaling(s);
    r.m[3] = t.x;
    r.m[7] = t.y;
    r.m[11] = t.z;
    return r;
}
Mat4 Mat4::operator*(const Mat4& o) const {
    Mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += m[i * 4 + k] * o.m[k * 4 + j];
            r.m[i * 4 + j] = s;
        }
    return r;
}
Mat4 Mat4::from_rows(const double (&r)[16]) {
    Mat4 o;
    for (int i = 0; i < 16; i++) o.m[i] = r[i];
    return o;
}
Mat4 Mat4::affine_inverse() const {
    // inverse of [A t; 0 1] = [A^-1  -A^-1 t; 0 1], A^-1 by cofactors
    double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    double id = 1.0 / det;
    Mat4 r = identity();
    r.m[0] = (e * i - f * h) * id; r.m[1] = (c * h - b * i) * id; r.m[2] = (b * f - c * e) * id;
    r.m[4] = (f * g - d * i) * id; r.m[5] = (a * i - c * g) * id; r.m[6] = (c * d - a * f) * id;
    r.m[8] = (d * h - e * g) * id; r.m[9] = (b * g - a * h) * id; r.m[10] = (a * e - b * d) * id;
    r.m[3] = -(r.m[0] * m[3] + r.m[1] * m[7] + r.m[2] * m[11]);
    r.m[7] = -(r.m[4] * m[3] + r.m[5] * m[7] + r.m[6] * m[11]);
    r.m[11] = -(r.m[8] * m[3] + r.m[9] * m[7] + r.m[10] * m[11]);
    return r;
}
Vec3 Mat4::transform_point(Vec3 p) const {
    return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
            m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
}
Vec3 Mat4::transform_vector(Vec3 v) const {
    return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
            m[8] * v.x + m[9] * v.y + m[10] * v.z};
}

// ---------------------------------------------------------------- Camera
Camera Camera::create(Vec3 from, Vec3 to, Vec3 up, double aspect_ratio, double vfov, double aperture,
                      double focus_dist) {
    return new_motion_blur(from, to, up, aspect_ratio, vfov, aperture, focus_dist, 0.0, 0.0);
}
// geometry.rs:133-175
Camera Camera::new_motion_blur(Vec3 from, Vec3 to, Vec3 up, double aspect_ratio, double vfov, double aperture,
                               double focus_dist, double t0, double t1) {
    Vec3 w = normalize(to - from);
    Vec3 u = -normalize(cross(up, w));
    Vec3 v = -normalize(cross(w, u));
    double theta = vfov * PI / 180.0;
    double h = dm_sin(theta / 2.0) / dm_cos(theta / 2.0);  // tan(theta/2) from the contract's sin/cos
    double viewport_height = 2.0 * h;
    double viewport_width = viewport_height * aspect_ratio;
    Vec3 horizontal = u * (viewport_width * focus_dist);
    Vec3 vertical = v * (viewport_height * focus_dist);
    Vec3 ulc = from - horizontal * 0.5 + vertical * 0.5 + w * focus_dist;
    Camera cam;
    rt_camera& c = cam.c;
    c.origin[0] = from.x; c.origin[1] = from.y; c.origin[2] = from.z;
    c.upper_left_corner[0] = ulc.x; c.upper_left_corner[1] = ulc.y; c.upper_left_corner[2] = ulc.z;
    c.horizontal_offset[0] = horizontal.x; c.horizontal_offset[1] = horizontal.y; c.horizontal_offset[2] = horizontal.z;
    c.vertical_offset[0] = vertical.x; c.vertical_offset[1] = vertical.y; c.vertical_offset[2] = vertical.z;
    c.lens_radius = aperture / 2.0;
    c.t0 = t0;
    c.t1 = t1;
    c.u[0] = u.x; c.u[1] = u.y; c.u[2] = u.z;
    c.v[0] = v.x; c.v[1] = v.y; c.v[2] = v.z;
    c.w[0] = w.x; c.w[1] = w.y; c.w[2] = w.z;
    return cam;
}

// ------------------------------------------------------------- Primitive
static Primitive blank(uint32_t kind, uint32_t mat_index) {
    Primitive p;
    std::memset(&p.r, 0, sizeof(p.r));
    std::memset(&p.xform, 0, sizeof(p.xform));
    p.r.kind = kind;
    p.r.mat_index = mat_index;
    p.r.light_index = -1;
    p.r.xform_index = -1;
    return p;
}
// primitive.rs:64-76
Primitive Primitive::new_sphere(Vec3 c, double r, uint32_t mat_index) {
    Primitive p = blank(RT_PRIM_SPHERE, mat_index);
    p.r.v[0] = c.x; p.r.v[1] = c.y; p.r.v[2] = c.z; p.r.v[3] = r;
    p.r.bbox_min[0] = c.x - r; p.r.bbox_min[1] = c.y - r; p.r.bbox_min[2] = c.z - r;
    p.r.bbox_max[0] = c.x + r; p.r.bbox_max[1] = c.y + r; p.r.bbox_max[2] = c.z + r;
    return p;
}
// util.rs:493-517 get_new_box
static void get_new_box(const double* bmin, const double* bmax, const Mat4& t, double* omin, double* omax) {
    for (int a = 0; a < 3; a++) {
        omin[a] = INF;
        omax[a] = -INF;
    }
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int k = 0; k < 2; k++) {
                Vec3 p{i == 0 ? bmin[0] : bmax[0], j == 0 ? bmin[1] : bmax[1], k == 0 ? bmin[2] : bmax[2]};
                Vec3 np = t.transform_point(p);
                double c[3] = {np.x, np.y, np.z};
                for (int a = 0; a < 3; a++) {
                    omin[a] = std::fmin(omin[a], c[a]);
                    omax[a] = std::fmax(omax[a], c[a]);
                }
            }
}
// primitive.rs:78-229: rect AABBs are padded by SMALL along the normal (Q5)
static Primitive rect(uint32_t kind, double a0, double b0, double a1, double b1, double k, uint32_t mat_index,
                      const Mat4* transform) {
    Primitive p = blank(kind, mat_index);
    p.r.v[0] = a0; p.r.v[1] = b0; p.r.v[2] = a1; p.r.v[3] = b1; p.r.v[4] = k;
    double bmin[3], bmax[3];
    if (kind == RT_PRIM_XY_RECT) {
        bmin[0] = a0; bmin[1] = b0; bmin[2] = k - SMALL;
        bmax[0] = a1; bmax[1] = b1; bmax[2] = k + SMALL;
    } else if (kind == RT_PRIM_XZ_RECT) {
        bmin[0] = a0; bmin[1] = k - SMALL; bmin[2] = b0;
        bmax[0] = a1; bmax[1] = k + SMALL; bmax[2] = b1;
    } else {
        bmin[0] = k - SMALL; bmin[1] = a0; bmin[2] = b0;
        bmax[0] = k + SMALL; bmax[1] = a1; bmax[2] = b1;
    }
    if (transform) {
        get_new_box(bmin, bmax, *transform, p.r.bbox_min, p.r.bbox_max);
        p.has_xform = 1;
        Mat4 inv = transform->affine_inverse();
        std::memcpy(p.xform.fwd, transform->m, sizeof(double) * 12);
        std::memcpy(p.xform.inv, inv.m, sizeof(double) * 12);
    } else {
        std::memcpy(p.r.bbox_min, bmin, sizeof(bmin));
        std::memcpy(p.r.bbox_max, bmax, sizeof(bmax));
    }
    return p;
}
Primitive Primitive::new_xy_rect(double x0, double y0, double x1, double y1, double k, uint32_t m) {
    return rect(RT_PRIM_XY_RECT, x0, y0, x1, y1, k, m, nullptr);
}
Primitive Primitive::new_xz_rect(double x0, double z0, double x1, double z1, double k, uint32_t m) {
    return rect(RT_PRIM_XZ_RECT, x0, z0, x1, z1, k, m, nullptr);
}
Primitive Primitive::new_yz_rect(double y0, double z0, double y1, double z1, double k, uint32_t m) {
    return rect(RT_PRIM_YZ_RECT, y0, z0, y1, z1, k, m, nullptr);
}
Primitive Primitive::new_xy_rect_transform(double x0, double y0, double x1, double y1, double k, uint32_t m,
                                           const Mat4* t) {
    return rect(RT_PRIM_XY_RECT, x0, y0, x1, y1, k, m, t);
}
Primitive Primitive::new_xz_rect_transform(double x0, double z0, double x1, double z1, double k, uint32_t m,
                                           const Mat4* t) {
    return rect(RT_PRIM_XZ_RECT, x0, z0, x1, z1, k, m, t);
}
Primitive Primitive::new_yz_rect_transform(double y0, double z0, double y1, double z1, double k, uint32_t m,
                                           const Mat4* t) {
    return rect(RT_PRIM_YZ_RECT, y0, z0, y1, z1, k, m, t);
}
Primitive Primitive::new_flip_face(Primitive obj) {
    obj.r.flip = 1;  // FlipFace{FlipFace{..}} never occurs in the reference
    return obj;
}
// primitive.rs:339-359
double Primitive::area(const Objects& objs) const {
    switch (r.kind) {
        case RT_PRIM_SPHERE: return 2.0 * PI * r.v[3];
        case RT_PRIM_TRIANGLE: {
            const Mesh& m = objs.meshes[r.mesh_index];
            uint32_t i0 = m.ind[r.tri_ind], i1 = m.ind[r.tri_ind + 1], i2 = m.ind[r.tri_ind + 2];
            Vec3 p0{m.p[3 * i0], m.p[3 * i0 + 1], m.p[3 * i0 + 2]};
            Vec3 p1{m.p[3 * i1], m.p[3 * i1 + 1], m.p[3 * i1 + 2]};
            Vec3 p2{m.p[3 * i2], m.p[3 * i2 + 1], m.p[3 * i2 + 2]};
            return 0.5 * norm(cross(p1 - p0, p2 - p0));
        }
        default: return (r.v[2] - r.v[0]) * (r.v[3] - r.v[1]);
    }
}

// ------------------------------------------------------------------ Cube
Cube Cube::new_transform(Vec3 min, Vec3 max, uint32_t mat_index, const Mat4& transform) {
    return Cube{min, max, mat_index, transform};
}
// hittable.rs:788-846: order z0, z1, y0, y1, x0, x1
std::vector<Primitive> Cube::get_sides() const {
    const Mat4* t = &transform;
    Primitive z0 = Primitive::new_flip_face(Primitive::new_xy_rect_transform(min.x, min.y, max.x, max.y, min.z, mat_index, t));
    Primitive z1 = Primitive::new_xy_rect_transform(min.x, min.y, max.x, max.y, max.z, mat_index, t);
    Primitive x0 = Primitive::new_flip_face(Primitive::new_yz_rect_transform(min.y, min.z, max.y, max.z, min.x, mat_index, t));
    Primitive x1 = Primitive::new_yz_rect_transform(min.y, min.z, max.y, max.z, max.x, mat_index, t);
    Primitive y0 = Primitive::new_flip_face(Primitive::new_xz_rect_transform(min.x, min.z, max.x, max.z, min.y, mat_index, t));
    Primitive y1 = Primitive::new_xz_rect_transform(min.x, min.z, max.x, max.z, max.y, mat_index, t);
    return {z0, z1, y0, y1, x0, x1};
}

// ------------------------------------------------------ Texture / Material
rt_texture Texture::new_solid_color(Vec3 c) {
    rt_texture t;
    std::memset(&t, 0, sizeof(t));
    t.kind = RT_TEX_SOLID;
    t.color[0] = c.x; t.color[1] = c.y; t.color[2] = c.z;
    return t;
}
rt_texture Texture::new_checkered(uint32_t even, uint32_t odd, double frequency) {
    rt_texture t;
    std::memset(&t, 0, sizeof(t));
    t.kind = RT_TEX_CHECKERED;
    t.even = even;
    t.odd = odd;
    t.frequency = frequency;
    return t;
}
// ---- Texture::Hdr (row f4)
// image 0.23.12 (Cargo.lock; not vendored) image::hdr, restated from its published source:
//   Rgbe8Pixel::to_hdr:  e == 0 -> 0, else exp2(e - (128 + 8)) * c[i]           (f32)
//   to_rgbe8:  mx = max(r, g, b); mx <= 0 -> (0,0,0,0); else exp = floor(log2(mx)) + 1,
//              c[i] = trunc(v[i] / 2^exp * 256) as u8, e = (exp + 128) as u8    (f32)
void Texture::rgbe_to_hdr(const uint8_t* q, float* rgb) {
    if (q[3] == 0) {
        rgb[0] = rgb[1] = rgb[2] = 0.0f;
        return;
    }
    const float ex = std::exp2((float)q[3] - (128.0f + 8.0f));
    for (int i = 0; i < 3; i++) rgb[i] = ex * (float)q[i];
}
static uint8_t sat_u8(float x) {  // Rust `f32 as u8`: saturating, NaN -> 0
    if (!(x > 0.0f)) return 0;
    if (x >= 255.0f) return 255;
    return (uint8_t)x;
}
void Texture::hdr_to_rgbe8(const float* rgb, uint8_t* q) {
    const float mx = std::fmax(rgb[0], std::fmax(rgb[1], rgb[2]));
    if (mx <= 0.0f) {
        q[0] = q[1] = q[2] = q[3] = 0;
        return;
    }
    const int ex = (int)std::floor(std::log2(mx)) + 1;
    const float mul = std::ldexp(1.0f, ex);
    for (int i = 0; i < 3; i++) q[i] = sat_u8(std::trunc(rgb[i] / mul * 256.0f));
    q[3] = (uint8_t)(ex + 128);
}
static rt_texture hdr_texture(Objects& objs, std::shared_ptr<std::vector<uint8_t>> texels, uint32_t w, uint32_t h) {
    rt_texture t;
    std::memset(&t, 0, sizeof(t));
    t.kind = RT_TEX_HDR;
    t.width = w;
    t.height = h;
    t.rgbe = texels->data();
    objs.hdr_store.push_back(std::move(texels));
    return t;
}
// Radiance RGBE reader: header lines up to the blank line, "-Y h +X w", then scanlines, new-style RLE
// (2 2 hi lo, then the four channels run-length coded) or flat RGBE quadruples.
bool Texture::new_hdr(Objects& objs, const std::string& path, rt_texture& out, std::string& err) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) {
        err = "Unable to open file " + path;  // material.rs:632
        return false;
    }
    std::vector<uint8_t> buf;
    {
        uint8_t chunk[65536];
        size_t got;
        while ((got = std::fread(chunk, 1, sizeof(chunk), f)) > 0) buf.insert(buf.end(), chunk, chunk + got);
        std::fclose(f);
    }
    size_t pos = 0;
    auto line = [&](std::string& l) {
        l.clear();
        if (pos >= buf.size()) return false;
        while (pos < buf.size() && buf[pos] != '\n') l.push_back((char)buf[pos++]);
        pos++;
        return true;
    };
    std::string l;
    if (!line(l) || (l.rfind("#?RADIANCE", 0) != 0 && l.rfind("#?RGBE", 0) != 0)) {
        err = "Failure decoding hdr " + path;
        return false;
    }
    while (line(l) && !l.empty()) {}
    long w = 0, h = 0;
    if (!line(l) || std::sscanf(l.c_str(), "-Y %ld +X %ld", &h, &w) != 2 || w <= 0 || h <= 0 || w > 65535 || h > 65535) {
        err = "Failure decoding hdr " + path + " (only -Y h +X w is supported)";
        return false;
    }
    auto texels = std::make_shared<std::vector<uint8_t>>((size_t)w * h * 4);
    std::vector<uint8_t> row((size_t)w * 4);
    for (long y = 0; y < h; y++) {
        bool ok = true;
        if (pos + 4 <= buf.size() && buf[pos] == 2 && buf[pos + 1] == 2 && ((buf[pos + 2] << 8) | buf[pos + 3]) == w &&
            w >= 8 && w < 32768) {
            pos += 4;
            for (int ch = 0; ch < 4 && ok; ch++) {
                long x = 0;
                while (x < w && ok) {
                    if (pos >= buf.size()) { ok = false; break; }
                    uint8_t cnt = buf[pos++];
                    if (cnt > 128) {
                        cnt -= 128;
                        if (pos >= buf.size() || x + cnt > w) { ok = false; break; }
                        const uint8_t v = buf[pos++];
                        for (int k = 0; k < cnt; k++) row[(size_t)(x++) * 4 + ch] = v;
                    } else {
                        if (cnt == 0 || pos + cnt > buf.size() || x + cnt > w) { ok = false; break; }
                        for (int k = 0; k < cnt; k++) row[(size_t)(x++) * 4 + ch] = buf[pos++];
                    }
                }
            }
        } else {
            if (pos + (size_t)w * 4 > buf.size()) ok = false;
            else {
                std::memcpy(row.data(), &buf[pos], (size_t)w * 4);
                pos += (size_t)w * 4;
            }
        }
        if (!ok) {
            err = "Failure to decode data from hdr " + path;
            return false;
        }
        for (long x = 0; x < w; x++) {  // read_image_hdr -> Rgb<f32>, then get_value's to_rgbe8
            float rgb[3];
            rgbe_to_hdr(&row[(size_t)x * 4], rgb);
            hdr_to_rgbe8(rgb, &(*texels)[((size_t)y * w + x) * 4]);
        }
    }
    out = hdr_texture(objs, std::move(texels), (uint32_t)w, (uint32_t)h);
    return true;
}
rt_texture Texture::new_hdr_procedural(Objects& objs, uint32_t w, uint32_t h) {
    auto texels = std::make_shared<std::vector<uint8_t>>((size_t)w * h * 4);
    const double pi = 3.141592653589793;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            // lat-long: v = 0 is the zenith.  Blue-ish sky brightening to the horizon, a warm ground, one sun.
            const double v = (y + 0.5) / h, u = (x + 0.5) / w;
            const double el = (0.5 - v) * pi;  // elevation
            double r, g, b;
            if (el > 0.0) {
                const double t = std::pow(1.0 - std::sin(el), 3.0);
                r = 0.25 + 0.9 * t; g = 0.45 + 0.8 * t; b = 0.9 + 0.5 * t;
            } else {
                const double t = std::pow(1.0 + std::sin(el), 4.0);
                r = 0.22 + 0.5 * t; g = 0.18 + 0.45 * t; b = 0.12 + 0.4 * t;
            }
            const double du = (u - 0.3) * 2.0 * pi * std::cos(el), dv = el - 0.7;  // sun at azimuth 0.3, elevation 0.7 rad
            const double d2 = du * du + dv * dv;
            const double sun = 400.0 * std::exp(-d2 / (2.0 * 0.03 * 0.03)) + 6.0 * std::exp(-d2 / (2.0 * 0.25 * 0.25));
            const float rgb[3] = {(float)(r + sun), (float)(g + 0.9 * sun), (float)(b + 0.7 * sun)};
            hdr_to_rgbe8(rgb, &(*texels)[((size_t)y * w + x) * 4]);
        }
    return hdr_texture(objs, std::move(texels), w, h);
}
static rt_material mat_blank(uint32_t kind) {
    rt_material m;
    std::memset(&m, 0, sizeof(m));
    m.kind = kind;
    for (int i = 0; i < 5; i++) m.tex[i] = RT_NO_TEXTURE;
    return m;
}
rt_material Material::make_matte(uint32_t k_d_id, double sigma, uint32_t) {
    rt_material m = mat_blank(RT_MAT_MATTE);
    m.tex[0] = k_d_id;
    m.f[0] = sigma;
    return m;
}
rt_material Material::make_light(uint32_t texture_id) {
    rt_material m = mat_blank(RT_MAT_LIGHT);
    m.tex[0] = texture_id;
    return m;
}
rt_material Material::make_plastic(uint32_t k_d_id, uint32_t k_s_id, uint32_t, double roughness, bool remap) {
    rt_material m = mat_blank(RT_MAT_PLASTIC);
    m.tex[0] = k_d_id;
    m.tex[1] = k_s_id;
    m.f[0] = roughness;
    m.remap_roughness = remap ? 1 : 0;
    return m;
}
rt_material Material::make_glass(uint32_t k_r_id, uint32_t k_t_id, double ur, double vr, double index, uint32_t,
                                 bool remap) {
    rt_material m = mat_blank(RT_MAT_GLASS);
    m.tex[0] = k_r_id;
    m.tex[1] = k_t_id;
    m.f[0] = ur;
    m.f[1] = vr;
    m.f[2] = index;
    m.remap_roughness = remap ? 1 : 0;
    return m;
}
// material.rs:452-470: argument order (eta, k, u_r, v_r, r, bump, remap)
rt_material Material::make_metal(uint32_t eta_id, uint32_t k_id, uint32_t u_r_id, uint32_t v_r_id, uint32_t r_id,
                                 uint32_t, bool remap) {
    rt_material m = mat_blank(RT_MAT_METAL);
    m.tex[0] = eta_id;
    m.tex[1] = k_id;
    m.tex[2] = r_id;
    m.tex[3] = u_r_id;
    m.tex[4] = v_r_id;
    m.remap_roughness = remap ? 1 : 0;
    return m;
}
rt_material Material::make_mirror(uint32_t color_id, uint32_t) {
    rt_material m = mat_blank(RT_MAT_MIRROR);
    m.tex[0] = color_id;
    return m;
}

rt_light Light::make_diffuse_light(const Objects& objs, uint32_t prim_index, Vec3 color, uint32_t, bool two_sided,
                                   bool) {
    rt_light l;
    std::memset(&l, 0, sizeof(l));
    l.kind = RT_LIGHT_DIFFUSE;
    l.prim_index = prim_index;
    l.two_sided = two_sided ? 1 : 0;
    l.color[0] = color.x; l.color[1] = color.y; l.color[2] = color.z;
    l.area = objs.objs[prim_index].area(objs);
    l.xform_index = -1;
    return l;
}
rt_light Light::make_infinite_light(Objects& objs, const Mat4* to_world, uint32_t, uint32_t text_id) {
    rt_light l;
    std::memset(&l, 0, sizeof(l));
    l.kind = RT_LIGHT_INFINITE;
    l.tex_index = text_id;
    l.world_radius = 10000.0;  // light.rs:636
    l.xform_index = -1;
    if (to_world) {
        rt_xform x;
        const Mat4 inv = to_world->affine_inverse();  // to_obj = to_world.inverse() (light.rs:609)
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) {
                x.fwd[r * 4 + c] = to_world->m[r * 4 + c];
                x.inv[r * 4 + c] = inv.m[r * 4 + c];
            }
        objs.light_xforms.emplace_back((uint32_t)objs.lights.size(), x);  // the caller pushes the light next
    }
    return l;
}

// hittable.rs:257-288
std::vector<Primitive> generate_triangles(const std::vector<Mesh>& meshes, uint32_t mesh_index, uint32_t mat_index) {
    std::vector<Primitive> out;
    const Mesh& m = meshes[mesh_index];
    out.reserve(m.ind.size() / 3);
    for (size_t index = 0; index + 2 < m.ind.size(); index += 3) {
        Primitive p = blank(RT_PRIM_TRIANGLE, mat_index);
        p.r.mesh_index = mesh_index;
        p.r.tri_ind = (uint32_t)index;
        for (int a = 0; a < 3; a++) {
            double v1 = m.p[3 * m.ind[index] + a], v2 = m.p[3 * m.ind[index + 1] + a], v3 = m.p[3 * m.ind[index + 2] + a];
            p.r.bbox_min[a] = std::fmin(v1, std::fmin(v2, v3));
            p.r.bbox_max[a] = std::fmax(v1, std::fmax(v2, v3));
        }
        out.push_back(p);
    }
    return out;
}

// ---------------------------------------------------------------- flatten
void FlatScene::build(Objects&& objs) {
    mesh_store = std::move(objs.meshes);
    meshes.clear();
    for (const Mesh& m : mesh_store) {
        rt_mesh r;
        r.p = m.p.data();
        r.n = m.n.empty() ? nullptr : m.n.data();
        r.uv = m.uv.empty() ? nullptr : m.uv.data();
        r.ind = m.ind.data();
        r.n_p = m.p.size() / 3;
        r.n_n = m.n.size() / 3;
        r.n_uv = m.uv.size() / 2;
        r.n_ind = m.ind.size();
        meshes.push_back(r);
    }
    prims.clear();
    xforms.clear();
    for (const Primitive& p : objs.objs) {
        rt_primitive r = p.r;
        if (p.has_xform) {
            // identical transforms (the six sides of a Cube share one Arc) are stored once
            int found = -1;
            for (size_t i = 0; i < xforms.size(); i++)
                if (std::memcmp(&xforms[i], &p.xform, sizeof(rt_xform)) == 0) found = (int)i;
            if (found < 0) {
                xforms.push_back(p.xform);
                found = (int)xforms.size() - 1;
            }
            r.xform_index = found;
        }
        prims.push_back(r);
    }
    materials = std::move(objs.materials);
    textures = std::move(objs.textures);
    hdr_store = std::move(objs.hdr_store);
    lights = std::move(objs.lights);
    for (auto& lx : objs.light_xforms) {
        xforms.push_back(lx.second);
        lights[lx.first].xform_index = (int32_t)xforms.size() - 1;
    }
    desc.meshes = meshes.data();       desc.n_meshes = meshes.size();
    desc.prims = prims.data();         desc.n_prims = prims.size();
    desc.xforms = xforms.data();       desc.n_xforms = xforms.size();
    desc.materials = materials.data(); desc.n_materials = materials.size();
    desc.textures = textures.data();   desc.n_textures = textures.size();
    desc.lights = lights.data();       desc.n_lights = lights.size();
}

}  // namespace rr

// shading.h -- materials, BSDF lobes, microfacet distribution and area-light
// sampling for gfx950, register-resident (no heap, at most two lobes per vertex).
//
// Replaces with identical f64 results:
//   Texture::value                        src/material.rs:542-565       (Q19)
//   Material::compute_scattering          src/material.rs:80-244
//   Bsdf::{f,sample_f,pdf}                src/bsdf.rs:83-189            (Q10, Q11)
//   Bxdf::{f,sample_f,pdf}                src/bxdf.rs:328-392, 532-607, 640-684, 721-741, 815-835
//   fr_dielectric / fr_conductor          src/bxdf.rs:113-211
//   TrowbridgeReitz d/lambda/g/pdf/sample src/microfacet.rs:53-68, 109-172, 240-282, 442-512
//   Primitive::{area,sample_area,pdf}     src/primitive.rs:339-359, 438-539 (Q8, Q9)
//   Light::l                              src/light.rs:475-496           (Q7)
// next-row f4:
//   Texture::Hdr arm of get_value         src/material.rs:570-587
//   Bxdf::MicrofacetTransmission          src/bxdf.rs:393-441, 608-638, 742-763
//   Distribution1D/2D sampling and pdf    src/distribution.rs:64-78, 133-166
//   Light::Infinite sample_li/pdf_li/le   src/light.rs:204-245, 285-294, 499-512
#pragma once
#include "geom.h"

namespace rtd {

RTD bool same_hemisphere(D3 v, D3 w) { return v.z * w.z > 0.0; }  // util.rs:591-593
RTD D3 reflect(D3 v, D3 n) {                                     // util.rs:203-206
    double scale = 2.0 * dot(v, n);
    return -v + n * scale;
}
RTD bool refract(D3 vec, D3 n, double eta, D3& out) {  // util.rs:376-385
    double cos_theta_i = dot(n, vec) / norm(vec);
    double sin2_theta_i = rmax(0.0, 1.0 - cos_theta_i * cos_theta_i);
    double sin2_theta_t = eta * eta * sin2_theta_i;
    if (sin2_theta_t >= 1.0) return false;
    double cos_theta_t = dm_sqrt(1.0 - sin2_theta_t);
    out = eta * (-vec) + (eta * cos_theta_i - cos_theta_t) * d3(n.x, n.y, n.z);
    return true;
}
// util.rs:127-148
RTD D3 rand_cosine_dir(double r1, double r2) {
    double u1 = 2.0 * r1 - 1.0, u2 = 2.0 * r2 - 1.0;
    if (u1 == 0.0 && u2 == 0.0) return d3(0.0, 0.0, 1.0);
    double theta, r;
    if (absd(u1) > absd(u2)) {
        r = u1;
        theta = kPi / 4.0 * (u2 / u1);
    } else {
        r = u2;
        theta = kPi / 2.0 - kPi / 4.0 * (u1 / u2);
    }
    const SinCos sc_ = sincos2(theta);
    double x = r * sc_.c;
    double y = r * sc_.s;
    double z = dm_sqrt(rmax(0.0, 1.0 - x * x - y * y));
    return d3(x, y, z);
}

// material.rs:542-565: Checkered follows even/odd ids (bounded depth, like the oracle)
// noinline + rolled loop: inlining this (two dm_sin per checker level, eight levels, a dozen call
// sites in compute_scattering) multiplied k_shade's code size several times over the I-cache.
RTD uint32_t sat_u32(double x) {  // Rust `f64 as u32`: saturating, NaN -> 0
    if (!(x > 0.0)) return 0u;
    if (x >= 4294967295.0) return 4294967295u;
    return (uint32_t)x;
}
// material.rs:570-587; texels are image::hdr::to_rgbe8(data[i]) (rt_abi.h)
RTD D3 hdr_value(const rt_texture& t, double u, double v) {
    const uint32_t width = t.width, height = t.height;
    uint32_t x = sat_u32(__builtin_round((1.0 - u) * (double)width));
    uint32_t y = sat_u32(__builtin_round(v * (double)height));
    x = x % width;
    y = y % height;
    const uint32_t q = reinterpret_cast<const uint32_t*>(t.rgbe)[(size_t)y * width + x];  // c0 | c1<<8 | c2<<16 | e<<24
    const f64_t sc = dm_from_bits((uint64_t)(1023 + (int)(q >> 24) - 128) << 52);        // 2^(e-128), exact
    return d3(((double)(q & 0xffu) + 0.5) * sc / 256.0, ((double)((q >> 8) & 0xffu) + 0.5) * sc / 256.0,
              ((double)((q >> 16) & 0xffu) + 0.5) * sc / 256.0);
}
RTDN D3 texture_value(const DevScene& sc, uint32_t index, double u, double v) {
#pragma unroll 1
    for (int depth = 0; depth < 8; depth++) {
        const rt_texture& t = sc.texs[index];
        if (t.kind != RT_TEX_CHECKERED) break;
        double mult = dm_sin(t.frequency * u * 2.0 * kPi) * dm_sin(t.frequency * v * 2.0 * kPi);
        index = (mult < 0.0) ? t.even : t.odd;
    }
    const rt_texture& t = sc.texs[index];
    if (t.kind == RT_TEX_HDR) return hdr_value(t, u, v);
    return d3(t.color[0], t.color[1], t.color[2]);
}

// bxdf.rs:113-136
RTD double fr_dielectric(double cos_theta_i, double eta_i, double eta_t) {
    cos_theta_i = clampd(cos_theta_i, -1.0, 1.0);
    double index_i = eta_i, index_t = eta_t;
    if (cos_theta_i < 0.0) {
        index_i = eta_t;
        index_t = eta_i;
        cos_theta_i = absd(cos_theta_i);
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
RTD D3 fr_conductor(double cos_theta_i, D3 eta, D3 eta_k) {
    cos_theta_i = clampd(cos_theta_i, -1.0, 1.0);
    double c2 = cos_theta_i * cos_theta_i;
    double s2 = 1.0 - c2;
    D3 eta2 = cmul(eta, eta);
    D3 etak2 = cmul(eta_k, eta_k);
    D3 t0 = (eta2 - etak2) - d3(s2, s2, s2);
    D3 a2pb2 = cmul(t0, t0) + cmul(eta2, etak2) * 4.0;
    a2pb2 = d3(dm_sqrt(a2pb2.x), dm_sqrt(a2pb2.y), dm_sqrt(a2pb2.z));
    D3 t1 = a2pb2 + d3(c2, c2, c2);
    D3 a = (a2pb2 + t0) * 0.5;
    a = d3(dm_sqrt(a.x), dm_sqrt(a.y), dm_sqrt(a.z));
    D3 t2 = a * (2.0 * cos_theta_i);
    D3 rs = cdiv(t1 - t2, t1 + t2);
    D3 t3 = a2pb2 * c2 + d3(s2 * s2, s2 * s2, s2 * s2);
    D3 t4 = t2 * s2;
    D3 rp = cmul(rs, cdiv(t3 - t4, t3 + t4));
    return (rp + rs) * 0.5;
}

enum LobeKind { LOBE_LAMBERT = 0, LOBE_MICROFACET = 1, LOBE_FRESNEL_SPECULAR = 2, LOBE_SPECULAR_REFL = 3, LOBE_MICRO_TRANS = 4 };
enum FresnelKind { FR_DIELECTRIC = 0, FR_CONDUCTOR = 1, FR_NOOP = 2 };

struct Lobe {
    int kind;
    uint32_t type;
    D3 color;  // Lambert / microfacet colour, FresnelSpecular r
    D3 t;      // FresnelSpecular t; FresnelConductor eta
    D3 k;      // FresnelConductor k
    int fresnel;
    double p0, p1;  // dielectric (eta_i, eta_t) | FresnelSpecular, MicrofacetTransmission (eta_a, eta_b)
    double alpha_x, alpha_y;
};
struct Bsdf {
    D3 ns, ng, ss, ts;
    int n;
    Lobe lobes[2];
};

RTD D3 fresnel_evaluate(const Lobe& l, double cos_theta_i) {  // bxdf.rs:190-211
    if (l.fresnel == FR_DIELECTRIC) {
        double v = fr_dielectric(absd(cos_theta_i), l.p1, l.p0);
        return d3(v, v, v);
    }
    if (l.fresnel == FR_CONDUCTOR) return fr_conductor(absd(cos_theta_i), l.t, l.k);
    return white();
}

// bxdf.rs:12-56
RTD double cos_sq_theta(D3 v) { return v.z * v.z; }
RTD double sin_sq_theta(D3 v) { return rmax(0.0, 1.0 - cos_sq_theta(v)); }
RTD double sin_theta(D3 v) { return dm_sqrt(sin_sq_theta(v)); }
RTD double tan_theta(D3 v) { return sin_theta(v) / v.z; }
RTD double tan_sq_theta(D3 v) { return sin_sq_theta(v) / cos_sq_theta(v); }
RTD double cos_phi(D3 v) {
    double s = sin_theta(v);
    return s == 0.0 ? 1.0 : clampd(v.x / s, -1.0, 1.0);
}
RTD double sin_phi(D3 v) {
    double s = sin_theta(v);
    return s == 0.0 ? 1.0 : clampd(v.y / s, -1.0, 1.0);
}
RTD double cos_sq_phi(D3 v) { return cos_phi(v) * cos_phi(v); }
RTD double sin_sq_phi(D3 v) { return sin_phi(v) * sin_phi(v); }

constexpr double kHugeInf = __builtin_huge_val();

RTD double tr_d(double ax, double ay, D3 wh) {  // microfacet.rs:53-68
    double t2 = tan_sq_theta(wh);
    if (t2 == kHugeInf) return 0.0;
    double cos4 = cos_sq_theta(wh) * cos_sq_theta(wh);
    double e = (cos_sq_phi(wh) / (ax * ax) + sin_sq_phi(wh) / (ay * ay)) * t2;
    return 1.0 / (kPi * ax * ay * cos4 * (1.0 + e) * (1.0 + e));
}
RTD double tr_lambda(double ax, double ay, D3 w) {  // microfacet.rs:109-123
    double abs_tan = absd(tan_theta(w));
    if (abs_tan == kHugeInf) return 0.0;
    double alpha = dm_sqrt(cos_sq_phi(w) * ax * ax + sin_sq_phi(w) * ay * ay);
    double a2t2 = (alpha * abs_tan) * (alpha * abs_tan);
    return (-1.0 + dm_sqrt(1.0 + a2t2)) / 2.0;
}
RTD double tr_g1(double ax, double ay, D3 w) { return 1.0 / (1.0 + tr_lambda(ax, ay, w)); }
RTD double tr_g(double ax, double ay, D3 wo, D3 wi) {
    return 1.0 / (1.0 + tr_lambda(ax, ay, wo) + tr_lambda(ax, ay, wi));
}
RTD double tr_pdf(double ax, double ay, D3 wo, D3 wh) {  // microfacet.rs:163-168
    return tr_d(ax, ay, wh) * tr_g1(ax, ay, wo) * absd(dot(wo, wh)) / absd(wo.z);
}
RTD void tr_sample_11(double cos_theta, double u1, double u2, double& sx, double& sy) {  // microfacet.rs:470-512
    if (cos_theta > 0.9999) {
        double r = dm_sqrt(u1 / (1.0 - u1));
        double phi = 2.0 * kPi * u2;
        const SinCos sc_ = sincos2(phi);
        sx = r * sc_.c;
        sy = r * sc_.s;
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
RTD D3 tr_sample(D3 wi, double ax, double ay, double u1, double u2) {  // microfacet.rs:448-468
    D3 wi_s = normalize(d3(ax * wi.x, ay * wi.y, wi.z));
    double sx, sy;
    tr_sample_11(wi_s.z, u1, u2, sx, sy);
    double sp = sin_phi(wi_s), cp = cos_phi(wi_s);
    double tmp = cp * sx - sp * sy;
    sy = sp * sx + cp * sy;
    sx = tmp;
    sx = ax * sx;
    sy = ay * sy;
    return normalize(d3(-sx, -sy, 1.0));
}
RTD D3 tr_sample_wh(double ax, double ay, D3 wo, double u0, double u1) {  // microfacet.rs:272-281
    bool flip = wo.z < 0.0;
    D3 wh = tr_sample(flip ? -wo : wo, ax, ay, u0, u1);
    return flip ? -wh : wh;
}
RTD double tr_roughness_to_alpha(double roughness) {  // microfacet.rs:442-446
    roughness = rmax(roughness, 1e-5);
    double x = dm_log(roughness);
    return 1.62142 + 0.819955 * x + 0.1734 * x * x + 0.0171201 * x * x * x + 0.000640711 * x * x * x * x;
}

RTD bool matches_flags(uint32_t flag, uint32_t other) { return (flag & other) == flag; }

// FEAT: which lobes / features a kernel instance is compiled with.  rt_scene_commit picks the smallest
// instance that covers the scene's materials and lights, so a Lambertian scene does not pay (in registers,
// hence occupancy of the latency-bound shading kernel, and in code) for microfacet, specular and two-lobe
// materials it does not contain, nor any scene for the row-f4 features it does not use.
constexpr int kFeatMicro = 1;  // microfacet lobes: Plastic, Metal (and rough Glass)
constexpr int kFeatSpec = 2;   // perfectly specular lobes: smooth Glass, Mirror
constexpr int kFeatTwo = 4;    // two-lobe materials: Plastic (and rough Glass)
constexpr int kFeatTrans = 8;   // row f4: MicrofacetTransmission (rough Glass; comes with kFeatMicro | kFeatTwo)
constexpr int kFeatEnv = 16;    // row f4: the infinite light and its HDR map
constexpr int kNumFeatVariants = 9;
constexpr int kFeatVariants[kNumFeatVariants] = {0,
                                                 kFeatMicro,
                                                 kFeatSpec,
                                                 kFeatMicro | kFeatSpec,
                                                 kFeatMicro | kFeatSpec | kFeatTwo,
                                                 kFeatMicro | kFeatSpec | kFeatEnv,
                                                 kFeatMicro | kFeatSpec | kFeatTwo | kFeatEnv,
                                                 kFeatMicro | kFeatSpec | kFeatTwo | kFeatTrans,
                                                 kFeatMicro | kFeatSpec | kFeatTwo | kFeatTrans | kFeatEnv};
// instances without two-lobe materials are compiled for 3 waves/SIMD (kernels.hip)
constexpr bool feat_three_waves(int feat) { return (feat & (kFeatTwo | kFeatTrans)) == 0; }
template <int FEAT>
RTD D3 bxdf_f(const Lobe& l, D3 wo, D3 wi) {
    if (l.kind == LOBE_LAMBERT) return l.color * kInvPi;
    if ((FEAT & kFeatMicro) && l.kind == LOBE_MICROFACET) {
        double cos_o = absd(wo.z), cos_i = absd(wi.z);
        D3 wh = wi + wo;
        if (cos_i == 0.0 || cos_o == 0.0) return black();
        if (is_black(wh)) return black();
        wh = normalize(wh);
        D3 f = fresnel_evaluate(l, dot(wi, face_forward(wh, d3(0, 0, 1))));
        D3 comp1 = l.color * tr_d(l.alpha_x, l.alpha_y, wh) * tr_g(l.alpha_x, l.alpha_y, wo, wi);
        return cmul(comp1, f * (1.0 / (4.0 * cos_i * cos_o)));
    }
    if (((FEAT & kFeatTrans) != 0) && l.kind == LOBE_MICRO_TRANS) {  // bxdf.rs:393-441 (mode == RADIANCE); eta_a = p0, eta_b = p1
        if (same_hemisphere(wo, wi)) return black();
        const double cos_theta_o = wo.z, cos_theta_i = wi.z;
        if (cos_theta_i == 0.0 || cos_theta_o == 0.0) return black();
        const double eta = wo.z > 0.0 ? l.p1 / l.p0 : l.p0 / l.p1;
        D3 wh = normalize(wo + wi * eta);
        if (wh.z < 0.0) wh = -wh;
        if (dot(wo, wh) * dot(wi, wh) > 0.0) return black();
        const D3 f = fresnel_evaluate(l, dot(wo, wh));
        const double sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
        const double factor = 1.0 / eta;
        const D3 c = cmul(white() - f, l.color);
        return c * absd(tr_d(l.alpha_x, l.alpha_y, wh) * tr_g(l.alpha_x, l.alpha_y, wo, wi) * eta * eta *
                        absd(dot(wi, wh)) * absd(dot(wo, wh)) * factor * factor /
                        (cos_theta_i * cos_theta_o * sqrt_denom * sqrt_denom));
    }
    return black();
}
template <int FEAT>
RTD double bxdf_pdf(const Lobe& l, D3 wo, D3 wi) {
    if (l.kind == LOBE_LAMBERT || l.kind == LOBE_SPECULAR_REFL)
        return same_hemisphere(wo, wi) ? absd(wi.z) * kInvPi : 0.0;
    if ((FEAT & kFeatMicro) && l.kind == LOBE_MICROFACET) {
        if (!same_hemisphere(wo, wi)) return 0.0;
        D3 wh = normalize(wo + wi);
        return tr_pdf(l.alpha_x, l.alpha_y, wo, wh) / (4.0 * dot(wo, wh));
    }
    if (((FEAT & kFeatTrans) != 0) && l.kind == LOBE_MICRO_TRANS) {  // bxdf.rs:742-763
        if (same_hemisphere(wo, wi)) return 0.0;
        const double eta = wo.z > 0.0 ? l.p1 / l.p0 : l.p0 / l.p1;
        const D3 wh = normalize(wo + wi * eta);
        if (dot(wo, wh) * dot(wi, wh) > 0.0) return 0.0;
        const double sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
        const double dwh_dwi = absd(eta * eta * dot(wi, wh)) / (sqrt_denom * sqrt_denom);
        return tr_pdf(l.alpha_x, l.alpha_y, wo, wh) * dwh_dwi;
    }
    return 0.0;
}
// `rng` supplies default_sample_f's two entropy draws (bxdf.rs:815-827)
template <int FEAT>
RTD void bxdf_sample_f(const Lobe& l, D3 wo, double u0, double u1, uint64_t& rng, D3& f, D3& wi, double& pdf) {
    f = black();
    wi = black();
    pdf = 0.0;
    if (l.kind == LOBE_LAMBERT) {
        double r1 = rng_next(rng);
        double r2 = rng_next(rng);
        wi = rand_cosine_dir(r1, r2);
        if (wo.z < 0.0) wi.z *= -1.0;
        pdf = bxdf_pdf<FEAT>(l, wo, wi);
        f = bxdf_f<FEAT>(l, wo, wi);
    } else if ((FEAT & kFeatMicro) && l.kind == LOBE_MICROFACET) {
        if (wo.z == 0.0) return;
        D3 wh = tr_sample_wh(l.alpha_x, l.alpha_y, wo, u0, u1);
        D3 w2 = reflect(wo, wh);  // Q14: the wo.wh < 0 early-out is a no-op (bxdf.rs:598-600)
        if (!same_hemisphere(wo, w2)) return;
        wi = w2;
        pdf = tr_pdf(l.alpha_x, l.alpha_y, wo, wh) / (4.0 * dot(wo, wh));
        f = bxdf_f<FEAT>(l, wo, wi);
    } else if ((FEAT & kFeatSpec) && l.kind == LOBE_FRESNEL_SPECULAR) {  // Q15
        double fr = fr_dielectric(wo.z / norm(wo), l.p0, l.p1);
        if (u0 < fr) {
            wi = d3(-wo.x, -wo.y, wo.z);
            pdf = fr;
            f = l.color * fr;
            return;
        }
        bool entering = wo.z > 0.0;
        double eta_i = entering ? l.p0 : l.p1;
        double eta_t = entering ? l.p1 : l.p0;
        D3 dirv;
        if (refract(wo, face_forward(d3(0, 0, 1), wo), eta_i / eta_t, dirv)) {
            D3 ft = l.t * (1.0 - fr);
            ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));  // mode == RADIANCE
            f = ft;
            wi = dirv;
            pdf = 1.0 - fr;
        }
    } else if (((FEAT & kFeatTrans) != 0) && l.kind == LOBE_MICRO_TRANS) {  // bxdf.rs:608-638
        if (wo.z == 0.0) return;
        const D3 wh = tr_sample_wh(l.alpha_x, l.alpha_y, wo, u0, u1);
        if (dot(wo, wh) < 0.0) return;
        const double eta = wo.z > 0.0 ? l.p0 / l.p1 : l.p1 / l.p0;
        D3 t;
        if (!refract(wo, wh, eta, t)) return;
        wi = t;
        pdf = bxdf_pdf<FEAT>(l, wo, wi);
        f = bxdf_f<FEAT>(l, wo, wi);
    } else if (FEAT & kFeatSpec) {  // LOBE_SPECULAR_REFL, bxdf.rs:543-552
        wi = d3(-wo.x, -wo.y, wo.z);
        f = cmul(l.color, fresnel_evaluate(l, wi.z));
        pdf = 1.0;
    }
}

// The lobe array is only ever indexed with compile-time constants (or through pick_lobe's
// field-wise select) so that it lives in VGPRs; a run-time index would push it to scratch.
template <int FEAT>
RTD Lobe pick_lobe(const Bsdf& b, int i) {
    const Lobe& a = b.lobes[0];
    const Lobe& c = b.lobes[1];
    if (!(FEAT & kFeatTwo) || i == 0) return a;
    return c;
}

RTD D3 w2l(const Bsdf& b, D3 v) { return d3(dot(v, b.ss), dot(v, b.ts), dot(v, b.ns)); }
RTD D3 l2w(const Bsdf& b, D3 v) {
    return d3(b.ss.x * v.x + b.ts.x * v.y + b.ns.x * v.z, b.ss.y * v.x + b.ts.y * v.y + b.ns.y * v.z,
              b.ss.z * v.x + b.ts.z * v.y + b.ns.z * v.z);
}
template <int FEAT>
RTD int num_components(const Bsdf& b, uint32_t flags) {
    int c = 0;
    if (b.n > 0 && matches_flags(b.lobes[0].type, flags)) c++;
    if ((FEAT & kFeatTwo) && b.n > 1 && matches_flags(b.lobes[1].type, flags)) c++;
    return c;
}
template <int FEAT>
RTD D3 bsdf_f(const Bsdf& b, D3 wow, D3 wiw, uint32_t flags) {  // bsdf.rs:83-98 (Q10)
    D3 wi = w2l(b, wiw), wo = w2l(b, wow);
    bool refl = dot(wiw, b.ng) * dot(wow, b.ng) > 0.0;
    D3 f = black();
#pragma unroll
    for (int i = 0; i < ((FEAT & kFeatTwo) ? 2 : 1); i++) {
        if (i >= b.n) break;
        const Lobe& l = b.lobes[i];
        if ((matches_flags(l.type, flags) && (refl && (l.type & RT_BSDF_REFLECTION) > 0)) ||
            (!refl && (l.type & RT_BSDF_TRANSMISSION) > 0))
            f = f + bxdf_f<FEAT>(l, wo, wi);
    }
    return f;
}
template <int FEAT>
RTD double bsdf_pdf(const Bsdf& b, D3 wow, D3 wiw, uint32_t flags) {  // bsdf.rs:166-189 (Q11)
    int nc = num_components<FEAT>(b, RT_BSDF_ALL);
    if (nc == 0) return 0.0;
    D3 wo = w2l(b, wow), wi = w2l(b, wiw);
    if (wo.z == 0.0) return 0.0;
    double pdf = 0.0;
    int matching = 0;
#pragma unroll
    for (int i = 0; i < ((FEAT & kFeatTwo) ? 2 : 1); i++) {
        if (i >= nc) break;
        if (matches_flags(b.lobes[i].type, flags)) {
            matching++;
            pdf += bxdf_pdf<FEAT>(b.lobes[i], wo, wi);
        }
    }
    return matching > 0 ? pdf : 0.0;
}
template <int FEAT>
RTD void bsdf_sample_f(const Bsdf& b, D3 wow, double u0, double u1, uint32_t type, uint64_t& rng, D3& color, D3& wiw,
                       double& pdf, uint32_t& sampled) {  // bsdf.rs:102-164
    int matching = num_components<FEAT>(b, type);
    color = black();
    wiw = black();
    pdf = 0.0;
    sampled = 0;
    if (matching == 0) return;
    int comp_ = (int)(uint32_t)__builtin_floor(u0 * (double)matching);
    if (comp_ > matching - 1) comp_ = matching - 1;
    int count = comp_, used = 0;
#pragma unroll
    for (int i = 0; i < ((FEAT & kFeatTwo) ? 2 : 1); i++) {
        if (i >= b.n) break;
        if (matches_flags(b.lobes[i].type, type)) {
            if (count == 0) {
                used = i;
                break;
            }
            count--;
        }
    }
    const Lobe l = pick_lobe<FEAT>(b, used);
    D3 wo = normalize(w2l(b, wow));
    if (wo.z == 0.0) return;
    D3 f, wi;
    double p;
    bxdf_sample_f<FEAT>(l, wo, u0, u1, rng, f, wi, p);
    if (p == 0.0) return;
    D3 wiw_ = l2w(b, wi);
    if ((l.type & RT_BSDF_SPECULAR) == 0 && matching > 1) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            if (i >= b.n) break;
            if (i != used && matches_flags(b.lobes[i].type, type)) p += bxdf_pdf<FEAT>(b.lobes[i], wo, wi);
        }
    }
    if (matching > 1) p = p / (double)matching;
    if ((l.type & RT_BSDF_SPECULAR) == 0) {
        bool refl = dot(wiw_, b.ng) * dot(wow, b.ng) > 0.0;
        f = black();
#pragma unroll
        for (int i = 0; i < 2; i++) {
            if (i >= b.n) break;
            const Lobe& li = b.lobes[i];
            if (matches_flags(li.type, type) && ((refl && (li.type & RT_BSDF_REFLECTION) > 0) ||
                                                 (!refl && (li.type & RT_BSDF_TRANSMISSION) > 0)))
                f = f + bxdf_f<FEAT>(li, wo, wi);
        }
    }
    color = f;
    wiw = wiw_;
    pdf = p;
    sampled = l.type;
}

RTD void bsdf_init(Bsdf& b, const HitRec& h) {  // bsdf.rs:26-35
    b.ns = h.sh_n;
    b.ng = h.n;
    b.ss = h.sh_dpdu;
    b.ts = normalize(cross(h.sh_n, h.sh_dpdu));
    b.n = 0;
}
RTD Lobe lobe_zero() {
    Lobe l;
    l.kind = 0;
    l.type = 0;
    l.color = black();
    l.t = black();
    l.k = black();
    l.fresnel = FR_NOOP;
    l.p0 = l.p1 = 0.0;
    l.alpha_x = l.alpha_y = 0.0;
    return l;
}
RTD Lobe make_lambert(D3 c) {
    Lobe l = lobe_zero();
    l.kind = LOBE_LAMBERT;
    l.type = RT_BSDF_REFLECTION | RT_BSDF_DIFFUSE;
    l.color = c;
    return l;
}
RTD Lobe make_microfacet(D3 c, double ax, double ay) {  // + microfacet.rs:340-348 clamp
    Lobe l = lobe_zero();
    l.kind = LOBE_MICROFACET;
    l.type = RT_BSDF_REFLECTION | RT_BSDF_GLOSSY;
    l.color = c;
    l.alpha_x = rmax(ax, 1e-3);
    l.alpha_y = rmax(ay, 1e-3);
    return l;
}

// texture k of a material: the embedded colour when the texture is a solid one (same value as Texture::value)
RTD D3 mat_tex(const DevScene& sc, const DevMat& dm, int k, double u, double v) {
    if (dm.solid_mask & (1u << k)) return d3(dm.col[k][0], dm.col[k][1], dm.col[k][2]);
    return texture_value(sc, dm.tex[k], u, v);
}

// material.rs:80-244 with mode = RADIANCE, allow_lobes = true
template <int FEAT>
RTD void compute_scattering(const DevScene& sc, const HitRec& h, Bsdf& b) {
    const DevMat& dm = sc.mats[h.mat];
    const rt_material& m = dm.m;
    b.n = 0;
    b.lobes[0] = lobe_zero();
    b.lobes[1] = lobe_zero();
    if (m.kind == RT_MAT_MATTE) {
        D3 color = mat_tex(sc, dm, 0, h.u, h.v);
        if (!is_black(color)) {
            bsdf_init(b, h);
            b.lobes[0] = make_lambert(color);
            b.n = 1;
        }
    } else if ((FEAT & kFeatTwo) && m.kind == RT_MAT_PLASTIC) {
        D3 color = mat_tex(sc, dm, 0, h.u, h.v);
        bool inited = false;
        if (!is_black(color)) {
            bsdf_init(b, h);
            inited = true;
            b.lobes[0] = make_lambert(color);
            b.n = 1;
        }
        D3 spec = mat_tex(sc, dm, 1, h.u, h.v);
        if (!is_black(spec)) {
            if (!inited) bsdf_init(b, h);
            double rough = m.f[0];
            if (m.remap_roughness) rough = tr_roughness_to_alpha(rough);
            Lobe l = make_microfacet(spec, rough, rough);
            l.fresnel = FR_DIELECTRIC;
            l.p0 = 1.5;
            l.p1 = 1.0;
            if (b.n == 0)
                b.lobes[0] = l;
            else
                b.lobes[1] = l;
            b.n++;
        }
    } else if ((FEAT & kFeatSpec) && m.kind == RT_MAT_GLASS) {
        D3 r = mat_tex(sc, dm, 0, h.u, h.v);
        D3 t = mat_tex(sc, dm, 1, h.u, h.v);
        bsdf_init(b, h);
        if (!(is_black(r) && is_black(t))) {
            double urough = m.f[0], vrough = m.f[1];
            if (!((FEAT & kFeatTrans) != 0) || (urough == 0.0 && vrough == 0.0)) {  // is_specular && allow_lobes
                Lobe l = lobe_zero();
                l.kind = LOBE_FRESNEL_SPECULAR;
                l.type = RT_BSDF_TRANSMISSION | RT_BSDF_REFLECTION | RT_BSDF_SPECULAR;
                l.color = r;
                l.t = t;
                l.p0 = m.f[2];
                l.p1 = 1.0;
                b.lobes[0] = l;
                b.n = 1;
            } else {  // material.rs:161-189: MicrofacetReflection + MicrofacetTransmission
                if (m.remap_roughness) {
                    urough = tr_roughness_to_alpha(urough);
                    vrough = tr_roughness_to_alpha(vrough);
                }
                if (!is_black(r)) {
                    Lobe l = make_microfacet(r, urough, vrough);
                    l.fresnel = FR_DIELECTRIC;
                    l.p0 = m.f[2];
                    l.p1 = 1.0;
                    b.lobes[0] = l;
                    b.n = 1;
                }
                if (!is_black(t)) {
                    Lobe l = make_microfacet(t, urough, vrough);
                    l.kind = LOBE_MICRO_TRANS;
                    l.type = RT_BSDF_TRANSMISSION | RT_BSDF_GLOSSY;
                    l.fresnel = FR_DIELECTRIC;
                    l.p0 = m.f[2];
                    l.p1 = 1.0;
                    if (b.n == 0)
                        b.lobes[0] = l;
                    else
                        b.lobes[1] = l;
                    b.n++;
                }
            }
        }
    } else if ((FEAT & kFeatMicro) && m.kind == RT_MAT_METAL) {
        bsdf_init(b, h);
        D3 ur = mat_tex(sc, dm, 3, h.u, h.v);
        D3 vr = mat_tex(sc, dm, 4, h.u, h.v);
        double ua = m.remap_roughness ? tr_roughness_to_alpha(ur.x) : ur.x;
        double va = m.remap_roughness ? tr_roughness_to_alpha(vr.x) : vr.x;
        Lobe l = make_microfacet(white(), ua, va);
        l.fresnel = FR_CONDUCTOR;
        l.t = mat_tex(sc, dm, 0, h.u, h.v);
        l.k = mat_tex(sc, dm, 1, h.u, h.v);
        b.lobes[0] = l;
        b.n = 1;
    } else if ((FEAT & kFeatSpec) && m.kind == RT_MAT_MIRROR) {
        bsdf_init(b, h);
        D3 color = mat_tex(sc, dm, 0, h.u, h.v);
        if (!is_black(color)) {
            Lobe l = lobe_zero();
            l.kind = LOBE_SPECULAR_REFL;
            l.type = RT_BSDF_REFLECTION | RT_BSDF_SPECULAR;
            l.color = color;
            l.fresnel = FR_NOOP;
            b.lobes[0] = l;
            b.n = 1;
        }
    }
    // RT_MAT_LIGHT: no lobes (material.rs:102)
}

// ------------------------------------------------------------------ lights
RTD double prim_area(const DevScene& sc, const rt_primitive& pr) {  // primitive.rs:339-359
    if (pr.kind == RT_PRIM_SPHERE) return 2.0 * kPi * pr.v[3];
    if (pr.kind == RT_PRIM_TRIANGLE) {
        D3 p0, p1, p2;
        uint32_t i1, i2, i3;
        load_tri(sc, pr, p0, p1, p2, i1, i2, i3);
        return 0.5 * norm(cross(p1 - p0, p2 - p0));
    }
    return (pr.v[2] - pr.v[0]) * (pr.v[3] - pr.v[1]);
}
RTD D3 uniform_sample_sphere(double u0, double u1) {  // util.rs:51-56
    double z = 1.0 - 2.0 * u0;
    double r = dm_sqrt(rmax(0.0, 1.0 - z * z));
    double phi = 2.0 * kPi * u1;
    const SinCos sc_ = sincos2(phi);
    return d3(r * sc_.c, r * sc_.s, z);
}
RTD void sample_area(const DevScene& sc, const rt_primitive& pr, double u0, double u1, D3& p, D3& n, double& pdf) {
    if (pr.kind == RT_PRIM_SPHERE) {
        p = pr.v[3] * uniform_sample_sphere(u0, u1);
        n = normalize(p);
    } else if (pr.kind == RT_PRIM_TRIANGLE) {
        D3 p0, p1, p2;
        uint32_t i0, i1, i2;
        load_tri(sc, pr, p0, p1, p2, i0, i1, i2);
        const DevMesh& m = sc.meshes[pr.mesh_index];
        double s0 = dm_sqrt(u0);
        double b0 = 1.0 - s0, b1 = u1 * s0;
        p = (b0 * p0 + b1 * p1 + (1.0 - b0 - b1) * p2);
        if (m.n) {
            D3 n0 = d3(m.n[3 * i0], m.n[3 * i0 + 1], m.n[3 * i0 + 2]);
            D3 n1 = d3(m.n[3 * i1], m.n[3 * i1 + 1], m.n[3 * i1 + 2]);
            D3 n2 = d3(m.n[3 * i2], m.n[3 * i2 + 1], m.n[3 * i2 + 2]);
            n = normalize(b0 * n0 + b1 * n1 + (1.0 - b0 - b1) * n2);
        } else {
            n = normalize(cross(p1 - p0, p2 - p0));
        }
    } else if (pr.kind == RT_PRIM_XY_RECT) {
        n = d3(0, 0, 1);
        p = d3(pr.v[0] + u0 * (pr.v[2] - pr.v[0]), pr.v[1] + u1 * (pr.v[3] - pr.v[1]), pr.v[4]);
    } else if (pr.kind == RT_PRIM_XZ_RECT) {
        n = d3(0, 1, 0);
        p = d3(pr.v[0] + u0 * (pr.v[2] - pr.v[0]), pr.v[4], pr.v[1] + u1 * (pr.v[3] - pr.v[1]));
    } else {
        n = d3(1, 0, 0);
        p = d3(pr.v[4], pr.v[0] + u0 * (pr.v[2] - pr.v[0]), pr.v[1] + u1 * (pr.v[3] - pr.v[1]));
    }
    pdf = 1.0 / prim_area(sc, pr);
    if (pr.flip) n = -n;
}
RTD D3 light_l(const rt_light& lt, D3 n, D3 w) {  // light.rs:475-496
    if (dot(n, w) > 0.0 || lt.two_sided) return d3(lt.color[0], lt.color[1], lt.color[2]);
    return black();
}
RTD double prim_pdf(const DevScene& sc, const rt_primitive& pr, D3 rec_p, D3 dir) {  // primitive.rs:462-473
    if (pr.kind >= RT_PRIM_XY_RECT && pr.xform_index < 0) {
        // the usual emitter, an axis-aligned rect: intersects_obj's record is only read for its point and for
        // |dot(n, -dir)|; n is a signed unit axis (rect_record), so the dot product is dir's component along that
        // axis, exactly -- the same value without building the record (two normalisations, two uv divisions)
        double t, a, b;
        D3 to, td;
        if (!rect_core(sc, pr, rec_p, dir, 0.0, kInf, t, a, b, to, td)) return 0.0;
        const D3 dist = rec_p - (to + td * t);
        return norm2(dist) / (prim_area(sc, pr) * absd(rect_axis_comp(pr.kind, dir)));
    }
    HitRec nh;
    if (!intersects_obj(sc, pr, rec_p, dir, 0.0, kInf, nh)) return 0.0;
    D3 dist = rec_p - nh.p;
    return norm2(dist) / (prim_area(sc, pr) * absd(dot(nh.n, -dir)));
}
// ------------------------------------------------------ Light::Infinite
// distribution.rs:152-166 find_interval over cdf[0..size) with pred = cdf[i] <= u
RTD uint32_t find_interval(const f64_t* __restrict__ cdf, uint32_t size, double u) {
    uint32_t first = 0, len = size;
    while (len > 0) {
        const uint32_t half = len >> 1, middle = first + half;
        if (cdf[middle] <= u) {
            first = middle + 1;
            len = len - half - 1;
        } else {
            len = half;
        }
    }
    // clamp((first - 1) as f64, 0, size - 2) as usize; first == 0 (u < 0 / NaN) wraps to usize::MAX in release
    const double x = first == 0 ? 18446744073709551615.0 : (double)(first - 1);
    return sat_u32(clampd(x, 0.0, (double)(size - 2)));
}
// distribution.rs:64-78 sample_continuous
RTD void dist1d_sample(const f64_t* __restrict__ func, const f64_t* __restrict__ cdf, uint32_t n, double func_int,
                       double u, double& x, double& pdf, uint32_t& offset) {
    offset = find_interval(cdf, n + 1, u);
    double du = u - cdf[offset];
    if (cdf[offset + 1] - cdf[offset] > 0.0) du = du / (cdf[offset + 1] - cdf[offset]);
    pdf = func_int > 0.0 ? func[offset] / func_int : 0.0;
    x = ((double)offset + du) / (double)n;
}
RTD void dist2d_sample(const DevEnv& e, double u0, double u1, double& x0, double& x1, double& pdf) {  // :133-137
    double pdf1, pdf0;
    uint32_t v, off;
    dist1d_sample(e.marg_func, e.marg_cdf, e.nv, e.marg_int, u1, x1, pdf1, v);
    dist1d_sample(e.img + v, e.cond_cdf + (size_t)v * (e.nu + 1), e.nu, e.marg_func[v], u0, x0, pdf0, off);
    pdf = pdf1 * pdf0;
}
RTD double dist2d_pdf(const DevEnv& e, double p0, double p1) {  // distribution.rs:139-145
    uint32_t iu = sat_u32(p0 * (double)e.nu);
    if (iu > e.nu - 1) iu = e.nu - 1;
    uint32_t iv = sat_u32(p1 * (double)e.nv);
    if (iv > e.nv - 1) iv = e.nv - 1;
    return e.img[(size_t)iv + iu] / e.marg_int;
}
RTD double spherical_theta(D3 v) { return dm_acos(clampd(v.y, -1.0, 1.0)); }  // util.rs:153-155
RTD double spherical_phi(D3 v) {                                              // util.rs:160-167
    const double p = dm_atan2(v.z, v.x);
    return p < 0.0 ? p + 2.0 * kPi : p;
}
RTD D3 light_to_world(const DevScene& sc, const rt_light& lt, D3 v) {
    return lt.xform_index >= 0 ? xf_vector(sc.xforms[lt.xform_index].fwd, v) : v;
}
RTD D3 light_to_obj(const DevScene& sc, const rt_light& lt, D3 v) {
    return lt.xform_index >= 0 ? xf_vector(sc.xforms[lt.xform_index].inv, v) : v;
}
// light.rs:499-512 Light::le of the infinite light
RTDN D3 infinite_le(const DevScene& sc, const rt_light& lt, D3 dir) {
    const D3 w = normalize(light_to_obj(sc, lt, dir));
    const double kInv2Pi = 1.0 / (2.0 * kPi);
    return texture_value(sc, lt.tex_index, spherical_phi(w) * kInv2Pi, spherical_theta(w) * kInvPi);
}
// light.rs:285-294 pdf_li (through to_world, as the reference does)
RTDN double infinite_pdf_li(const DevScene& sc, const rt_light& lt, D3 wi) {
    const D3 w = light_to_world(sc, lt, wi);
    const double theta = spherical_theta(w), phi = spherical_phi(w);
    const double sin_theta = dm_sin(theta);
    if (sin_theta == 0.0) return 0.0;
    const double kInv2Pi = 1.0 / (2.0 * kPi);
    return dist2d_pdf(sc.env, phi * kInv2Pi, theta * kInvPi) / (2.0 * kPi * kPi * sin_theta);
}
// light.rs:204-245 sample_li: direction (normalised), pdf, radiance and the far end of the Visibility segment
RTDN void infinite_sample_li(const DevScene& sc, const rt_light& lt, D3 p, double u0, double u1, D3& wi, double& pdf,
                             D3& color, D3& far_end) {
    double uv0, uv1, map_pdf;
    dist2d_sample(sc.env, u0, u1, uv0, uv1, map_pdf);
    if (map_pdf == 0.0) {
        wi = black();
        pdf = 0.0;
        color = black();
        far_end = black();
        return;
    }
    const double theta = uv1 * kPi, phi = uv0 * 2.0 * kPi;
    const SinCos sct = sincos2(theta), scp = sincos2(phi);
    const double cos_theta = sct.c, sin_theta = sct.s, cos_phi = scp.c, sin_phi = scp.s;
    const D3 v = d3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
    const D3 wiv = light_to_world(sc, lt, v);
    pdf = map_pdf / (2.0 * kPi * kPi * sin_theta);
    if (sin_theta == 0.0) pdf = 0.0;
    far_end = p + wiv * (2.0 * lt.world_radius);
    color = texture_value(sc, lt.tex_index, uv0, uv1);
    wi = normalize(wiv);
}

RTD double power_heuristic(int nf, double f_pdf, int ng, double g_pdf) {  // integrator.rs:655-659
    double f = (double)nf * f_pdf, g = (double)ng * g_pdf;
    return (f * f) / (f * f + g * g);
}

}  // namespace rtd

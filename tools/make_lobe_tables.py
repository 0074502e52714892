#!/usr/bin/env python3
"""Independent value tables for the BxDF lobes on the hot path -> tests/golden/lobe_tables.json.

Every number here is computed with mpmath (50 significant digits) from formulas transcribed
directly from the reference's Rust text -- NOT from oracle/ and NOT from the device headers:
    Fresnel            /root/reference/src/bxdf.rs:113-211
    Bxdf::f            src/bxdf.rs:328-441          Bxdf::sample_f   src/bxdf.rs:532-684
    Bxdf::pdf          src/bxdf.rs:721-763          default_*        src/bxdf.rs:815-835
    trig helpers       src/bxdf.rs:12-56
    Trowbridge-Reitz   src/microfacet.rs:53-68, 109-123, 143-172, 240-282, 442-512
    reflect / refract / face_forward / same_hemisphere / rand_cosine_dir   src/util.rs:62-94, 203-206, 376-385, 576-593
    PI (truncated)     src/consts.rs:31
    counter RNG        include/rt_abi.h (the ABI's replacement for the reference's unseeded streams)
The structure differs from the oracle on purpose (one closure per lobe over mp vectors), so a misreading
would have to be made twice, in two shapes, to go unnoticed.  tests/test_lobe_tables.py checks the oracle's
lobes against these tables; the GPU is bit-identical to the oracle in every render test.

Run in the build container:  python tools/make_lobe_tables.py
"""
import json
import os
import random

from mpmath import mp, mpf, sqrt, sin, cos, tan, atan, log, fabs

mp.dps = 50
PI = mpf("3.14159265358979")  # consts.rs:31
INV_PI = 1 / PI
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ----------------------------------------------------------------------------- small vector kit
def V(x, y, z):
    return (mpf(x), mpf(y), mpf(z))


def add(a, b): return tuple(x + y for x, y in zip(a, b))
def sub(a, b): return tuple(x - y for x, y in zip(a, b))
def scale(a, s): return tuple(x * s for x in a)
def mul(a, b): return tuple(x * y for x, y in zip(a, b))
def dot(a, b): return sum(x * y for x, y in zip(a, b))
def neg(a): return tuple(-x for x in a)
def length(a): return sqrt(dot(a, a))
def unit(a): return scale(a, 1 / length(a))
BLACK = V(0, 0, 0)
WHITE = V(1, 1, 1)


def clamp(x, lo, hi):
    return lo if x < lo else (hi if x > hi else x)


# bxdf.rs:12-56
def cos_theta(v): return v[2]
def cos2_theta(v): return v[2] * v[2]
def abs_cos_theta(v): return fabs(v[2])
def sin2_theta(v): return max(mpf(0), 1 - cos2_theta(v))
def sin_theta(v): return sqrt(sin2_theta(v))
def tan_theta(v): return sin_theta(v) / cos_theta(v)
def tan2_theta(v): return sin2_theta(v) / cos2_theta(v)


def cos_phi(v):
    s = sin_theta(v)
    return mpf(1) if s == 0 else clamp(v[0] / s, -1, 1)


def sin_phi(v):
    s = sin_theta(v)
    return mpf(1) if s == 0 else clamp(v[1] / s, -1, 1)  # (sic: 1, bxdf.rs:49)


def same_hemisphere(v, w): return v[2] * w[2] > 0            # util.rs:589-591
def face_forward(n, v): return neg(n) if dot(n, v) < 0 else n  # util.rs:576-579
def reflect(v, n): return add(neg(v), scale(n, 2 * dot(v, n)))  # util.rs:203-206


def refract(vec, n, eta):  # util.rs:376-385
    cos_i = dot(n, vec) / length(vec)
    sin2_i = max(mpf(0), 1 - cos_i * cos_i)
    sin2_t = eta * eta * sin2_i
    if sin2_t >= 1:
        return None
    cos_t = sqrt(1 - sin2_t)
    return add(scale(neg(vec), eta), scale(n, eta * cos_i - cos_t))


# ----------------------------------------------------------------------------------- Fresnel
def fr_dielectric(cos_i, eta_i, eta_t):  # bxdf.rs:113-136
    cos_i = clamp(cos_i, -1, 1)
    ii, it = eta_i, eta_t
    if cos_i < 0:
        ii, it = eta_t, eta_i
        cos_i = fabs(cos_i)
    sin_i = sqrt(max(mpf(0), 1 - cos_i * cos_i))
    sin_t = ii / it * sin_i
    cos_t = sqrt(max(mpf(0), 1 - sin_t * sin_t))
    if sin_t >= 1:
        return mpf(1)
    r_parl = ((it * cos_i) - (ii * cos_t)) / ((it * cos_i) + (ii * cos_t))
    r_perp = ((ii * cos_i) - (it * cos_t)) / ((ii * cos_i) + (it * cos_t))
    return (r_parl * r_parl + r_perp * r_perp) / 2


def fr_conductor(cos_i, eta, k):  # bxdf.rs:141-170, per channel
    cos_i = clamp(cos_i, -1, 1)
    c2 = cos_i * cos_i
    s2 = 1 - c2
    out = []
    for e, kk in zip(eta, k):
        e2, k2 = e * e, kk * kk
        t0 = e2 - k2 - s2
        a2b2 = sqrt(t0 * t0 + 4 * e2 * k2)
        t1 = a2b2 + c2
        a = sqrt((a2b2 + t0) / 2)
        t2 = 2 * cos_i * a
        rs = (t1 - t2) / (t1 + t2)
        t3 = c2 * a2b2 + s2 * s2
        t4 = t2 * s2
        rp = rs * (t3 - t4) / (t3 + t4)
        out.append((rp + rs) / 2)
    return tuple(out)


def fresnel(fr, cos_i):  # Fresnel::evaluate, bxdf.rs:190-211
    if fr["kind"] == "dielectric":  # note the swapped arguments (eta_t, eta_i)
        v = fr_dielectric(fabs(cos_i), fr["eta_t"], fr["eta_i"])
        return (v, v, v)
    if fr["kind"] == "conductor":
        return fr_conductor(fabs(cos_i), fr["eta"], fr["k"])
    return WHITE


# ------------------------------------------------------------------------- Trowbridge-Reitz
def tr_d(ax, ay, wh):  # microfacet.rs:53-68
    t2 = tan2_theta(wh)
    if cos2_theta(wh) == 0:
        return mpf(0)
    c4 = cos2_theta(wh) ** 2
    e = (cos_phi(wh) ** 2 / (ax * ax) + sin_phi(wh) ** 2 / (ay * ay)) * t2
    return 1 / (PI * ax * ay * c4 * (1 + e) * (1 + e))


def tr_lambda(ax, ay, w):  # microfacet.rs:109-123
    if cos_theta(w) == 0:
        return mpf(0)
    at = fabs(tan_theta(w))
    alpha = sqrt(cos_phi(w) ** 2 * ax * ax + sin_phi(w) ** 2 * ay * ay)
    x = (alpha * at) * (alpha * at)
    return (-1 + sqrt(1 + x)) / 2


def tr_g1(ax, ay, w): return 1 / (1 + tr_lambda(ax, ay, w))                        # :143-145
def tr_g(ax, ay, wo, wi): return 1 / (1 + tr_lambda(ax, ay, wo) + tr_lambda(ax, ay, wi))  # :146-157
def tr_pdf(ax, ay, wo, wh):                                                         # :159-172, visible area
    return tr_d(ax, ay, wh) * tr_g1(ax, ay, wo) * fabs(dot(wo, wh)) / abs_cos_theta(wo)


def tr_sample_11(ct, u1, u2):  # microfacet.rs:469-512
    if ct > mpf("0.9999"):
        r = sqrt(u1 / (1 - u1))
        phi = 2 * PI * u2
        return r * cos(phi), r * sin(phi)
    st = max(mpf(0), sqrt(1 - ct * ct))
    tt = st / ct
    a = 1 / tt
    g1 = 2 / (1 + sqrt(1 + 1 / (a * a)))
    a = 2 * u1 / g1 - 1
    tmp = 1 / (a * a - 1)
    if tmp > mpf("1e10"):
        tmp = mpf("1e10")
    b = tt
    d = sqrt(max(mpf(0), b * b * tmp * tmp - (a * a - b * b) * tmp))
    sx1, sx2 = b * tmp - d, b * tmp + d
    sx = sx1 if (a < 0 or sx2 > 1 / tt) else sx2
    if u2 > mpf("0.5"):
        s, nu2 = mpf(1), 2 * (u2 - mpf("0.5"))
    else:
        s, nu2 = mpf(-1), 2 * (mpf("0.5") - u2)
    z = (nu2 * (nu2 * (nu2 * mpf("0.27385") - mpf("0.73369")) + mpf("0.46341"))) / \
        (nu2 * (nu2 * (nu2 * mpf("0.093073") + mpf("0.309420")) - 1) + mpf("0.597999"))
    return sx, s * z * sqrt(1 + sx * sx)


def tr_sample(wi, ax, ay, u1, u2):  # microfacet.rs:442-467
    ws = unit((ax * wi[0], ay * wi[1], wi[2]))
    sx, sy = tr_sample_11(cos_theta(ws), u1, u2)
    sp, cp = sin_phi(ws), cos_phi(ws)
    tmp = cp * sx - sp * sy
    sy = sp * sx + cp * sy
    sx = tmp
    return unit((-(ax * sx), -(ay * sy), mpf(1)))


def tr_sample_wh(ax, ay, wo, u0, u1):  # microfacet.rs:274-282 (sample_visible_area = true)
    flip = wo[2] < 0
    wh = tr_sample(neg(wo) if flip else wo, ax, ay, u0, u1)
    return neg(wh) if flip else wh


def tr_roughness_to_alpha(r):  # microfacet.rs:436-440
    r = max(mpf(r), mpf("1e-5"))
    x = log(r)
    return mpf("1.62142") + mpf("0.819955") * x + mpf("0.1734") * x * x + mpf("0.0171201") * x ** 3 + \
        mpf("0.000640711") * x ** 4


# ------------------------------------------------------------------------------------- lobes
def lambert(color):
    def f(wo, wi): return scale(color, INV_PI)                         # bxdf.rs:337
    def pdf(wo, wi): return abs_cos_theta(wi) * INV_PI if same_hemisphere(wo, wi) else mpf(0)  # :829-835

    def sample(wo, u0, u1, r1, r2):                                      # default_sample_f :815-827
        a, b = 2 * r1 - 1, 2 * r2 - 1                                   # rand_cosine_dir util.rs:62-82
        if a == 0 and b == 0:
            wi = V(0, 0, 1)
        else:
            if fabs(a) > fabs(b):
                r, th = a, PI / 4 * (b / a)
            else:
                r, th = b, PI / 2 - PI / 4 * (a / b)
            x, y = r * cos(th), r * sin(th)
            wi = (x, y, sqrt(max(mpf(0), 1 - x * x - y * y)))
        if wo[2] < 0:
            wi = (wi[0], wi[1], -wi[2])
        return f(wo, wi), wi, pdf(wo, wi)
    return f, pdf, sample


def microfacet_reflection(color, fr, ax, ay):
    def f(wo, wi):  # bxdf.rs:366-392
        co, ci = abs_cos_theta(wo), abs_cos_theta(wi)
        wh = add(wi, wo)
        if ci == 0 or co == 0 or wh == BLACK:
            return BLACK
        wh = unit(wh)
        F = fresnel(fr, dot(wi, face_forward(wh, V(0, 0, 1))))
        comp = scale(color, tr_d(ax, ay, wh) * tr_g(ax, ay, wo, wi))
        return mul(comp, scale(F, 1 / (4 * ci * co)))

    def pdf(wo, wi):  # bxdf.rs:738-744
        if not same_hemisphere(wo, wi):
            return mpf(0)
        wh = unit(add(wo, wi))
        return tr_pdf(ax, ay, wo, wh) / (4 * dot(wo, wh))

    def sample(wo, u0, u1, r1, r2):  # bxdf.rs:590-607 (the wo.wh < 0 branch has no `return`)
        if wo[2] == 0:
            return BLACK, BLACK, mpf(0)
        wh = tr_sample_wh(ax, ay, wo, u0, u1)
        wi = reflect(wo, wh)
        if not same_hemisphere(wo, wi):
            return BLACK, BLACK, mpf(0)
        return f(wo, wi), wi, tr_pdf(ax, ay, wo, wh) / (4 * dot(wo, wh))
    return f, pdf, sample


def microfacet_transmission(color, ax, ay, eta_a, eta_b):
    fr = {"kind": "dielectric", "eta_i": eta_a, "eta_t": eta_b}  # make_microfacet_transmission :943-950

    def f(wo, wi):  # bxdf.rs:393-441, mode == RADIANCE
        if same_hemisphere(wo, wi):
            return BLACK
        co, ci = cos_theta(wo), cos_theta(wi)
        if ci == 0 or co == 0:
            return BLACK
        eta = eta_b / eta_a if co > 0 else eta_a / eta_b
        wh = unit(add(wo, scale(wi, eta)))
        if wh[2] < 0:
            wh = neg(wh)
        if dot(wo, wh) * dot(wi, wh) > 0:
            return BLACK
        F = fresnel(fr, dot(wo, wh))
        sd = dot(wo, wh) + eta * dot(wi, wh)
        factor = 1 / eta
        c = mul(sub(WHITE, F), color)
        return scale(c, fabs(tr_d(ax, ay, wh) * tr_g(ax, ay, wo, wi) * eta * eta * fabs(dot(wi, wh)) *
                             fabs(dot(wo, wh)) * factor * factor / (ci * co * sd * sd)))

    def pdf(wo, wi):  # bxdf.rs:745-763
        if same_hemisphere(wo, wi):
            return mpf(0)
        eta = eta_b / eta_a if cos_theta(wo) > 0 else eta_a / eta_b
        wh = unit(add(wo, scale(wi, eta)))
        if dot(wo, wh) * dot(wi, wh) > 0:
            return mpf(0)
        sd = dot(wo, wh) + eta * dot(wi, wh)
        return tr_pdf(ax, ay, wo, wh) * fabs(eta * eta * dot(wi, wh)) / (sd * sd)

    def sample(wo, u0, u1, r1, r2):  # bxdf.rs:608-638
        if wo[2] == 0:
            return BLACK, BLACK, mpf(0)
        wh = tr_sample_wh(ax, ay, wo, u0, u1)
        if dot(wo, wh) < 0:
            return BLACK, BLACK, mpf(0)
        eta = eta_a / eta_b if cos_theta(wo) > 0 else eta_b / eta_a
        wi = refract(wo, wh, eta)
        if wi is None:
            return BLACK, BLACK, mpf(0)
        return f(wo, wi), wi, pdf(wo, wi)
    return f, pdf, sample


def fresnel_specular(r, t, eta_a, eta_b):
    def f(wo, wi): return BLACK          # bxdf.rs: arbitrary directions carry no specular energy
    def pdf(wo, wi): return mpf(0)       # bxdf.rs:780-793 ("unimplemented", returns 0)

    def sample(wo, u0, u1, r1, r2):      # bxdf.rs:640-684, mode == RADIANCE
        F = fr_dielectric(cos_theta(wo) / length(wo), eta_a, eta_b)
        if u0 < F:
            return scale(r, F), (-wo[0], -wo[1], wo[2]), F
        entering = cos_theta(wo) > 0
        ei, et = (eta_a, eta_b) if entering else (eta_b, eta_a)
        d = refract(wo, face_forward(V(0, 0, 1), wo), ei / et)
        if d is None:
            return BLACK, BLACK, mpf(0)
        return scale(scale(t, 1 - F), (ei * ei) / (et * et)), d, 1 - F
    return f, pdf, sample


def specular_reflection(color, fr):
    def f(wo, wi): return BLACK                                                   # bxdf.rs:334
    def pdf(wo, wi): return abs_cos_theta(wi) * INV_PI if same_hemisphere(wo, wi) else mpf(0)  # default_pdf :724

    def sample(wo, u0, u1, r1, r2):                                                 # bxdf.rs:543-552
        wi = (-wo[0], -wo[1], wo[2])
        return mul(color, fresnel(fr, cos_theta(wi))), wi, mpf(1)
    return f, pdf, sample


# --------------------------------------------------------------------------------- RNG (rt_abi.h)
M64 = (1 << 64) - 1
G, H, J = 0x9E3779B97F4A7C15, 0xD1B54A32D192ED03, 0x8CB92BA72F3D8DD7


def mix(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def rng_draws(seed, pixel, sample, n):
    s0 = mix((mix((seed * G + pixel) & M64) + sample * H + J) & M64)
    return [mpf(mix((s0 + (k + 1) * G) & M64) >> 11) / mpf(2) ** 53 for k in range(n)]


# -------------------------------------------------------------------------------------- tables
def fl(x): return float(x)
def fl3(v): return [float(x) for x in v]


def direction(rnd, hemi):
    """A unit vector exactly representable as three doubles after rounding (the oracle gets the same doubles)."""
    while True:
        z = rnd.uniform(0.03, 0.999) * hemi
        ph = rnd.uniform(0.0, 6.283)
        import math
        s = math.sqrt(max(0.0, 1 - z * z))
        v = [s * math.cos(ph), s * math.sin(ph), z]
        n = math.sqrt(sum(c * c for c in v))
        return [c / n for c in v]


def main():
    rnd = random.Random(20261004)
    lobes = []
    metal_alpha = tr_roughness_to_alpha(0.1)
    specs = [
        ("lambert", {"kind": 0, "fresnel": 2, "color": [0.73, 0.45, 0.12]}, None),
        ("microfacet_conductor_iso",  # C3's metal: eta (0.05,0.5,0.75), k 0, roughness 0.1 remapped (scenes.rs:582-606)
         {"kind": 1, "fresnel": 1, "color": [1.0, 1.0, 1.0], "eta": [0.05, 0.5, 0.75], "k": [0.0, 0.0, 0.0],
          "alpha_x": fl(metal_alpha), "alpha_y": fl(metal_alpha)}, None),
        ("microfacet_conductor_aniso",
         {"kind": 1, "fresnel": 1, "color": [1.0, 1.0, 1.0], "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14],
          "alpha_x": 0.35, "alpha_y": 0.08}, None),
        ("microfacet_dielectric_plastic",  # Plastic's specular lobe: FresnelDielectric{1.5, 1} (material.rs:118-121)
         {"kind": 1, "fresnel": 0, "color": [1.13, 1.13, 1.13], "eta_i": 1.5, "eta_t": 1.0,
          "alpha_x": 0.12, "alpha_y": 0.12}, None),
        ("microfacet_transmission",  # rough glass (material.rs:176-188): eta_a = index, eta_b = 1
         {"kind": 4, "fresnel": 0, "color": [0.9, 0.95, 1.0], "eta_i": 1.5, "eta_t": 1.0, "eta_a": 1.5, "eta_b": 1.0,
          "alpha_x": 0.3, "alpha_y": 0.15}, None),
        ("fresnel_specular",  # smooth glass (material.rs:153-157)
         {"kind": 2, "fresnel": 2, "color": [1.0, 0.9, 0.8], "t": [0.7, 0.8, 1.0], "eta_a": 1.5, "eta_b": 1.0}, None),
        ("specular_reflection_mirror", {"kind": 3, "fresnel": 2, "color": [0.9, 0.8, 0.7]}, None),
        ("specular_reflection_dielectric",
         {"kind": 3, "fresnel": 0, "color": [1.0, 1.0, 1.0], "eta_i": 1.5, "eta_t": 1.0}, None),
    ]
    for name, d, _ in specs:
        full = {"kind": d["kind"], "fresnel": d["fresnel"], "color": d.get("color", [0, 0, 0]),
                "t": d.get("t", [0, 0, 0]), "eta_i": d.get("eta_i", 1.0), "eta_t": d.get("eta_t", 1.0),
                "eta": d.get("eta", [0, 0, 0]), "k": d.get("k", [0, 0, 0]), "alpha_x": d.get("alpha_x", 0.0),
                "alpha_y": d.get("alpha_y", 0.0), "eta_a": d.get("eta_a", 1.0), "eta_b": d.get("eta_b", 1.0)}
        color = V(*full["color"])
        if full["fresnel"] == 0:
            fr = {"kind": "dielectric", "eta_i": mpf(full["eta_i"]), "eta_t": mpf(full["eta_t"])}
        elif full["fresnel"] == 1:
            fr = {"kind": "conductor", "eta": V(*full["eta"]), "k": V(*full["k"])}
        else:
            fr = {"kind": "noop"}
        ax, ay = mpf(full["alpha_x"]), mpf(full["alpha_y"])
        if full["kind"] == 0:
            fn = lambert(color)
        elif full["kind"] == 1:
            fn = microfacet_reflection(color, fr, ax, ay)
        elif full["kind"] == 2:
            fn = fresnel_specular(color, V(*full["t"]), mpf(full["eta_a"]), mpf(full["eta_b"]))
        elif full["kind"] == 3:
            fn = specular_reflection(color, fr)
        else:
            fn = microfacet_transmission(color, ax, ay, mpf(full["eta_a"]), mpf(full["eta_b"]))
        f, pdf, sample = fn
        evals, samples = [], []
        # f / pdf: 28 direction pairs -- same and opposite hemispheres, both signs of wo.z
        for i in range(28):
            ho = 1 if i % 4 < 2 else -1
            same = (i % 2 == 0) if full["kind"] != 4 else (i % 7 == 0)  # transmission lives in opposite hemispheres
            hi = ho if same else -ho
            wo, wi = direction(rnd, ho), direction(rnd, hi)
            mo, mi = V(*wo), V(*wi)
            evals.append({"wo": wo, "wi": wi, "f": fl3(f(mo, mi)), "pdf": fl(pdf(mo, mi))})
        # sample_f: 24 draws
        for i in range(24):
            wo = direction(rnd, 1 if i % 3 else -1)
            u0, u1 = rnd.uniform(0.02, 0.98), rnd.uniform(0.02, 0.98)
            if abs(u1 - 0.5) < 0.01:
                u1 += 0.03
            key = [rnd.randrange(1 << 20), rnd.randrange(1 << 20), rnd.randrange(1 << 10)]
            r1, r2 = rng_draws(key[0], key[1], key[2], 2)
            sf, swi, sp = sample(V(*wo), mpf(u0), mpf(u1), r1, r2)
            samples.append({"wo": wo, "u": [u0, u1], "rng_key": key, "r": [fl(r1), fl(r2)], "f": fl3(sf),
                            "wi": fl3(swi), "pdf": fl(sp)})
        lobes.append({"name": name, "lobe": full, "eval": evals, "sample": samples})
    # scalar tables: Fresnel, TR pieces, roughness remap
    scal = {"fr_dielectric": [], "fr_conductor": [], "tr": [], "roughness_to_alpha": []}
    for c in [-1.0, -0.7, -0.2, -0.01, 0.0, 0.01, 0.1, 0.3, 0.5, 0.75, 0.9, 1.0]:
        for ei, et in [(1.0, 1.5), (1.5, 1.0), (1.0, 1.3), (1.33, 1.0)]:
            scal["fr_dielectric"].append({"cos": c, "eta_i": ei, "eta_t": et, "value": fl(fr_dielectric(mpf(c), mpf(ei), mpf(et)))})
    for c in [0.0, 0.05, 0.2, 0.4, 0.6, 0.8, 0.95, 1.0]:
        for eta, k in [([0.05, 0.5, 0.75], [0.0, 0.0, 0.0]), ([0.2, 0.92, 1.1], [3.9, 2.45, 2.14]),
                       ([0.01, 0.0, 0.0], [1.0, 1.0, 1.0]), ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])]:
            scal["fr_conductor"].append({"cos": c, "eta": eta, "k": k, "value": fl3(fr_conductor(mpf(c), V(*eta), V(*k)))})
    for i in range(24):
        ax, ay = rnd.choice([(0.001, 0.001), (0.05, 0.05), (0.3, 0.3), (0.35, 0.08), (0.02, 0.6), (1.0, 1.0)])
        wo, wh = direction(rnd, 1 if i % 2 else -1), direction(rnd, 1)
        mo, mh = V(*wo), V(*wh)
        u0, u1 = rnd.uniform(0.02, 0.98), rnd.uniform(0.02, 0.46) + (0.5 if i % 2 else 0.0)
        scal["tr"].append({"alpha": [ax, ay], "wo": wo, "wh": wh, "u": [u0, u1],
                           "d": fl(tr_d(mpf(ax), mpf(ay), mh)), "lambda": fl(tr_lambda(mpf(ax), mpf(ay), mo)),
                           "g": fl(tr_g(mpf(ax), mpf(ay), mo, mh)), "pdf": fl(tr_pdf(mpf(ax), mpf(ay), mo, mh)),
                           "sample_wh": fl3(tr_sample_wh(mpf(ax), mpf(ay), mo, mpf(u0), mpf(u1)))})
    for r in [0.0, 1e-6, 1e-5, 0.001, 0.005, 0.01, 0.0111111, 0.05, 0.1, 0.3, 0.5, 0.9, 1.0]:
        scal["roughness_to_alpha"].append({"roughness": r, "alpha": fl(tr_roughness_to_alpha(r))})
    out = {"generator": "tools/make_lobe_tables.py (mpmath %d digits, formulas transcribed from the reference's "
                        "src/bxdf.rs, src/microfacet.rs, src/util.rs; independent of oracle/ and of the device code)" % mp.dps,
           "pi": "3.14159265358979", "lobes": lobes, "scalars": scal}
    path = os.path.join(ROOT, "tests", "golden", "lobe_tables.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path, sum(len(l["eval"]) + len(l["sample"]) for l in lobes), "lobe points")


if __name__ == "__main__":
    main()

/*
 * rt_detmath.h -- deterministic elementary functions of the numerical contract.
 *
 * The reference evaluates f64::sin/cos/atan2/acos/ln/powf through the
 * platform libm (rand_cosine_dir util.rs:127-148, concentric_sample_disk
 * util.rs:79-94, make_sphere_record intersects.rs:222-238, Checkered
 * material.rs:558-559, roughness_to_alpha microfacet.rs:442-446, the
 * cos_theta>0.9999 branch microfacet.rs:472-476, gamma util.rs:466-470), so
 * its low-order bits already depend on the host.  Path tracing is chaotic in
 * those bits (a one-ulp change of a bounce direction decorrelates the path a
 * few smooth-normal bounces later), so the ABI pins ONE implementation that
 * the HIP kernels and the CPU oracle both evaluate: the classic Sun fdlibm
 * algorithms (argument reduction + minimax polynomials), written here with
 * nothing but IEEE +,-,*,/,sqrt and integer bit moves, so that gcc on the host
 * and hipcc on gfx950 produce bit-identical results when FP contraction is
 * off (-ffp-contract=off is mandatory for every translation unit including
 * this header).  tests/test_host_cpu.py::test_detmath_close_to_libm bounds the distance to libm (<= 2 ulp;
 * <= 1 ulp observed for sin/cos/log/atan2/acos on the domains the path uses).
 */
#ifndef RT_DETMATH_H
#define RT_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ inline
#else
#define RT_HD static inline
#endif

RT_HD uint64_t dm_bits(double x) {
    uint64_t u;
    __builtin_memcpy(&u, &x, 8);
    return u;
}
RT_HD double dm_from_bits(uint64_t u) {
    double x;
    __builtin_memcpy(&x, &u, 8);
    return x;
}
RT_HD uint32_t dm_hi(double x) { return (uint32_t)(dm_bits(x) >> 32); }
RT_HD uint32_t dm_lo(double x) { return (uint32_t)(dm_bits(x) & 0xffffffffu); }
RT_HD double dm_with_hi(double x, uint32_t hi) {
    return dm_from_bits(((uint64_t)hi << 32) | (dm_bits(x) & 0xffffffffull));
}
RT_HD double dm_clear_lo(double x) { return dm_from_bits(dm_bits(x) & 0xffffffff00000000ull); }
RT_HD double dm_abs(double x) { return dm_from_bits(dm_bits(x) & 0x7fffffffffffffffull); }
/* IEEE sqrt: correctly rounded on both targets. */
RT_HD double dm_sqrt(double x) { return __builtin_sqrt(x); }

/* ---- sin / cos kernels on [-pi/4, pi/4] with a tail y (|y| << |x|) -------- */
RT_HD double dm_ksin(double x, double y, int iy) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    if (iy == 0) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

RT_HD double dm_kcos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    uint32_t ix = dm_hi(x) & 0x7fffffffu;
    double z = x * x;
    double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    if (ix < 0x3FD33333u) return 1.0 - (0.5 * z - (z * r - x * y));
    double qx;
    if (ix > 0x3fe90000u)
        qx = 0.28125;
    else
        qx = dm_from_bits((uint64_t)(ix - 0x00200000u) << 32);
    double hz = 0.5 * z - qx;
    double a = 1.0 - qx;
    return a - (hz - (z * r - x * y));
}

/* Reduce x to y0+y1 in [-pi/4,pi/4]; returns quadrant n mod 4.
 * Two Cody-Waite steps (66+ bits of pi/2 exact, 118 with the tail): exact
 * products for |x| < 2^20*pi/2, which covers every call site on the path
 * (largest argument: Checkered frequency 1e4 * 2*pi).                        */
RT_HD int dm_rem_pio2(double x, double* y0, double* y1) {
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                 pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    double t = dm_abs(x);
    double fn = __builtin_floor(t * invpio2 + 0.5);
    double r = t - fn * pio2_1;
    r = r - fn * pio2_2;
    double w = fn * pio2_2t;
    double a = r - w;
    double b = (r - a) - w;
    /* fn < 2^52 always fits int64; quadrant from the low bits */
    int n = (int)(((int64_t)fn) & 3);
    if (dm_bits(x) >> 63) {
        *y0 = -a;
        *y1 = -b;
        return (4 - n) & 3;
    }
    *y0 = a;
    *y1 = b;
    return n;
}

RT_HD double dm_sin(double x) {
    uint32_t ix = dm_hi(x) & 0x7fffffffu;
    if (ix <= 0x3fe921fbu) return dm_ksin(x, 0.0, 0);
    if (ix >= 0x7ff00000u) return x - x;
    double y0, y1;
    int n = dm_rem_pio2(x, &y0, &y1);
    switch (n) {
        case 0: return dm_ksin(y0, y1, 1);
        case 1: return dm_kcos(y0, y1);
        case 2: return -dm_ksin(y0, y1, 1);
        default: return -dm_kcos(y0, y1);
    }
}

RT_HD double dm_cos(double x) {
    uint32_t ix = dm_hi(x) & 0x7fffffffu;
    if (ix <= 0x3fe921fbu) return dm_kcos(x, 0.0);
    if (ix >= 0x7ff00000u) return x - x;
    double y0, y1;
    int n = dm_rem_pio2(x, &y0, &y1);
    switch (n) {
        case 0: return dm_kcos(y0, y1);
        case 1: return -dm_ksin(y0, y1, 1);
        case 2: return -dm_kcos(y0, y1);
        default: return dm_ksin(y0, y1, 1);
    }
}

/* sin and cos of one argument with one argument reduction.  Bit for bit the pair (dm_sin(x), dm_cos(x)): the same
 * kernels on the same reduced argument, only the duplicated reduction is gone.                                */
RT_HD void dm_sincos(double x, double* s, double* c) {
    uint32_t ix = dm_hi(x) & 0x7fffffffu;
    if (ix <= 0x3fe921fbu) {
        *s = dm_ksin(x, 0.0, 0);
        *c = dm_kcos(x, 0.0);
        return;
    }
    if (ix >= 0x7ff00000u) {
        *s = *c = x - x;
        return;
    }
    double y0, y1;
    int n = dm_rem_pio2(x, &y0, &y1);
    double ks = dm_ksin(y0, y1, 1), kc = dm_kcos(y0, y1);
    switch (n) {
        case 0: *s = ks; *c = kc; break;
        case 1: *s = kc; *c = -ks; break;
        case 2: *s = -ks; *c = -kc; break;
        default: *s = -kc; *c = ks; break;
    }
}

/* ---- atan / atan2 --------------------------------------------------------- */
RT_HD double dm_atan(double x) {
    const double atanhi0 = 4.63647609000806093515e-01, atanhi1 = 7.85398163397448278999e-01,
                 atanhi2 = 9.82793723247329054082e-01, atanhi3 = 1.57079632679489655800e+00;
    const double atanlo0 = 2.26987774529616870924e-17, atanlo1 = 3.06161699786838301793e-17,
                 atanlo2 = 1.39033110312309984516e-17, atanlo3 = 6.12323399573676603587e-17;
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
                 aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
                 aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
                 aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
                 aT10 = 1.62858201153657823623e-02;
    uint32_t hx = dm_hi(x);
    uint32_t ix = hx & 0x7fffffffu;
    int neg = (int)(hx >> 31);
    int id;
    double hi = 0.0, lo = 0.0;
    if (ix >= 0x44100000u) { /* |x| >= 2^66 */
        if (ix > 0x7ff00000u || (ix == 0x7ff00000u && dm_lo(x) != 0)) return x + x;
        return neg ? -(atanhi3 + atanlo3) : (atanhi3 + atanlo3);
    }
    if (ix < 0x3fdc0000u) { /* |x| < 0.4375 */
        if (ix < 0x3e200000u) return x;
        id = -1;
    } else {
        x = dm_abs(x);
        if (ix < 0x3ff30000u) {
            if (ix < 0x3fe60000u) {
                id = 0; hi = atanhi0; lo = atanlo0;
                x = (2.0 * x - 1.0) / (2.0 + x);
            } else {
                id = 1; hi = atanhi1; lo = atanlo1;
                x = (x - 1.0) / (x + 1.0);
            }
        } else {
            if (ix < 0x40038000u) {
                id = 2; hi = atanhi2; lo = atanlo2;
                x = (x - 1.5) / (1.0 + 1.5 * x);
            } else {
                id = 3; hi = atanhi3; lo = atanlo3;
                x = -1.0 / x;
            }
        }
    }
    double z = x * x;
    double w = z * z;
    double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    z = hi - ((x * (s1 + s2) - lo) - x);
    return neg ? -z : z;
}

RT_HD double dm_atan2(double y, double x) {
    const double pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16,
                 pi_o_2 = 1.5707963267948965580E+00, pi_o_4 = 7.8539816339744827900E-01;
    if (x != x || y != y) return x + y;
    uint32_t hx = dm_hi(x), lx = dm_lo(x), hy = dm_hi(y), ly = dm_lo(y);
    uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    if (hx == 0x3ff00000u && lx == 0) return dm_atan(y);
    int m = (int)((hy >> 31) & 1) | (int)((hx >> 30) & 2);
    if ((iy | ly) == 0) {
        switch (m) {
            case 0:
            case 1: return y;
            case 2: return pi;
            default: return -pi;
        }
    }
    if ((ix | lx) == 0) return (hy >> 31) ? -pi_o_2 : pi_o_2;
    if (ix == 0x7ff00000u) {
        if (iy == 0x7ff00000u) {
            switch (m) {
                case 0: return pi_o_4;
                case 1: return -pi_o_4;
                case 2: return 3.0 * pi_o_4;
                default: return -3.0 * pi_o_4;
            }
        } else {
            switch (m) {
                case 0: return 0.0;
                case 1: return -0.0;
                case 2: return pi;
                default: return -pi;
            }
        }
    }
    if (iy == 0x7ff00000u) return (hy >> 31) ? -pi_o_2 : pi_o_2;
    int k = ((int)iy - (int)ix) >> 20;
    double z;
    if (k > 60)
        z = pi_o_2 + 0.5 * pi_lo;
    else if ((hx >> 31) && k < -60)
        z = 0.0;
    else
        z = dm_atan(dm_abs(y / x));
    switch (m) {
        case 0: return z;
        case 1: return -z;
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

/* ---- acos ------------------------------------------------------------------ */
RT_HD double dm_acos(double x) {
    const double pi = 3.14159265358979311600e+00, pio2_hi = 1.57079632679489655800e+00,
                 pio2_lo = 6.12323399573676603587e-17;
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                 pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    uint32_t hx = dm_hi(x);
    uint32_t ix = hx & 0x7fffffffu;
    if (ix >= 0x3ff00000u) {
        if (((ix - 0x3ff00000u) | dm_lo(x)) == 0) {
            if (!(hx >> 31)) return 0.0;
            return pi + 2.0 * pio2_lo;
        }
        return (x - x) / (x - x);
    }
    if (ix < 0x3fe00000u) {
        if (ix <= 0x3c600000u) return pio2_hi + pio2_lo;
        double z = x * x;
        double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        double r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    } else if (hx >> 31) {
        double z = (1.0 + x) * 0.5;
        double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        double s = dm_sqrt(z);
        double r = p / q;
        double w = r * s - pio2_lo;
        return pi - 2.0 * (s + w);
    } else {
        double z = (1.0 - x) * 0.5;
        double s = dm_sqrt(z);
        double df = dm_clear_lo(s);
        double c = (z - df * df) / (s + df);
        double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        double r = p / q;
        double w = r * s + c;
        return 2.0 * (df + w);
    }
}

/* ---- log ------------------------------------------------------------------- */
RT_HD double dm_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 two54 = 1.80143985094819840000e+16, Lg1 = 6.666666666666735130e-01,
                 Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01,
                 Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
    int32_t hx = (int32_t)dm_hi(x);
    uint32_t lx = dm_lo(x);
    int k = 0;
    if (hx < 0x00100000) {
        if (((hx & 0x7fffffff) | lx) == 0) return -two54 / 0.0;
        if (hx < 0) return (x - x) / 0.0;
        k -= 54;
        x *= two54;
        hx = (int32_t)dm_hi(x);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int i = (hx + 0x95f64) & 0x100000;
    x = dm_with_hi(x, (uint32_t)(hx | (i ^ 0x3ff00000)));
    k += (i >> 20);
    double f = x - 1.0;
    double dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {
        if (f == 0.0) {
            if (k == 0) return 0.0;
            return dk * ln2_hi + dk * ln2_lo;
        }
        double R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    double s = f / (2.0 + f);
    double z = s * s;
    i = hx - 0x6147a;
    double w = z * z;
    int j = 0x6b851 - hx;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    double R = t2 + t1;
    if (i > 0) {
        double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

/* ---- exp / pow (tone-map row only: util.rs:466-470) ------------------------ */
RT_HD double dm_exp(double x) {
    const double o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02,
                 ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                 P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > o_threshold) return 1e308 * 1e308;
    if (x < u_threshold) return 0.0;
    double hi, lo = 0.0;
    int k = 0;
    double ax = dm_abs(x);
    if (ax > 0.34657359027997264) { /* 0.5 ln2 */
        double kf = __builtin_floor(invln2 * x + 0.5);
        k = (int)kf;
        hi = x - kf * ln2HI;
        lo = kf * ln2LO;
        x = hi - lo;
    } else if (ax < 3.7252902984619140625e-09) {
        return 1.0 + x;
    } else {
        hi = x;
    }
    double t = x * x;
    double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double y;
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    /* scale by 2^k in two steps to stay in range */
    if (k >= -1021 && k <= 1023) return y * dm_from_bits((uint64_t)(k + 1023) << 52);
    if (k > 1023) return y * dm_from_bits((uint64_t)(2046) << 52) * dm_from_bits((uint64_t)(k - 1023 + 1023) << 52);
    return y * dm_from_bits((uint64_t)(k + 1000 + 1023) << 52) * dm_from_bits((uint64_t)(1023 - 1000) << 52);
}

/* x^y for x >= 0 as exp(y*log x); exact cases x==0, x==1, y==0 first.        */
RT_HD double dm_pow(double x, double y) {
    if (y == 0.0 || x == 1.0) return 1.0;
    if (x != x || y != y) return x + y;
    if (x == 0.0) return y > 0.0 ? 0.0 : 1.0 / 0.0;
    if (x < 0.0) return (x - x) / (x - x);
    return dm_exp(y * dm_log(x));
}

#endif /* RT_DETMATH_H */

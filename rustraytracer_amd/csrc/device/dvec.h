// dvec.h -- f64 3-vectors, Rust-style min/max and the counter RNG for gfx950 device
// code (also compiled for the host by hipcc for the BVH builder's helpers).
// Every expression keeps the reference's evaluation order; the translation unit is
// built with -ffp-contract=off so that no mul+add pair is fused (numerical
// contract, include/rt_abi.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/rt_abi.h"
#include "../../../include/rt_detmath.h"

#define RTD __device__ __forceinline__
#define RTDN __device__ __noinline__

// Values that stay binary64 in the f32 fast mode too: everything that lives in HBM (scene records of the ABI, path
// state, film staging, the film).  tools/make_f32_sources.py turns every other `double` of the device sources into
// `float` when it generates the fast-mode kernels (namespace rtd32); these two typedefs are left alone.
typedef double f64_t;     // RT_KEEP_F64
typedef double2 f64x2_t;  // RT_KEEP_F64

namespace rtd {

#ifdef RT_F32
// ---- fast mode: hardware-rate single-precision elementary functions (not part of the numerical contract)
#undef dm_sin
#undef dm_cos
#undef dm_atan2
#undef dm_acos
#undef dm_log
RTD double dm_sin(double x) { return ::sinf(x); }
RTD double dm_cos(double x) { return ::cosf(x); }
RTD double dm_atan2(double y, double x) { return ::atan2f(y, x); }
RTD double dm_acos(double x) { return ::acosf(x); }
RTD double dm_log(double x) { return ::logf(x); }
RTD double dm_sqrt(double x) { return __builtin_sqrt(x); }
struct SinCos {
    double s, c;
};
RTD SinCos sincos2(double x) { return SinCos{::sinf(x), ::cosf(x)}; }
#else
// The elementary functions are called from dozens of sites of the shading code; inlined everywhere they blew
// the shading kernels up to 170-290 KB of code against a 64 KB instruction cache.  One out-of-line copy each:
#ifndef RT_INLINE_MATH
RTDN double ni_sin(double x) { return dm_sin(x); }
RTDN double ni_cos(double x) { return dm_cos(x); }
RTDN double ni_atan2(double y, double x) { return dm_atan2(y, x); }
RTDN double ni_acos(double x) { return dm_acos(x); }
RTDN double ni_log(double x) { return dm_log(x); }
struct SinCos {
    double s, c;
};
RTDN SinCos sincos2(double x) {  // == {dm_sin(x), dm_cos(x)} bit for bit, with one argument reduction (rt_detmath.h)
    SinCos r;
    dm_sincos(x, &r.s, &r.c);
    return r;
}
#define dm_sin ni_sin
#define dm_cos ni_cos
#define dm_atan2 ni_atan2
#define dm_acos ni_acos
#define dm_log ni_log
#else
struct SinCos {
    double s, c;
};
RTD SinCos sincos2(double x) {
    SinCos r;
    dm_sincos(x, &r.s, &r.c);
    return r;
}
#endif
#endif

// consts.rs:30-42
constexpr double kPi = 3.14159265358979;
constexpr double kSmall = 0.001;
constexpr double kInf = 1e308;
constexpr double kInvPi = 1.0 / kPi;

// f64::max / f64::min: the non-NaN operand wins
RTD double rmax(double a, double b) { return (a != a) ? b : ((b != b) ? a : (a < b ? b : a)); }
RTD double rmin(double a, double b) { return (a != a) ? b : ((b != b) ? a : (b < a ? b : a)); }
RTD double clampd(double x, double lo, double hi) { return rmin(rmax(x, lo), hi); }  // util.rs:28-30
RTD double absd(double x) { return __builtin_fabs(x); }

struct D3 {
    double x, y, z;
};
RTD D3 d3(double x, double y, double z) { return D3{x, y, z}; }
RTD D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
RTD D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
RTD D3 operator-(D3 a) { return {-a.x, -a.y, -a.z}; }
RTD D3 operator*(D3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
RTD D3 operator*(double s, D3 a) { return {s * a.x, s * a.y, s * a.z}; }
RTD D3 operator/(D3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
RTD D3 cmul(D3 a, D3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
RTD D3 cdiv(D3 a, D3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
RTD double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RTD D3 cross(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
RTD double norm2(D3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
RTD double norm(D3 a) { return dm_sqrt(norm2(a)); }
// (inlined at its ~40 call sites in the shading kernels: the out-of-line call of rounds 1-2 -- chosen for code size -- cost
// k_shade 2-4 %, profiles/r03_exp_inline_normalize.txt; the elementary-function wrappers above stay out of line: no gain)
RTD D3 normalize(D3 a) { return a / norm(a); }
RTD bool is_black(D3 a) { return a.x == 0.0 && a.y == 0.0 && a.z == 0.0; }
RTD D3 black() { return {0.0, 0.0, 0.0}; }
RTD D3 white() { return {1.0, 1.0, 1.0}; }
RTD double comp(D3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// RNG block of include/rt_abi.h
RTD uint64_t rng_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
RTD uint64_t rng_init(uint64_t seed, uint64_t pixel, uint64_t sample) {
    return rng_mix(rng_mix(seed * RT_RNG_G + pixel) + sample * RT_RNG_H + RT_RNG_J);
}
RTD double rng_next(uint64_t& s) {
    s += RT_RNG_G;
#ifdef RT_F32
    return (double)(uint32_t)(rng_mix(s) >> 40) * (1.0 / 16777216.0);  // the top 24 bits of the same draw
#else
    return (double)(rng_mix(s) >> 11) * (1.0 / 9007199254740992.0);
#endif
}

}  // namespace rtd

// ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
// product path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may load it.  PARITY UNPINNED: the reference holds no golden vectors, tests
// or fixtures and cannot be built here (Rust, no toolchain), so
// this restatement is pinned only by hand-derived known-answer tests.
//
// vecmath.h -- the nalgebra 0.21.1 Vector3<f64>/Point3<f64> operations the hot
// path uses (SURVEY.md 8c "third-party arithmetic"), restated as plain f64.
#pragma once
#include <cmath>
#include <cstdint>

#include "../include/rt_abi.h"
#include "../include/rt_detmath.h"

namespace orc {

// Rust f64::max / f64::min: "if one argument is NaN the other is returned".
inline double rmax(double a, double b) { return (a != a) ? b : ((b != b) ? a : (a < b ? b : a)); }
inline double rmin(double a, double b) { return (a != a) ? b : ((b != b) ? a : (b < a ? b : a)); }
// util::clamp (util.rs:28-30): x.max(min).min(max)
inline double clampd(double x, double lo, double hi) { return rmin(rmax(x, lo), hi); }

struct V3 {
    double x, y, z;
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 cmul(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 cdiv(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double norm2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline double norm(V3 a) { return dm_sqrt(norm2(a)); }
// Unit::new_normalize: v / ||v|| (NaN for the zero vector, relied on at hittable.rs:121)
inline V3 normalize(V3 a) { return a / norm(a); }
inline bool eq(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool is_black(V3 a) { return a.x == 0.0 && a.y == 0.0 && a.z == 0.0; }
inline V3 black() { return {0.0, 0.0, 0.0}; }
inline V3 white() { return {1.0, 1.0, 1.0}; }

// Counter RNG of include/rt_abi.h ("RNG" block).
inline uint64_t rng_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct Rng {
    uint64_t s;
    Rng(uint64_t seed, uint64_t pixel, uint64_t sample) {
        s = rng_mix(rng_mix(seed * RT_RNG_G + pixel) + sample * RT_RNG_H + RT_RNG_J);
    }
    double next() {
        s += RT_RNG_G;
        return (double)(rng_mix(s) >> 11) * (1.0 / 9007199254740992.0);
    }
};

}  // namespace orc

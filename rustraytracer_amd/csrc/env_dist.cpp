// env_dist.cpp -- see env_dist.h.  Runs once per rt_scene_commit; plain sequential f64 prefix sums in the
// reference's order, so the tables are bit-identical to the ones its constructor builds from the same texels.
#include "env_dist.h"

#include <cmath>

#include "../../include/rt_detmath.h"

namespace rtd {

namespace {

uint32_t sat_u32(double x) {  // Rust `f64 as u32`
    if (!(x > 0.0)) return 0u;
    if (x >= 4294967295.0) return 4294967295u;
    return (uint32_t)x;
}

// src/material.rs:570-587 (Texture::Hdr arm of get_value)
void hdr_value(const rt_texture& t, double u, double v, double* rgb) {
    uint32_t x = sat_u32(std::round((1.0 - u) * (double)t.width));
    uint32_t y = sat_u32(std::round(v * (double)t.height));
    x = x % t.width;
    y = y % t.height;
    const uint8_t* q = t.rgbe + 4 * ((size_t)y * t.width + x);
    const double sc = std::ldexp(1.0, (int)q[3] - 128);
    for (int c = 0; c < 3; c++) rgb[c] = ((double)q[c] + 0.5) * sc / 256.0;
}

// src/distribution.rs:27-55: cdf of n values; returns func_int
double make_cdf(const double* f, size_t n, double* cdf) {
    cdf[0] = 0.0;
    for (size_t i = 1; i < n + 1; i++) cdf[i] = cdf[i - 1] + f[i - 1] / (double)n;
    const double func_int = cdf[n];
    if (func_int == 0.0) {
        for (size_t i = 1; i < n + 1; i++) cdf[i] = (double)i / (double)n;
    } else {
        for (size_t i = 1; i < n + 1; i++) cdf[i] = cdf[i] / func_int;
    }
    return func_int;
}

}  // namespace

void build_env_dist(const rt_texture& t, EnvDist& d) {
    const size_t width = (size_t)t.width * 2, height = (size_t)t.height * 2;  // light.rs:613-614
    d.nu = (uint32_t)width;
    d.nv = (uint32_t)height;
    d.img.assign(width * height, 0.0);
    for (size_t v = 0; v < height; v++) {
        const double vp = ((double)v + 0.5) / (double)height;
        const double sin_theta = dm_sin(RT_PI * ((double)v + 0.5) / (double)height);
        for (size_t u = 0; u < width; u++) {
            const double up = (double)u / (double)width;
            double c[3];
            hdr_value(t, up, vp, c);
            const double lum = 0.2126 * c[0] + 0.7152 * c[1] + 0.0722 * c[2];  // util.rs:169-171
            d.img[u + v * width] = lum * sin_theta;
        }
    }
    // distribution.rs:115-128; row v = f[v .. v + nu) -- the reference's own slicing, kept
    d.cond_cdf.assign(height * (width + 1), 0.0);
    d.marg_func.assign(height, 0.0);
    for (size_t v = 0; v < height; v++) d.marg_func[v] = make_cdf(d.img.data() + v, width, &d.cond_cdf[v * (width + 1)]);
    d.marg_cdf.assign(height + 1, 0.0);
    d.marg_int = make_cdf(d.marg_func.data(), height, d.marg_cdf.data());
}

}  // namespace rtd

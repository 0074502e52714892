// env_dist.h -- host-side construction of the environment light's sampling tables (next-row f4).
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/rt_abi.h"

namespace rtd {

// Flattened Distribution2D of Light::make_infinite_light (src/light.rs:608-638, src/distribution.rs:27-55, 115-128).
struct EnvDist {
    std::vector<double> img;        // nu * nv: luminance * sin(theta) of the 2x-upsampled map
    std::vector<double> cond_cdf;   // nv rows of nu + 1 (row v is over img[v .. v + nu), as the reference slices it)
    std::vector<double> marg_func;  // nv: func_int of every row
    std::vector<double> marg_cdf;   // nv + 1
    double marg_int = 0.0;
    uint32_t nu = 0, nv = 0;
};

// `t` is an RT_TEX_HDR texture whose rgbe pointer is readable on the host.
void build_env_dist(const rt_texture& t, EnvDist& out);

}  // namespace rtd

// bvh_build.h -- host BVH builder interface (see bvh_build.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "device/scene_dev.h"

namespace rtd {

struct BvhOut {
    std::vector<DevNode> nodes;   // node 0 = root (always internal)
    std::vector<uint32_t> order;  // leaf order -> primitive index
    uint32_t depth = 0;
};

void build_bvh(const rt_primitive* prims, size_t n, BvhOut& out);

}  // namespace rtd

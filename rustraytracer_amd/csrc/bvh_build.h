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

// experiment RT_BVH8: the same binned-SAH binary tree collapsed to 8 children per node, boxes quantised on the
// node's grid (scene_dev.h: DevNode8).  One primitive per leaf.  stack_need = the deepest the traversal's stack can get
// (7 pending siblings per level).
struct Bvh8Out {
    std::vector<DevNode8> nodes;
    std::vector<uint32_t> order;  // leaf order -> primitive index (a node's leaf children are consecutive)
    uint32_t depth = 0;
    uint32_t stack_need = 0;
    double quant_area_ratio = 0.0;  // diagnostic: sum of decoded child areas / sum of exact child areas
};
void build_bvh8(const rt_primitive* prims, size_t n, Bvh8Out& out);

}  // namespace rtd

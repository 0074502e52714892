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

// Binned-SAH BINARY tree over n boxes (6 doubles each: min xyz, max xyz), one box per leaf -- the top of the device
// builder's tree (bvh_gpu.hip: clusters of the Morton-order tree become the leaves).  Internal nodes are numbered
// 0 .. n-2 with the root at 0; a child reference >= 0 is an internal node, < 0 is ~(box index).  n >= 2.
void build_sah_binary(const double* boxes, size_t n, std::vector<int32_t>& left, std::vector<int32_t>& right);

}  // namespace rtd

// bvh_cache.h -- the host BVH shared between the processes of one node (see bvh_cache.cpp).
#pragma once
#include <cstddef>

#include "bvh_build.h"

namespace rtd {
// build_bvh() once per node when RT_BVH_CACHE names a directory: returns 1 when the tree came from the cache file, 0
// when this process built it (and, if it could, published it).
int build_bvh_shared(const rt_primitive* prims, size_t np, BvhOut& bvh);
// (exposed for the tests) everything a cached tree must satisfy before the traversal may walk it: child references and
// leaf ranges inside the arrays, every node reached exactly once from the root (no cycle, no sharing), true depth.
bool bvh_validate(const BvhOut& bvh, size_t np, uint32_t* depth_out);
}  // namespace rtd

// bvh_gpu.h -- device BVH builder interface (see bvh_gpu.hip).  SURVEY.md section 8, row f3.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "device/scene_dev.h"

namespace rtd {

struct DeviceBvh {
    DevNode* nodes = nullptr;      // n_nodes entries, node 0 = root (always internal), breadth-first order
    uint32_t* leaf_prim = nullptr; // n entries: leaf order -> prim index | kLeafOther
    double* leaf_tri = nullptr;    // n * 9 doubles: the leaf slots (geom.h: leaf_step)
    double* leaf_nrm = nullptr;    // n * 9 doubles: vertex normals per slot (null when no mesh has normals)
    LeafMeta* leaf_meta = nullptr; // n: {mat | flags, light_index} per slot (scene_dev.h)
    uint32_t n_nodes = 0;
    uint32_t depth = 0;            // depth of the 4-wide tree (root = 0)
    uint64_t n_triangles = 0;
    float build_ms = 0.0f;         // device time of the whole build (HIP events)
    uint32_t rotation_passes = 0;  // refit passes that applied tree rotations (0: plain Morton-order tree)
    uint32_t sah_top_clusters = 0; // leaves of the host-built SAH top (0: the top is the Morton-order tree's own)
    uint32_t sah_bottom_clusters = 0; // clusters whose inner topology was rebuilt by binned SAH on the device
    uint32_t ploc_passes = 0;      // > 0: the topology is PLOC's (agglomeration passes it took); 0: the Morton-order tree
};

// Builds the traversal structure of `n` primitives that are already resident in HBM (`d_prims`, with the
// caller-supplied f64 boxes of rt_primitive, and `d_meshes` for the triangle vertices).  The result
// arrays are hipMalloc'ed and owned by the caller.  Returns 0, or a negative rt_status with a message in `err`.
int build_bvh_device(hipStream_t stream, const rt_primitive* d_prims, const DevMesh* d_meshes, uint32_t n,
                     bool any_normals, DeviceBvh* out, char* err, size_t err_len);

}  // namespace rtd

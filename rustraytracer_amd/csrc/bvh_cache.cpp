// bvh_cache.cpp -- host BVH shared between the processes of one node (RT_BVH_CACHE=<directory>, e.g. /dev/shm/...).
// bench.py --gpus N runs one process per GPU and every rank commits the same scene: without this, eight ranks each
// run the 16-thread SAH build of 1.74 M primitives at the same time on the same host cores.  With the variable set,
// the ranks take an exclusive flock() on <dir>/rtbvh_<key>.lock in turn: the first one builds and publishes
// <dir>/rtbvh_<key>.bin (write to a temporary name, then rename), the others read it.  The key hashes everything
// build_bvh() looks at -- the primitive records incl. their f64 boxes, the builder's environment parameters -- and a
// format tag.  Results cannot depend on the file: the tree only culls (geom.h).  But a kernel WALKS it, so a file is
// trusted no further than it is checked: the checksum covers header and payload, child references and leaf
// ranges must lie inside the arrays, the nodes must form a tree (one visit each from the root: no cycle can loop a
// wave, no shared subtree), and the depth the traversal stack is sized against is recomputed, not read.  The directory
// must belong to this user and be writable by nobody else.  Anything that fails is ignored and the tree rebuilt.
#include "bvh_cache.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

namespace rtd {
namespace {

struct BvhCacheHeader {
    char magic[8];  // "RTBVH\0\0\3"
    uint64_t key, n_prims, n_nodes;
    uint32_t depth, node_bytes;
    uint64_t sum;  // cache_sum over this header (with sum = 0), nodes and order: a torn or damaged file is rebuilt
};
const char kMagic[8] = {'R', 'T', 'B', 'V', 'H', 0, 0, 3};

struct Fnv {
    uint64_t h = 0xcbf29ce484222325ull;
    void mix(const void* p, size_t bytes) {
        const unsigned char* c = static_cast<const unsigned char*>(p);
        size_t i = 0;
        for (; i + 8 <= bytes; i += 8) {
            uint64_t w;
            std::memcpy(&w, c + i, 8);
            h = (h ^ w) * 0x100000001b3ull;
            h ^= h >> 31;
        }
        for (; i < bytes; i++) h = (h ^ c[i]) * 0x100000001b3ull;
    }
};
uint64_t cache_sum(BvhCacheHeader hd, const BvhOut& b) {
    hd.sum = 0;
    Fnv f;
    f.mix(&hd, sizeof(hd));
    f.mix(b.nodes.data(), b.nodes.size() * sizeof(DevNode));
    f.mix(b.order.data(), b.order.size() * sizeof(uint32_t));
    return f.h;
}
uint64_t cache_key(const rt_primitive* prims, size_t n) {
    Fnv f;
    const uint64_t tag[2] = {(uint64_t)sizeof(DevNode), (uint64_t)kLeafTargetPrims};
    f.mix(tag, sizeof(tag));
    // what build_bvh() reads from the environment (bvh_build.cpp): another setting is another tree
    for (const char* name : {"RT_BVH_BINS", "RT_BVH_LEAF", "RT_BVH_SAH_DEPTH", "RT_BVH_OUTLIER", "RT_BVH_THREADS"}) {
        const char* v = getenv(name);
        f.mix(name, std::strlen(name) + 1);
        if (v) f.mix(v, std::strlen(v) + 1);
    }
    f.mix(prims, n * sizeof(rt_primitive));
    return f.h ^ (uint64_t)n;
}
bool dir_is_ours(const char* dir) {
    struct stat st;
    if (stat(dir, &st) != 0 || !S_ISDIR(st.st_mode)) return false;
    return st.st_uid == geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}
bool cache_load(const std::string& path, uint64_t key, size_t np, BvhOut& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    BvhCacheHeader hd;
    bool ok = fread(&hd, sizeof(hd), 1, f) == 1 && std::memcmp(hd.magic, kMagic, 8) == 0 && hd.key == key &&
              hd.n_prims == np && hd.node_bytes == sizeof(DevNode) && hd.n_nodes > 0 && hd.n_nodes <= 2 * np + 2;
    if (ok) {
        out.nodes.resize(hd.n_nodes);
        out.order.resize(np);
        ok = fread(out.nodes.data(), sizeof(DevNode), hd.n_nodes, f) == hd.n_nodes &&
             fread(out.order.data(), sizeof(uint32_t), np, f) == np;
        ok = ok && cache_sum(hd, out) == hd.sum;
        uint32_t depth = 0;
        ok = ok && bvh_validate(out, np, &depth);
        out.depth = depth;
    }
    fclose(f);
    return ok;
}
void cache_store(const std::string& path, uint64_t key, size_t np, const BvhOut& bvh) {
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    BvhCacheHeader hd{};
    std::memcpy(hd.magic, kMagic, 8);
    hd.key = key;
    hd.n_prims = np;
    hd.n_nodes = bvh.nodes.size();
    hd.depth = bvh.depth;
    hd.node_bytes = sizeof(DevNode);
    hd.sum = cache_sum(hd, bvh);
    const bool ok = fwrite(&hd, sizeof(hd), 1, f) == 1 &&
                    fwrite(bvh.nodes.data(), sizeof(DevNode), bvh.nodes.size(), f) == bvh.nodes.size() &&
                    fwrite(bvh.order.data(), sizeof(uint32_t), np, f) == np;
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
}

}  // namespace

bool bvh_validate(const BvhOut& bvh, size_t np, uint32_t* depth_out) {
    const size_t nn = bvh.nodes.size();
    if (nn == 0 || bvh.order.size() != np) return false;
    for (size_t i = 0; i < np; i++)
        if (bvh.order[i] >= np) return false;
    // depth-first from the root with a visit budget of one per node: a second visit is a cycle or a shared subtree
    std::vector<uint8_t> seen(nn, 0);
    std::vector<std::pair<uint32_t, uint32_t>> stack;  // {node, depth}
    stack.emplace_back(0u, 0u);
    seen[0] = 1;
    uint32_t depth = 0;
    size_t visited = 0;
    while (!stack.empty()) {
        const auto [node, d] = stack.back();
        stack.pop_back();
        visited++;
        depth = d > depth ? d : depth;
        for (int k = 0; k < 4; k++) {
            const int32_t c = bvh.nodes[node].child[k];
            if (c == kNoChild) continue;
            if (c >= 0) {
                if ((size_t)c >= nn || seen[c]) return false;
                seen[c] = 1;
                stack.emplace_back((uint32_t)c, d + 1);
            } else {
                const uint32_t code = (uint32_t)(-1 - c) & ~kLeafCodeOther;
                if ((uint64_t)(code >> 3) + (code & 7u) >= np) return false;
            }
        }
    }
    if (visited != nn) return false;  // (nodes nobody points at: not what build_bvh() writes)
    if (depth_out) *depth_out = depth;
    return true;
}

int build_bvh_shared(const rt_primitive* prims, size_t np, BvhOut& bvh) {
    const char* dir = getenv("RT_BVH_CACHE");
    if (!dir || !*dir || np < 4096 || !dir_is_ours(dir)) {  // small scenes build in milliseconds
        build_bvh(prims, np, bvh);
        return 0;
    }
    const uint64_t key = cache_key(prims, np);
    char name[64];
    snprintf(name, sizeof(name), "/rtbvh_%016llx", (unsigned long long)key);
    const std::string base = std::string(dir) + name;
    const int lock = open((base + ".lock").c_str(), O_CREAT | O_RDWR | O_NOFOLLOW, 0600);
    if (lock >= 0) (void)flock(lock, LOCK_EX);  // (no lock, e.g. a read-only directory: everybody builds, as before)
    int from_cache = 0;
    if (cache_load(base + ".bin", key, np, bvh)) {
        from_cache = 1;
    } else {
        bvh = BvhOut{};
        build_bvh(prims, np, bvh);
        if (lock >= 0) cache_store(base + ".bin", key, np, bvh);
    }
    if (lock >= 0) {
        (void)flock(lock, LOCK_UN);
        close(lock);
    }
    return from_cache;
}

}  // namespace rtd

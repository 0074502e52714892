// host_san_driver.cpp -- TEST INFRASTRUCTURE: drives the product's HOST code (scene presets, OBJ / HDR readers, the SAH
// builder with its fork-join pool, the shared tree cache) in a sanitizer build, without a GPU (tests/test_host_sanitizers.py).
// Linked against csrc/host/*.cpp, bvh_build.cpp, bvh_cache.cpp, env_dist.cpp compiled with -fsanitize=address,undefined
// or -fsanitize=thread.  The C-ABI entry points of abi.hip that host_api.cpp forwards to are stubbed: nothing here uploads.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_host.h"
#include "../../rustraytracer_amd/csrc/bvh_cache.h"
#include "../../rustraytracer_amd/csrc/env_dist.h"

extern "C" {
int rt_render(rt_context*, rt_scene*, const rt_camera*, const rt_render_cfg*, double*, uint32_t*, rt_stats*) { return RT_ERR_NO_DEVICE; }
int rt_scene_commit_ex(rt_scene*, uint32_t) { return RT_ERR_NO_DEVICE; }
int rt_scene_create(rt_context*, rt_scene**) { return RT_ERR_NO_DEVICE; }
int rt_scene_destroy(rt_scene*) { return RT_ERR_NO_DEVICE; }
int rt_scene_set_lights(rt_scene*, const rt_light*, uint64_t) { return RT_ERR_NO_DEVICE; }
int rt_scene_set_materials(rt_scene*, const rt_material*, uint64_t) { return RT_ERR_NO_DEVICE; }
int rt_scene_set_meshes(rt_scene*, const rt_mesh*, uint64_t) { return RT_ERR_NO_DEVICE; }
int rt_scene_set_primitives(rt_scene*, const rt_primitive*, uint64_t) { return RT_ERR_NO_DEVICE; }
int rt_scene_set_textures(rt_scene*, const rt_texture*, uint64_t) { return RT_ERR_NO_DEVICE; }
int rt_scene_set_transforms(rt_scene*, const rt_xform*, uint64_t) { return RT_ERR_NO_DEVICE; }
}

using namespace rtd;

static bool same_tree(const BvhOut& a, const BvhOut& b) {
    return a.depth == b.depth && a.order == b.order && a.nodes.size() == b.nodes.size() &&
           std::memcmp(a.nodes.data(), b.nodes.data(), a.nodes.size() * sizeof(DevNode)) == 0;
}

// usage: driver preset <name> <faces> <variant> [mesh_path]      build the preset, its tree, validate it
//        driver cache <dir> <name> <faces> [threads]             build_bvh_shared from `threads` threads at once, twice
//        driver load <dir> <name> <faces>                        build_bvh_shared once: prints from_cache
int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const std::string cmd = argv[1];
    const char* preset = cmd == "preset" ? argv[2] : argv[3];
    const uint64_t faces = std::strtoull(cmd == "preset" ? argv[3] : argv[4], nullptr, 10);
    const int variant = cmd == "preset" ? std::atoi(argv[4]) : 0;
    const char* mesh_path = (cmd == "preset" && argc > 5) ? argv[5] : nullptr;
    rrh_scene* sc = nullptr;
    const int rc = rrh_scene_build(preset, 1.0, faces, mesh_path, variant, &sc);
    if (rc != RT_OK) {
        std::printf("status %d error %s\n", rc, rrh_last_error());
        return 0;  // a refused input is a RESULT, not a crash
    }
    const rt_scene_desc* d = rrh_scene_desc(sc);
    std::printf("status 0 prims %llu meshes %llu lights %llu\n", (unsigned long long)d->n_prims, (unsigned long long)d->n_meshes,
                (unsigned long long)d->n_lights);
    if (cmd == "preset") {
        BvhOut bvh;
        build_bvh(d->prims, d->n_prims, bvh);
        uint32_t depth = 0;
        const bool ok = bvh_validate(bvh, d->n_prims, &depth);
        std::printf("tree nodes %zu depth %u validated %d recomputed_depth %u\n", bvh.nodes.size(), bvh.depth, (int)ok, depth);
        for (uint64_t i = 0; i < d->n_textures; i++)
            if (d->textures[i].kind == RT_TEX_HDR) {
                EnvDist ed;
                build_env_dist(d->textures[i], ed);
                std::printf("env %u x %u marg_int %.6g\n", ed.nu, ed.nv, ed.marg_int);
            }
    } else {
        setenv("RT_BVH_CACHE", argv[2], 1);
        const int threads = argc > 5 ? std::atoi(argv[5]) : 1;
        std::vector<BvhOut> trees((size_t)threads);
        std::vector<int> from((size_t)threads, -1);
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++)
            th.emplace_back([&, t] { from[(size_t)t] = build_bvh_shared(d->prims, d->n_prims, trees[(size_t)t]); });
        for (auto& t : th) t.join();
        int cached = 0;
        for (int t = 0; t < threads; t++) cached += from[(size_t)t];
        bool same = true;
        for (int t = 1; t < threads; t++) same = same && same_tree(trees[0], trees[(size_t)t]);
        BvhOut fresh;
        build_bvh(d->prims, d->n_prims, fresh);
        std::printf("from_cache %d of %d identical %d equals_fresh_build %d nodes %zu depth %u\n", cached, threads, (int)same,
                    (int)same_tree(trees[0], fresh), trees[0].nodes.size(), trees[0].depth);
    }
    rrh_scene_destroy(sc);
    return 0;
}

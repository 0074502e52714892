/* gpu_tile.c -- a host written against the C ABI alone, in plain C99: what the reference's Rust side would do through
 * its `extern "C"` block, compiled here by a C compiler because the image has no rustc.
 *
 *   render::tile_multithread(path, camera, sampler, int_type)   (src/render.rs:13)
 *     -> flatten Objects (here: one of the scenes.rs presets through rt_host.h's stand-in for the Rust host)
 *     -> rt_context_create / rt_scene_create / rt_scene_set_* / rt_scene_commit      (rrh_scene_upload does the five)
 *     -> rt_render                                                                   (the whole image, one call)
 *     -> rt_resolve_rgb8 + PNG                                                       (util::draw_picture, util.rs:387-398)
 *
 * build: gcc -std=c99 -O2 -Iinclude examples/gpu_tile.c -o /tmp/gpu_tile -Lrustraytracer_amd -l:librt_amd.so \
 *            -Wl,-rpath,$PWD/rustraytracer_amd -Wl,-rpath,/opt/rocm/lib
 * run:   /tmp/gpu_tile cornell_box 256 256 16 out.png      (needs a HIP device: there is no CPU fallback)
 * It prints one line: rays, kernel time, a 64-bit FNV-1a checksum of the film's bytes (tests compare it with the
 * Python binding's film of the same call).                                                                         */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_abi.h"
#include "rt_host.h"

static unsigned long long fnv1a(const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    unsigned long long h = 0xcbf29ce484222325ull;
    size_t i;
    for (i = 0; i < n; i++) h = (h ^ b[i]) * 0x100000001b3ull;
    return h;
}

int main(int argc, char** argv) {
    const char* preset = argc > 1 ? argv[1] : "cornell_box";
    const uint32_t W = argc > 2 ? (uint32_t)atoi(argv[2]) : 128, H = argc > 3 ? (uint32_t)atoi(argv[3]) : 128;
    const uint32_t spp = argc > 4 ? (uint32_t)atoi(argv[4]) : 16;
    const char* png = argc > 5 ? argv[5] : NULL;
    rrh_scene* host = NULL;
    rt_context* ctx = NULL;
    rt_scene* scene = NULL;
    rt_render_cfg cfg;
    rt_stats st;
    double* rgb;
    uint32_t* n;
    int dev = 0, rc;

    if (rt_abi_version() != RT_ABI_VERSION) {
        fprintf(stderr, "librt_amd.so speaks ABI %d, this program was built for %d\n", rt_abi_version(), RT_ABI_VERSION);
        return 2;
    }
    if (rrh_scene_build(preset, (double)W / (double)H, 20000, NULL, 0, &host) != RT_OK) {
        fprintf(stderr, "scene: %s\n", rrh_last_error());
        return 1;
    }
    if ((rc = rt_context_create(&dev, 1, &ctx)) != RT_OK || (rc = rrh_scene_upload(ctx, rrh_scene_desc(host), &scene)) != RT_OK) {
        fprintf(stderr, "rt error %d: %s\n", rc, rt_last_error());
        return 1;
    }
    memset(&cfg, 0, sizeof(cfg));
    cfg.width = W;
    cfg.height = H;
    cfg.spp = spp;
    cfg.max_depth = 25; /* consts.rs:7 */
    cfg.seed = 0;
    cfg.tile_size = 16; /* consts.rs:10 */
    cfg.tile_world = 1;
    rgb = (double*)calloc((size_t)W * H * 3, sizeof(double));
    n = (uint32_t*)calloc((size_t)W * H, sizeof(uint32_t));
    if (!rgb || !n) return 1;
    if ((rc = rt_render(ctx, scene, rrh_scene_camera(host), &cfg, rgb, n, &st)) != RT_OK) {
        fprintf(stderr, "rt_render: %d: %s\n", rc, rt_last_error());
        return 1;
    }
    printf("%s %ux%u@%u rays %llu kernel_ms %.3f film_fnv %016llx count_fnv %016llx\n", rrh_scene_name(host), W, H, spp,
           (unsigned long long)(st.rays_extension + st.rays_shadow + st.rays_probe), st.kernel_ms,
           fnv1a(rgb, (size_t)W * H * 3 * sizeof(double)), fnv1a(n, (size_t)W * H * sizeof(uint32_t)));
    if (png) {
        uint8_t* rgb8 = (uint8_t*)malloc((size_t)W * H * 3);
        if (!rgb8 || rt_resolve_rgb8(ctx, rgb, n, W, H, rgb8) != RT_OK || rrh_write_png(png, rgb8, W, H) != RT_OK) {
            fprintf(stderr, "png: %s / %s\n", rt_last_error(), rrh_last_error());
            return 1;
        }
        free(rgb8);
    }
    free(rgb);
    free(n);
    rt_scene_destroy(scene);
    rt_context_destroy(ctx);
    rrh_scene_destroy(host);
    return 0;
}

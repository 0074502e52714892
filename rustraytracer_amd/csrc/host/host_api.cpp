// host_api.cpp -- C entry points of the host-side mirror (rr_host.hpp), so that the
// Python tests and bench.py can build the reference's scene presets and drive the
// C ABI the way the reference's render driver would:
//   rrh_gpu_tile()  <->  render::tile_multithread(path, camera, sampler, int_type)
//                        src/render.rs:13-159, with IntType::Path{max_depth,
//                        invisible_light:false} (src/main.rs:262-265).
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <algorithm>
#include <vector>

#include "../../../include/rt_host.h"
#include "rr_host.hpp"

static thread_local std::string g_host_err;

struct rrh_scene {
    rr::FlatScene flat;
};

extern "C" {

const char* rrh_last_error(void) { return g_host_err.c_str(); }

int rrh_scene_build(const char* preset, double aspect_ratio, uint64_t mesh_faces, const char* mesh_path, int variant,
                    rrh_scene** out) {
    if (!preset || !out) {
        g_host_err = "rrh_scene_build: null argument";
        return RT_ERR_INVALID_ARG;
    }
    rrh_scene* s = new (std::nothrow) rrh_scene();
    if (!s) return RT_ERR_OOM;
    rr::PresetParams p;
    p.aspect_ratio = aspect_ratio;
    p.mesh_faces = mesh_faces;
    p.mesh_path = mesh_path;
    p.variant = variant;
    std::string err;
    if (!rr::build_preset(preset, p, s->flat, err)) {
        g_host_err = err;
        delete s;
        return RT_ERR_INVALID_ARG;
    }
    *out = s;
    return RT_OK;
}

int rrh_scene_destroy(rrh_scene* s) {
    delete s;
    return RT_OK;
}
const rt_scene_desc* rrh_scene_desc(const rrh_scene* s) { return s ? &s->flat.desc : nullptr; }
const rt_camera* rrh_scene_camera(const rrh_scene* s) { return s ? &s->flat.camera.c : nullptr; }
const char* rrh_scene_name(const rrh_scene* s) { return s ? s->flat.name.c_str() : ""; }

int rrh_camera_new(const double* from, const double* to, const double* up, double aspect_ratio, double vfov,
                   double aperture, double focus_dist, double t0, double t1, rt_camera* out) {
    if (!from || !to || !up || !out) return RT_ERR_INVALID_ARG;
    rr::Camera c = rr::Camera::new_motion_blur({from[0], from[1], from[2]}, {to[0], to[1], to[2]},
                                               {up[0], up[1], up[2]}, aspect_ratio, vfov, aperture, focus_dist, t0, t1);
    *out = c.c;
    return RT_OK;
}

// Commit a flattened scene into a GPU scene handle through the C ABI.
int rrh_scene_upload(rt_context* ctx, const rt_scene_desc* d, rt_scene** out) {
    return rrh_scene_upload_ex(ctx, d, RT_COMMIT_HOST_SAH, out);
}

int rrh_scene_upload_ex(rt_context* ctx, const rt_scene_desc* d, uint32_t commit_flags, rt_scene** out) {
    if (!ctx || !d || !out) return RT_ERR_INVALID_ARG;
    rt_scene* s = nullptr;
    int rc = rt_scene_create(ctx, &s);
    if (rc != RT_OK) return rc;
    if ((rc = rt_scene_set_meshes(s, d->meshes, d->n_meshes)) != RT_OK ||
        (rc = rt_scene_set_primitives(s, d->prims, d->n_prims)) != RT_OK ||
        (rc = rt_scene_set_transforms(s, d->xforms, d->n_xforms)) != RT_OK ||
        (rc = rt_scene_set_materials(s, d->materials, d->n_materials)) != RT_OK ||
        (rc = rt_scene_set_textures(s, d->textures, d->n_textures)) != RT_OK ||
        (rc = rt_scene_set_lights(s, d->lights, d->n_lights)) != RT_OK || (rc = rt_scene_commit_ex(s, commit_flags)) != RT_OK) {
        rt_scene_destroy(s);
        return rc;
    }
    *out = s;
    return RT_OK;
}

// render::tile_multithread's GPU sibling: one rt_render call for the whole image.
int rrh_gpu_tile(rt_context* ctx, rt_scene* scene, const rt_camera* camera, uint32_t width, uint32_t height,
                 uint32_t samples_per_pixel, uint32_t max_depth, uint64_t seed, double* rgb_sum, uint32_t* n,
                 rt_stats* stats) {
    rt_render_cfg cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.width = width;
    cfg.height = height;
    cfg.spp = samples_per_pixel;
    cfg.max_depth = max_depth;
    cfg.seed = seed;
    cfg.tile_size = 16;  // consts.rs:10
    cfg.tile_world = 1;
    return rt_render(ctx, scene, camera, &cfg, rgb_sum, n, stats);
}


// ---- PNG (RFC 2083) with stored deflate blocks (RFC 1951 3.2.4): no compression library needed
static uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return crc;
}
static void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
static void put_chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
    put_be32(out, (uint32_t)data.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, crc32_update(0xffffffffu, &out[start], out.size() - start) ^ 0xffffffffu);
}

int rrh_write_png(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height) {
    if (!path || !rgb8 || width == 0 || height == 0) {
        g_host_err = "rrh_write_png: bad argument";
        return RT_ERR_INVALID_ARG;
    }
    // raw scanlines: filter byte 0 + RGB
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * (1 + 3 * (size_t)width));
    for (uint32_t y = 0; y < height; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb8 + (size_t)y * width * 3, rgb8 + (size_t)(y + 1) * width * 3);
    }
    std::vector<uint8_t> z;  // zlib stream: header, stored blocks of <= 65535 bytes, adler32
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t pos = 0; pos < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xff)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xff)); z.push_back((uint8_t)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = 0; i < n; i++) {
            a = (a + raw[pos + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        pos += n;
    }
    put_be32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, width);
    put_be32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    put_chunk(out, "IHDR", ihdr);
    put_chunk(out, "IDAT", z);
    put_chunk(out, "IEND", {});
    FILE* f = std::fopen(path, "wb");
    if (!f || std::fwrite(out.data(), 1, out.size(), f) != out.size()) {
        if (f) std::fclose(f);
        g_host_err = std::string("rrh_write_png: cannot write ") + path;
        return RT_ERR_INVALID_ARG;
    }
    std::fclose(f);
    return RT_OK;
}

}  // extern "C"

/*
 * rt_host.h -- C entry points of the host-side mirror that stands in for the
 * reference's Rust host (scene presets + render driver) above the C ABI.
 * A Rust build would not use these: it flattens its own `Objects` and calls
 * rt_abi.h directly.  They exist so that tests and bench.py can
 * build the reference's presets (src/scenes.rs) without a Rust toolchain.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rrh_scene rrh_scene;

/* preset: "cornell_box" (scenes.rs:89-197), "cornell_box_spheres" (C1s),
 * "cornell_box_statue" (scenes.rs:200-307), "plastic_dragon" (scenes.rs:310-375),
 * "sphere_roughness" (scenes.rs:474-546), "two_dragons" (scenes.rs:549-624),
 * "material_hdr" (scenes.rs:627-741; variant = mat_num, mesh_path = the
 * reference's data/material directory), "teapot_hdr" (scenes.rs:744-808;
 * mesh_path = a directory with models/Mesh00{0,1}.obj + textures/envmap.hdr,
 * procedural stand-ins for what is missing).
 * mesh_faces: procedural stand-in face count (0 = preset default);
 * mesh_path: OBJ to load instead (NULL = procedural); variant: see scenes.cpp.   */
int rrh_scene_build(const char* preset, double aspect_ratio, uint64_t mesh_faces, const char* mesh_path,
                    int variant, rrh_scene** out);
int rrh_scene_destroy(rrh_scene* s);
const rt_scene_desc* rrh_scene_desc(const rrh_scene* s);
const rt_camera* rrh_scene_camera(const rrh_scene* s);
const char* rrh_scene_name(const rrh_scene* s);
const char* rrh_last_error(void);

/* Camera::new_motion_blur (src/geometry.rs:133-175) */
int rrh_camera_new(const double* from, const double* to, const double* up, double aspect_ratio, double vfov,
                   double aperture, double focus_dist, double t0, double t1, rt_camera* out);

/* rt_scene_create + set_* + commit in one call */
int rrh_scene_upload(rt_context* ctx, const rt_scene_desc* desc, rt_scene** out);
/* same with rt_scene_commit_ex flags (RT_COMMIT_DEVICE_LBVH: BVH built on the GPU) */
int rrh_scene_upload_ex(rt_context* ctx, const rt_scene_desc* desc, uint32_t commit_flags, rt_scene** out);

/* GPU sibling of render::tile_multithread (src/render.rs:13): whole image, one call */
int rrh_gpu_tile(rt_context* ctx, rt_scene* scene, const rt_camera* camera, uint32_t width, uint32_t height,
                 uint32_t samples_per_pixel, uint32_t max_depth, uint64_t seed, double* rgb_sum, uint32_t* n,
                 rt_stats* stats);

/* util::draw_picture's last step (src/util.rs:387-398 saves through the `image`
 * crate): the 8-bit picture of rt_resolve_rgb8 as an RGB PNG (8 bits, no
 * interlace, stored deflate blocks).  Next-row f1.                               */
int rrh_write_png(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */

// rr_host.hpp -- C++ stand-in for the reference's Rust host side above the C ABI.
//
// The reference builds its scene through Rust constructors and a process-global
// `Objects` (src/geometry.rs:13-55).  No Rust toolchain exists in this image, so
// the host side of the drop-in is written in C++ with the SAME names, argument
// order and meaning as the reference's constructors, and flattens the result into
// the POD arrays of include/rt_abi.h:
//   Camera::new / new_motion_blur          src/geometry.rs:110-175
//   Primitive::new_sphere / new_*_rect*    src/primitive.rs:64-233
//   Cube::new_transform / get_sides        src/hittable.rs:755-846
//   Mesh::generate_triangles               src/hittable.rs:257-288
//   Material::make_*                       src/material.rs:401-516
//   Texture::new_solid_color / _checkered  src/material.rs:618-627
//   Light::make_diffuse_light              src/light.rs:585-606
//   Texture::new_hdr, Light::make_infinite_light   src/material.rs:631-641, src/light.rs:608-638 (row f4)
//   scenes::*                              src/scenes.rs
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/rt_abi.h"

namespace rr {

struct Vec3 {
    double x, y, z;
};

// 4x4 row-major; stands for nalgebra Projective3<f64> / Matrix4<f64>.
struct Mat4 {
    double m[16];
    static Mat4 identity();
    static Mat4 from_rows(const double (&r)[16]);                      // Matrix4::new (row-major arguments)
    static Mat4 translation(double x, double y, double z);             // Matrix4::append_translation
    static Mat4 from_euler_angles(double roll, double pitch, double yaw);  // Rotation3::from_euler_angles
    static Mat4 from_scaling(double s);                                // Similarity3::from_scaling
    static Mat4 similarity(Vec3 translation, double scaling);          // Similarity3::new(t, 0, s)
    Mat4 operator*(const Mat4& o) const;
    Mat4 affine_inverse() const;
    Vec3 transform_point(Vec3 p) const;
    Vec3 transform_vector(Vec3 v) const;
};

struct BoundingBox {
    double min[3], max[3];
};

// src/geometry.rs:95-209
struct Camera {
    rt_camera c;
    static Camera create(Vec3 from, Vec3 to, Vec3 up, double aspect_ratio, double vfov, double aperture,
                         double focus_dist);
    static Camera new_motion_blur(Vec3 from, Vec3 to, Vec3 up, double aspect_ratio, double vfov, double aperture,
                                  double focus_dist, double t0, double t1);
};

struct Mesh {  // src/hittable.rs:242-250
    std::vector<double> p, n, uv;
    std::vector<uint32_t> ind;
};

struct Objects;

// src/primitive.rs:10-61.  FlipFace{obj} is carried as `flip` on the wrapped primitive.
struct Primitive {
    rt_primitive r;
    int has_xform = 0;
    rt_xform xform;

    static Primitive new_sphere(Vec3 center, double radius, uint32_t mat_index);
    static Primitive new_xy_rect(double x0, double y0, double x1, double y1, double k, uint32_t mat_index);
    static Primitive new_xz_rect(double x0, double z0, double x1, double z1, double k, uint32_t mat_index);
    static Primitive new_yz_rect(double y0, double z0, double y1, double z1, double k, uint32_t mat_index);
    static Primitive new_xy_rect_transform(double x0, double y0, double x1, double y1, double k, uint32_t mat_index,
                                           const Mat4* transform);
    static Primitive new_xz_rect_transform(double x0, double z0, double x1, double z1, double k, uint32_t mat_index,
                                           const Mat4* transform);
    static Primitive new_yz_rect_transform(double y0, double z0, double y1, double z1, double k, uint32_t mat_index,
                                           const Mat4* transform);
    static Primitive new_flip_face(Primitive obj);
    void set_light_index(int32_t index) { r.light_index = index; }
    double area(const Objects& objs) const;  // src/primitive.rs:339-359
};

// src/hittable.rs:755-846
struct Cube {
    Vec3 min, max;
    uint32_t mat_index;
    Mat4 transform;
    static Cube new_transform(Vec3 min, Vec3 max, uint32_t mat_index, const Mat4& transform);
    std::vector<Primitive> get_sides() const;
};

struct Objects;
namespace Texture {
rt_texture new_solid_color(Vec3 color);
rt_texture new_checkered(uint32_t even, uint32_t odd, double frequency);
// material.rs:631-641 Texture::new_hdr: Radiance .hdr -> Rgb<f32> (image 0.23.12 HdrDecoder::read_image_hdr) ->
// per texel image::hdr::to_rgbe8, the only form Texture::get_value reads (material.rs:577-585).  The texels are
// kept alive by `objs`.  Returns false + err when the file cannot be read.
bool new_hdr(Objects& objs, const std::string& path, rt_texture& out, std::string& err);
// Deterministic stand-in environment (sky gradient, sun, ground) for machines without the reference's data/
rt_texture new_hdr_procedural(Objects& objs, uint32_t width, uint32_t height);
// image::hdr helpers (f32 arithmetic, as the crate): Rgbe8Pixel::to_hdr and to_rgbe8
void rgbe_to_hdr(const uint8_t* rgbe, float* rgb);
void hdr_to_rgbe8(const float* rgb, uint8_t* rgbe);
}  // namespace Texture

namespace Material {
rt_material make_matte(uint32_t k_d_id, double sigma, uint32_t bump_id);
rt_material make_light(uint32_t texture_id);
rt_material make_plastic(uint32_t k_d_id, uint32_t k_s_id, uint32_t bump_id, double roughness, bool remap);
rt_material make_glass(uint32_t k_r_id, uint32_t k_t_id, double u_roughness, double v_roughness, double index,
                       uint32_t bump_id, bool remap);
rt_material make_metal(uint32_t eta_id, uint32_t k_id, uint32_t u_r_id, uint32_t v_r_id, uint32_t r_id,
                       uint32_t bump_id, bool remap);
rt_material make_mirror(uint32_t color_id, uint32_t bump_id);
}  // namespace Material

// src/geometry.rs:13-21
struct Objects {
    std::vector<Mesh> meshes;
    std::vector<Primitive> objs;
    std::vector<rt_light> lights;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<std::shared_ptr<std::vector<uint8_t>>> hdr_store;  // texels of Texture::Hdr entries
    std::vector<std::pair<uint32_t, rt_xform>> light_xforms;       // to_world of Light::Infinite entries
};

namespace Light {
// src/light.rs:585-606 (to_world is the identity in every preset and is not read on the path)
rt_light make_diffuse_light(const Objects& objs, uint32_t prim_index, Vec3 color, uint32_t n_samples,
                            bool two_sided, bool is_mesh);
// src/light.rs:608-638; to_world == nullptr is Projective3::identity().  The Distribution2D the reference
// builds here is rebuilt by the library at rt_scene_commit from the texture (include/rt_abi.h).
rt_light make_infinite_light(Objects& objs, const Mat4* to_world, uint32_t n_samples, uint32_t text_id);
}

// Mesh::generate_triangles (src/hittable.rs:257-288)
std::vector<Primitive> generate_triangles(const std::vector<Mesh>& meshes, uint32_t mesh_index, uint32_t mat_index);

// Procedural stand-in for the meshes missing from the checkout (SURVEY.md fact 8,
// section 8d "P-N"): geodesic icosphere with >= n_faces faces trimmed to exactly
// n_faces, radial displacement by 5 octaves of hash value noise (seed 1234),
// area-weighted vertex normals, no uvs, bbox normalised to
// [-0.5,0.5]*(1,0.7,0.45); positions rounded to f32 then widened like tobj
// (src/parser.rs:25-27) before `trans` is applied (src/parser.rs:29,45).
Mesh procedural_mesh(uint64_t n_faces, const Mat4& trans);
// tobj-style OBJ ingestion (src/parser.rs:8-87): first model, triangulated, single index.
bool parse_obj(const std::string& path, const Mat4& trans, Mesh& out, std::string& err);

// Flattened scene that owns its arrays; `desc` points into them.
struct FlatScene {
    std::vector<rt_mesh> meshes;
    std::vector<Mesh> mesh_store;
    std::vector<rt_primitive> prims;
    std::vector<rt_xform> xforms;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_light> lights;
    std::vector<std::shared_ptr<std::vector<uint8_t>>> hdr_store;
    rt_scene_desc desc;
    Camera camera;
    std::string name;
    void build(Objects&& objs);
};

struct PresetParams {
    double aspect_ratio = 1.0;
    uint64_t mesh_faces = 0;      // procedural face count (0 = preset default)
    const char* mesh_path = nullptr;  // OBJ to load instead of the procedural mesh
    int variant = 0;              // preset-specific material variant
};

// scenes.rs presets.  Returns false + err for an unknown name or unreadable OBJ.
bool build_preset(const std::string& name, const PresetParams& p, FlatScene& out, std::string& err);

}  // namespace rr

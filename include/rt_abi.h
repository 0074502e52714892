/*
 * rt_abi.h -- C ABI of the MI355X path-tracing core (librt_amd.so).
 *
 * This is the drop-in boundary for the per-pixel path-tracing hot path of
 * calvin-godfrey/RustRaytracer.  The reference has no FFI seam of its own
 * (SURVEY.md section 8b): its render driver calls
 *     get_integrator(IntType, Camera, Samplers)   src/integrator.rs:55-87
 *     Integrator::render(&mut grid, px, py)       src/integrator.rs:336-348
 * once per pixel from src/render.rs:46-47,85 and reads the scene from the
 * process-global `Objects` (src/geometry.rs:13-55).  The replacement is called
 * once per image instead: a Rust maintainer adds `render::gpu_tile` beside
 * `render::tile_multithread` (src/render.rs:13), flattens `get_objects()` into
 * the POD arrays below, calls rt_render() and hands the returned sums to
 * `util::draw_picture` (src/util.rs:387-398).  INTEGRATION.md shows that stub.
 *
 * Every struct here is plain old data with fixed-width fields; every function
 * returns an int status (RT_OK or a negative rt_status) and never unwinds.
 *
 * Numerical contract (part of the ABI, shared with the CPU oracle):
 *   - all arithmetic of the parity mode is IEEE binary64 without FMA
 *     contraction, in the reference's expression order;
 *   - elementary functions are the ones in rt_detmath.h;
 *   - random numbers come from the counter generator specified at rt_rng_*
 *     below, drawn in the order of SURVEY.md section 3.3;
 *   - exact-t ties between two primitives are won by the larger prim index
 *     (the reference's own winner is run-to-run random, hittable.rs:604-616,
 *     645-652);
 *   - a ray with a NaN origin or direction component misses (in the reference it
 *     passes every slab test, f64::min/max dropping NaN, and "hits" whichever
 *     triangle its randomly built tree visits first).
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 5  /* 4 (round 4): rt_stats.classify_ms, rt_scene_info.n_classes appended; 5: rt_stats.light_ms */

/* ---------------------------------------------------------------- status */
typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARG = -1, /* null pointer, out-of-range index, bad enum     */
    RT_ERR_NO_DEVICE = -2,   /* no usable HIP device / device id out of range  */
    RT_ERR_HIP = -3,         /* a HIP runtime call failed (see rt_last_error)  */
    RT_ERR_UNSUPPORTED = -4, /* valid but outside the hot-path scope           */
    RT_ERR_STATE = -5,       /* call order violated (e.g. render before commit)*/
    RT_ERR_OOM = -6,         /* host or device allocation failed               */
    RT_ERR_CANCELLED = -7    /* rt_render_cfg.cancel became non-zero; the film is
                                incomplete (the reference's stop_render poll,
                                render.rs:93)                                   */
} rt_status;

/* ------------------------------------------------------------- constants */
/* src/consts.rs:14-20 (BxDF flag bits) -- values are part of the contract.   */
#define RT_BSDF_REFLECTION 1u
#define RT_BSDF_TRANSMISSION 2u
#define RT_BSDF_DIFFUSE 4u
#define RT_BSDF_GLOSSY 8u
#define RT_BSDF_SPECULAR 16u
#define RT_BSDF_ALL 31u
/* src/consts.rs:7,30-32 */
#define RT_MAX_DEPTH 25u
#define RT_INFINITY 1e308
#define RT_PI 3.14159265358979 /* truncated on purpose: src/consts.rs:31 */
#define RT_SMALL 0.001

/* --------------------------------------------------------------- scene  */
/* Affine 3x4 transform, row-major: p' = M[0..2][0..2]*p + M[.][3].
 * Mirrors the `Projective3<f64>` carried by transformed rects
 * (src/primitive.rs:32,43,54).  `inv` is the inverse the reference recomputes
 * on every test through Ray::transform (src/geometry.rs:231-235).            */
typedef struct rt_xform {
    double fwd[12];
    double inv[12];
} rt_xform;

/* src/primitive.rs:10-61.  FlipFace{obj} is the `flip` bit on the inner
 * primitive (only ever wraps rects in the reference's presets).              */
typedef enum rt_prim_kind {
    RT_PRIM_SPHERE = 0,   /* v[0..2] = center, v[3] = r                        */
    RT_PRIM_TRIANGLE = 1, /* mesh_index, tri_ind = 3*face (Mesh.ind offset)    */
    RT_PRIM_XY_RECT = 2,  /* v = x0,y0,x1,y1,k                                 */
    RT_PRIM_XZ_RECT = 3,  /* v = x0,z0,x1,z1,k                                 */
    RT_PRIM_YZ_RECT = 4   /* v = y0,z0,y1,z1,k                                 */
} rt_prim_kind;

typedef struct rt_primitive {
    uint32_t kind;        /* rt_prim_kind                                      */
    uint32_t flip;        /* 1 = wrapped in Primitive::FlipFace                */
    uint32_t mat_index;
    int32_t light_index;  /* -1 = usize::MAX = not an emitter                  */
    uint32_t mesh_index;  /* triangles only                                    */
    uint32_t tri_ind;     /* triangles only: offset into Mesh.ind (3*face)     */
    int32_t xform_index;  /* rects only: -1 = None                             */
    uint32_t reserved;
    double v[5];
    double bbox_min[3];   /* Primitive::get_bounding_box, f64 (hittable.rs:274,
                             primitive.rs:66-68,93-113, util.rs:493-517)       */
    double bbox_max[3];
} rt_primitive;

/* src/hittable.rs:242-250: ith triangle = p[ind[3i]], p[ind[3i+1]], ...      */
typedef struct rt_mesh {
    const double* p;      /* n_p * 3                                           */
    const double* n;      /* n_n * 3, n_n == 0 or n_p (un-normalised allowed)  */
    const double* uv;     /* n_uv * 2, n_uv == 0 or n_p                        */
    const uint32_t* ind;  /* n_ind, multiple of 3                              */
    uint64_t n_p, n_n, n_uv, n_ind;
} rt_mesh;

/* src/material.rs:519-539; the kinds the hot-path scope covers.                */
typedef enum rt_texture_kind {
    RT_TEX_SOLID = 0,     /* color                                             */
    RT_TEX_CHECKERED = 1, /* odd, even texture ids + frequency                 */
    RT_TEX_HDR = 2        /* Texture::Hdr (next-row f4): width x height texels
                             of 4 bytes, row-major from the top: what
                             image::hdr::to_rgbe8(data[i]) returns for texel i
                             (c[0], c[1], c[2], e), which is all that
                             Texture::get_value reads (material.rs:570-587)    */
} rt_texture_kind;

typedef struct rt_texture {
    uint32_t kind;
    uint32_t odd;         /* Checkered.odd  (material.rs:527-531)              */
    uint32_t even;
    uint32_t reserved;
    double color[3];
    double frequency;
    const uint8_t* rgbe;  /* RT_TEX_HDR only, copied by rt_scene_set_textures  */
    uint32_t width, height;
} rt_texture;

/* src/material.rs:17-73 */
typedef enum rt_material_kind {
    RT_MAT_MATTE = 0,    /* tex[0]=k_d; f[0]=sigma (only sigma==0 supported)   */
    RT_MAT_LIGHT = 1,    /* no lobes                                           */
    RT_MAT_PLASTIC = 2,  /* tex[0]=k_d, tex[1]=k_s; f[0]=roughness             */
    RT_MAT_GLASS = 3,    /* tex[0]=k_r, tex[1]=k_t; f[0]=u_rough f[1]=v_rough
                            f[2]=index; rough = MicrofacetReflection +
                            MicrofacetTransmission (material.rs:153-189)       */
    RT_MAT_METAL = 4,    /* tex[0]=eta, tex[1]=k, tex[2]=rough, tex[3]=urough,
                            tex[4]=vrough (RT_NO_TEXTURE = usize::MAX)         */
    RT_MAT_MIRROR = 5    /* tex[0]=color                                       */
} rt_material_kind;

#define RT_NO_TEXTURE 0xFFFFFFFFu

typedef struct rt_material {
    uint32_t kind;
    uint32_t remap_roughness;
    uint32_t tex[5];
    uint32_t reserved;
    double f[3];
} rt_material;

/* src/light.rs:59-93: Light::Diffuse, and Light::Infinite (next-row f4).       */
typedef enum rt_light_kind {
    RT_LIGHT_DIFFUSE = 0,
    RT_LIGHT_INFINITE = 1 /* environment map: tex_index = an RT_TEX_HDR texture,
                             xform_index = to_world (-1 = identity),
                             world_radius = 10000 in make_infinite_light
                             (light.rs:608-638); at most one per scene.  The
                             library rebuilds the Distribution2D of
                             make_infinite_light from the texture at commit.   */
} rt_light_kind;

typedef struct rt_light {
    uint32_t kind;
    uint32_t prim_index;  /* Diffuse                                           */
    uint32_t two_sided;   /* Diffuse                                           */
    uint32_t tex_index;   /* Infinite                                          */
    int32_t xform_index;  /* Infinite                                          */
    uint32_t reserved;
    double color[3];      /* Diffuse                                           */
    double area;          /* Diffuse: Primitive::area() at construction
                             (light.rs:603)                                    */
    double world_radius;  /* Infinite                                          */
} rt_light;

/* Flattened `Objects` (src/geometry.rs:13-21), caller-owned.                  */
typedef struct rt_scene_desc {
    const rt_mesh* meshes;        uint64_t n_meshes;
    const rt_primitive* prims;    uint64_t n_prims;
    const rt_xform* xforms;       uint64_t n_xforms;
    const rt_material* materials; uint64_t n_materials;
    const rt_texture* textures;   uint64_t n_textures;
    const rt_light* lights;       uint64_t n_lights;
} rt_scene_desc;

/* The ten fields of `Camera` (src/geometry.rs:96-108), precomputed by the
 * caller with Camera::new_motion_blur (geometry.rs:133-175).                  */
typedef struct rt_camera {
    double origin[3];
    double upper_left_corner[3];
    double horizontal_offset[3];
    double vertical_offset[3];
    double lens_radius;
    double t0, t1;
    double u[3], v[3], w[3];
} rt_camera;

/* ---------------------------------------------------------------- render */
typedef enum rt_precision {
    RT_PRECISION_F64 = 0, /* parity mode: bit-comparable with the oracle       */
    RT_PRECISION_F32 = 1  /* fast mode: the same kernels in binary32 with hardware-rate
                             elementary functions (scene records, path state and the
                             film stay binary64 in memory).  Not bit-comparable: paths
                             diverge from the f64 ones after a few bounces, so the
                             per-pixel error at low spp is Monte-Carlo noise, not
                             rounding; reported by bench.py, not gated by the tests'
                             RMSE < 1e-4 bar (tolerance row).          */
} rt_precision;

typedef struct rt_render_cfg {
    uint32_t width, height;   /* GLOBAL_STATE image size (main.rs:82-88)       */
    uint32_t spp;             /* rounded up to a power of two like
                                 sampler.rs:633-642                            */
    uint32_t max_depth;       /* 25 in the reference (consts.rs:7)             */
    uint64_t seed;
    uint32_t x0, y0, x1, y1;  /* pixel window [x0,x1) x [y0,y1); 0,0,0,0 = all */
    uint32_t tile_size;       /* 16 (consts.rs:10); 0 = 16                     */
    uint32_t tile_rank;       /* this caller renders the tiles (tx, ty) with   */
    uint32_t tile_world;      /*   rt_tile_owner(tx, ty, tile_world) ==
                                   tile_rank (below); 0 = 1                    */
    uint32_t precision;       /* rt_precision                                  */
    uint32_t paths_in_flight; /* path-state slots kept alive per device (about
                                   0.85 KB each); 0 = library default: a whole
                                   batch, at most 2^28 (238 GB), halved until it
                                   fits and leaves 12 GB of the device's free
                                   memory to everybody else                      */
    uint32_t flags;           /* RT_RENDER_* below                             */
    /* Progressive passes (next-row f4; render.rs:161-324 refines the picture while
     * it is displayed): this call renders samples [sample_first, sample_first +
     * sample_count) of every pixel; sample_count 0 = all the remaining ones.
     * Sample k of a pixel is the same path whichever pass renders it.          */
    uint32_t sample_first;
    uint32_t sample_count;
    /* Optional cancellation flag (NULL = none), the counterpart of GLOBAL_STATE.stop_render that the
     * reference's workers poll per pixel (render.rs:93, main.rs:315-321).  The library polls it between
     * kernel launches; once it reads non-zero the call drains the device and returns RT_ERR_CANCELLED.  */
    const volatile int32_t* cancel;
} rt_render_cfg;

/* Which of `world` callers owns tile (tx, ty) (tiles of tile_size x tile_size pixels, tx and ty counted from the
 * image's upper left corner).  The reference hands tiles to its workers in row-major order (render.rs:49-71);
 * "tile k -> k % world" would do the same here, but with a tile row that is a multiple of `world` long (1920 / 16 =
 * 120 tiles for 2, 4, 8 GPUs) that deals whole tile COLUMNS to a rank, and the cost of a picture is not spread
 * evenly over its columns (measured on two_dragons, 8 ranks: slowest rank 2.3 % above the mean).  So the owner is
 * a rank-1 lattice instead: (tx + ty * s) % world, s = the integer nearest to world / golden ratio that is coprime
 * to world -- every row and every column of tiles visits all ranks in turn.  Pixels do not
 * depend on who renders them, so any rule gives the same film; this one is part of the interface because the
 * gather (rustraytracer_amd/dist.py, a host's own) has to know it: rt_tile_owner() below exports it, the inline
 * form is for code that does not link the library.                                                              */
static inline uint32_t rtabi_tile_stride(uint32_t world) {
    if (world <= 1) return 0;
    uint32_t s = (uint32_t)(((uint64_t)world * 618034u + 500000u) / 1000000u);
    if (s == 0) s = 1;
    for (;; s++) {
        uint32_t a = s, b = world;
        while (b) { const uint32_t t = a % b; a = b; b = t; }
        if (a == 1) return s % world;
    }
}
static inline uint32_t rtabi_tile_owner(uint32_t tx, uint32_t ty, uint32_t world) {
    if (world <= 1) return 0;
    return (uint32_t)(((uint64_t)tx + (uint64_t)ty * rtabi_tile_stride(world)) % world);
}

#define RT_RENDER_COUNT_TRAVERSAL 1u /* fill the node/prim test counters       */
#define RT_RENDER_ACCUMULATE 2u      /* add to the film passed in instead of
                                        zeroing it: passes rendered in sample
                                        order give the one-shot film bit for bit */

/* Counters.  rays_* are the three root closest-hit call sites of
 * R1 integrator.rs:388, R2 integrator.rs:584 (hittable.rs:30),
 * R3 integrator.rs:615.  They must equal the oracle's exactly.                */
typedef struct rt_stats {
    uint64_t paths;            /* camera samples started                       */
    uint64_t rays_extension;   /* R1, primary included                         */
    uint64_t rays_shadow;      /* R2                                           */
    uint64_t rays_probe;       /* R3                                           */
    uint64_t vertices_shaded;  /* compute_scattering calls                     */
    uint64_t nodes_fetched;    /* BVH nodes fetched (RT_RENDER_COUNT_TRAVERSAL)*/
    uint64_t tris_tested;
    uint64_t others_tested;    /* sphere / rect records tested                 */
    double kernel_ms;          /* all device work of this call (HIP events)    */
    double trace_ms;           /* traversal kernel only (HIP events)           */
    uint64_t trace_launches;
    /* The last few paths of a batch are finished by one fused launch (k_tail) instead of per-bounce
     * launches.  rays_* and, with RT_RENDER_COUNT_TRAVERSAL, nodes_fetched / tris_tested / others_tested
     * cover ALL rays; the four fields below are the fused launch's share, so that
     * (total - tail) is exactly what the traversal kernel timed by trace_ms processed.                 */
    uint64_t tail_rays;
    uint64_t tail_nodes_fetched;
    uint64_t tail_tris_tested;
    uint64_t tail_others_tested;
    /* Multi-device contexts: counters are summed over the devices, kernel_ms is the slowest device's
     * (they run side by side), trace_ms / trace_launches are summed (their ratio stays the mean launch
     * time), gather_ms is the host-clock time of the final peer-to-peer film gather.                   */
    double gather_ms;
    uint64_t n_devices;
    double shade_ms;           /* the class kernels only (HIP events), like trace_ms: per bounce one kernel
                                  per vertex class of the scene; the light kernel is in light_ms        */
    uint64_t shade_launches;   /* bounces shaded (not kernels)                 */
    double classify_ms;        /* (ABI 4) the counting sort that deals the traced paths to the lists of
                                  their vertex classes, between the two (HIP events)                  */
    double light_ms;           /* (ABI 5) the kernel for escaped / fold-only paths (HIP events on its own
                                  stream).  It runs BESIDE the class kernels and the next traversal
                                  launch, so this time overlaps trace_ms / shade_ms: the four do not add
                                  up to kernel_ms                                                      */
} rt_stats;

typedef struct rt_ray {
    double origin[3];
    double dir[3];
    double tmin;  /* SMALL for extension rays, 0 for shadow rays (Q4)          */
    double tmax;  /* RT_INFINITY                                               */
} rt_ray;

typedef struct rt_hit {
    double t;       /* RT_INFINITY on miss                                     */
    int32_t prim;   /* -1 on miss                                              */
    uint32_t reserved; /* diagnostic: BVH nodes | triangles << 8 | spheres/rects << 16
                          tested for this ray (each saturating at 255)            */
} rt_hit;

typedef struct rt_context rt_context;
typedef struct rt_scene rt_scene;

/* = rtabi_tile_owner(tx, ty, world): the rank that renders tile (tx, ty) under rt_render_cfg.tile_world = world */
uint32_t rt_tile_owner(uint32_t tx, uint32_t ty, uint32_t world);

/* device_ids may be NULL with n_devices == 0 (device 0).  With n_devices > 1 the
 * context spans several GPUs of one node: scenes are
 * replicated at commit, rt_render / rt_render_device run one host thread per
 * device on interleaved tiles and gather the film onto device_ids[0] with
 * direct peer-to-peer copies; the film pointers of rt_render_device live on
 * device_ids[0].  The result is bit-identical to a one-device render.  (The
 * alternative, one process per GPU with its own one-device context and
 * rt_render_cfg.tile_rank / tile_world, is what bench.py uses.)               */
int rt_context_create(const int* device_ids, int n_devices, rt_context** out);
int rt_context_destroy(rt_context* ctx);

/* Scene = flattened geometry::Objects.  The library copies on set_*.          */
int rt_scene_create(rt_context* ctx, rt_scene** out);
int rt_scene_set_meshes(rt_scene* s, const rt_mesh* meshes, uint64_t count);
int rt_scene_set_primitives(rt_scene* s, const rt_primitive* prims, uint64_t count);
int rt_scene_set_transforms(rt_scene* s, const rt_xform* xforms, uint64_t count);
int rt_scene_set_materials(rt_scene* s, const rt_material* mats, uint64_t count);
int rt_scene_set_textures(rt_scene* s, const rt_texture* texs, uint64_t count);
int rt_scene_set_lights(rt_scene* s, const rt_light* lights, uint64_t count);
/* Validates indices, builds the BVH (replaces BvhNode::new,
 * hittable.rs:637-752) and uploads everything to HBM.                         */
int rt_scene_commit(rt_scene* s);
/* Next-row f3: choose the BVH builder.  Both produce the same device layout and,
 * because a primitive is gated only by the f64 test of its own box, the same
 * film bit for bit; they differ in build time and traversal cost.
 *   RT_COMMIT_HOST_SAH     binned SAH on the host (default, best tree)
 *   RT_COMMIT_DEVICE_LBVH  Morton-order linear BVH built on the GPU from the
 *                          uploaded primitives (milliseconds; for callers that
 *                          re-commit geometry often)                            */
#define RT_COMMIT_HOST_SAH 0u
#define RT_COMMIT_DEVICE_LBVH 1u
int rt_scene_commit_ex(rt_scene* s, uint32_t flags);
int rt_scene_destroy(rt_scene* s);
/* Sizes of the committed device layout, for roofline accounting.             */
typedef struct rt_scene_info {
    uint64_t n_prims, n_triangles, n_others, n_bvh_nodes, bvh_depth;
    uint64_t node_bytes, tri_bytes, other_bytes, device_bytes_total;
    uint64_t build_flags;    /* RT_COMMIT_* the scene was committed with          */
    double build_ms;         /* BVH build + leaf layout + their upload, host clock */
    double build_device_ms;  /* RT_COMMIT_DEVICE_LBVH: device time of the build   */
    uint64_t build_from_cache; /* (ABI 3) 1: the host tree was read from the node-local cache
                                  another process published (environment RT_BVH_CACHE=<dir>:
                                  the ranks of one node build a scene's tree once)          */
    uint64_t n_classes;        /* (ABI 4) vertex classes the scene's primitives fall into (+ 1: escaped):
                                  shading kernels launched per bounce                                  */
} rt_scene_info;
int rt_scene_get_info(const rt_scene* s, rt_scene_info* out);

/* Whole-image replacement for the per-pixel Integrator::render loop.
 * rgb_sum[(y*W+x)*3+c] = sum of radiance samples, n[y*W+x] = sample count:
 * the reference's `grid[y][x] = (r,g,b,n)` (integrator.rs:44, util.rs:208-232).
 * Host pointers; either may be NULL to leave the film on the device.          */
int rt_render(rt_context* ctx, rt_scene* s, const rt_camera* cam, const rt_render_cfg* cfg,
              double* rgb_sum, uint32_t* n, rt_stats* stats);
/* Same, writing into caller-owned DEVICE buffers (W*H*3 doubles, W*H u32),
 * zeroed by the call.  Everything that touches the film (zeroing, the resolve
 * of each batch) is enqueued on `hip_stream`, a hipStream_t; NULL is the
 * default (null) stream, exactly as in any HIP call -- so work the caller
 * queued on that stream before the call happens before the film is touched and
 * work queued on it afterwards sees the finished film.  The call returns after
 * the film is complete (it synchronises `hip_stream`).                         */
int rt_render_device(rt_context* ctx, rt_scene* s, const rt_camera* cam, const rt_render_cfg* cfg,
                     double* d_rgb_sum, uint32_t* d_n, void* hip_stream, rt_stats* stats);

/* Kernel-level parity entry: closest hit of BvhNode::intersects
 * (hittable.rs:591-634) for n host rays.                                      */
int rt_intersect_batch(rt_context* ctx, rt_scene* s, const rt_ray* rays, uint64_t n, rt_hit* hits);
/* Same query through other code paths (kernel-level tests):
 *   RT_INTERSECT_F32        the binary32 traversal of the fast mode
 *   RT_INTERSECT_WAVEFRONT  the render's own traversal kernel (persistent waves, queue
 *                           reservations, refill, while-while scheduling) instead of the
 *                           run-to-completion loop; rays must use tmin = RT_SMALL, tmax =
 *                           RT_INFINITY (what extension and probe rays use)             */
#define RT_INTERSECT_F32 1u
#define RT_INTERSECT_WAVEFRONT 2u
int rt_intersect_batch_ex(rt_context* ctx, rt_scene* s, const rt_ray* rays, uint64_t n, rt_hit* hits, uint32_t flags);

/* Next-row f1: film resolve -> ACES approx -> gamma -> 8-bit
 * (util.rs:400-408, 441-471).  rgb8 = W*H*3 bytes, host pointers.             */
int rt_resolve_rgb8(rt_context* ctx, const double* rgb_sum, const uint32_t* n, uint32_t width,
                    uint32_t height, uint8_t* rgb8);

const char* rt_last_error(void);
int rt_abi_version(void);

/* ------------------------------------------------------------------ RNG
 * Counter generator replacing both of the reference's unseeded streams
 * (thread_rng, sampler.rs:317-337; StdRng::from_entropy, util.rs:32-34).
 * One stream per (seed, pixel = py*W+px, sample); draw k is
 *     s0  = mix(mix(seed * G + pixel) + sample * H + J)
 *     z_k = mix(s0 + (k+1) * G)          (splitmix64 finaliser)
 *     u_k = (z_k >> 11) * 2^-53          in [0,1)
 * with G = 0x9E3779B97F4A7C15, H = 0xD1B54A32D192ED03, J = 0x8CB92BA72F3D8DD7.  */
#define RT_RNG_G 0x9E3779B97F4A7C15ull
#define RT_RNG_H 0xD1B54A32D192ED03ull
#define RT_RNG_J 0x8CB92BA72F3D8DD7ull

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */

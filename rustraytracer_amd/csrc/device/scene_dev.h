// scene_dev.h -- HBM layout of a committed scene and of the in-flight path state.
#pragma once
#include <stdint.h>

#include "../../../include/rt_abi.h"

// The types of this header live in `rtc` (no functions there, so argument-dependent lookup adds nothing): they are
// shared by the f64 kernels (namespace rtd) and the f32 fast-mode kernels generated from the same sources
// (namespace rtd32, tools/make_f32_sources.py); both namespaces import them with a using-directive.
namespace rtc {

// BVH4 node, 128 B = exactly one L2 line / HBM burst pair: the four children's AABBs (f32, rounded
// OUTWARD from the f64 primitive boxes, stored per axis so a lane reads them with eight 16-B loads)
// and four child references.  One fetch decides four descents, which halves the length of the
// dependent fetch chain of a ray compared with a binary tree -- the traversal is bound by memory
// latency x outstanding requests, not by bytes or ALU.  The f64 slab test of
// the reference (hittable.rs:494-508) is applied to the boxes after an exact widen, which keeps the
// test conservative (see geom.h).
struct DevNode {
    float lo_x[4], lo_y[4], lo_z[4];
    float hi_x[4], hi_y[4], hi_z[4];
    int32_t child[4];  // >= 0: node index; kNoChild; else leaf: -1 - ((first*8 + count-1) | kLeafCodeOther?)
    int32_t pad[4];
};
static_assert(sizeof(DevNode) == 128, "DevNode must be 128 bytes");
constexpr int32_t kNoChild = INT32_MIN;
constexpr uint32_t kLeafOther = 0x80000000u;  // leaf entry is a sphere/rect, not a triangle
constexpr uint32_t kLeafCodeOther = 1u << 30;
constexpr uint64_t kMaxPrims = 1ull << 27;  // exclusive: leaf slots occupy bits 3..29 of a leaf code, bits 0..26 of a hit word
constexpr uint32_t kMetaHasNormals = 1u << 30, kMetaFlip = 1u << 31, kMetaMatMask = 0x3fffffffu;  // leaf code flag: the leaf holds one sphere/rect (they never share a leaf)
constexpr int kMaxLeafPrims = 4;     // what a leaf code can hold
// What the builders aim for.  One primitive per leaf: the parent's f32 test of the child box then culls each
// triangle on its own, before the f64 own-box + triangle test that a leaf step costs (measured on C3: trace
// -19 %, C2: -10 % against leaves of up to 4).
constexpr int kLeafTargetPrims = 1;
constexpr int kMaxBvhDepth = 24;  // of the 4-wide tree; traversal stack: 16 LDS + 58 private entries (geom.h)

// rt_material with its texture references resolved at commit: a solid-colour texture (the usual case) is
// embedded, so compute_scattering reads material -> colour in one fetch instead of material -> texture record.
struct DevMat {
    rt_material m;
    uint32_t tex[5];      // m.tex with Metal's RT_NO_TEXTURE fallbacks applied (material.rs:193-208)
    uint32_t solid_mask;  // bit k: tex[k] is a solid colour, col[k] holds it
    double col[5][3];
};

struct LeafMeta {
    uint32_t mat_flags;  // mat_index | kMetaHasNormals | kMetaFlip
    int32_t light;       // light_index of the primitive
};

struct DevMesh {
    const double* p;
    const double* n;   // null: no vertex normals
    const double* uv;  // null: default uvs (hittable.rs:455-460)
    const uint32_t* ind;
};

// Light::Infinite (next-row f4): the Distribution2D of make_infinite_light (light.rs:608-638,
// distribution.rs:93-150), flattened.  Row v of the reference's conditional distributions is the window
// img[v .. v + nu) of the (nu x nv) luminance image (distribution.rs:118 slices `f[v..(v + nu)]`), so the
// rows share `img`; their cdfs are stored one after the other, nu + 1 entries each.
struct DevEnv {
    const double* img;        // nu * nv, luminance * sin(theta)
    const double* cond_cdf;   // nv rows of nu + 1
    const double* marg_func;  // nv: func_int of row v
    const double* marg_cdf;   // nv + 1
    double marg_int;          // p_marginal.func_int
    uint32_t nu, nv;
    int32_t light;            // index of the infinite light, -1: none
    uint32_t pad;
};

struct DevScene {
    const DevNode* nodes;
    const uint32_t* leaf_prim;  // leaf order -> prim index | vertex class << kClsShift | kLeafOther
    const double* leaf_tri;     // leaf order -> 9 doubles p0,p1,p2 (triangles), 72 B
    // The traversal kernel's own copy of a leaf slot, one aligned 128-B line (16 doubles) per slot, so that a primitive
    // test touches ONE cache line instead of 2.4 (a 72-B record straddles lines 7 times in 16, the primitive index
    // was a gather of its own) and fetches it with 6 loads instead of 10:
    //   triangle: x0 x1 x2 y0 y1 y2 z0 z1 z2 x0 x1 x2 y0 y1 y2 -- the nine coordinates in the ray's permuted axis
    //             order (kx, ky, kz) are the 72 contiguous bytes from double 3 * kx;
    //   sphere / rect: v[0..4], {kind, transform index} (as in leaf_tri);
    //   both: the last double holds the leaf_prim word (primitive index | kLeafOther) in its low half.
    const double* leaf_trav;
    // Shading side of a triangle's leaf slot, so that rebuilding the winner's hit record needs one dependent
    // fetch (slot -> vertices + normals + meta) instead of five (prim -> mesh -> indices -> positions/normals):
    const double* leaf_nrm;     // leaf order -> 9 doubles n0,n1,n2 (null: no mesh has normals)
    const LeafMeta* leaf_meta;  // leaf order -> {mat_index | kMetaHasNormals | kMetaFlip, light_index}
    const rt_primitive* prims;  // original order (shading + sphere/rect tests)
    const DevMesh* meshes;
    const rt_xform* xforms;
    const DevMat* mats;
    const rt_texture* texs;
    const rt_light* lights;
    // (round 3) the primitive record of every light, in light order (lights[i].prim_index's record; zeros for the
    // infinite light): with n_mats / n_texs it lets the shading kernels stage the scene's small tables in LDS
    const rt_primitive* light_prims;
    uint32_t n_mats, n_texs;
    uint32_t n_prims, n_lights, n_nodes, mesh_has_uv;  // mesh_has_uv: any mesh carries uvs
    uint32_t simple_others, pad_so;                   // no sphere and no transformed rect in the scene
    DevEnv env;
    // RT_PRECISION_F32 only (made by the first fast-mode render of the scene): binary32 copies of leaf_tri /
    // leaf_nrm, 9 floats per slot.  A sphere / rect slot holds v[0..4] as floats and its {kind, transform} word
    // as the bit patterns of floats 5 and 6.
    const float* leaf_tri32;
    const float* leaf_nrm32;
};

// ----------------------------------------------------------------- path state
// What bounds every access pattern on this chip is the number of L2 REQUESTS, not bytes (tools/ubench_partial_write.hip,
// profiles/r04_ubench_partial_write.txt: 50-75 G requests/s whatever their size up to 128 B; a lane that reads its own
// 128-B line with eight 16-B loads makes eight requests, eight lanes that read one line together make one).  So:
//   * what the TRAVERSAL kernel touches -- the rays and their results -- are separate arrays indexed by slot (a wave's
//     queue entries are mostly consecutive slots: a 64-lane load is 4 requests): o, d, sp, pd; sh_prim, pr_prim;
//   * what only the SHADING kernels touch is one RECORD of 256 B per slot, two 128-B lines, read and written as WHOLE
//     LINES by eight lanes each through LDS (kernels.hip: stage_*), in whatever order the class lists name the slots --
//     which is what lets the shading kernels run on class-pure waves gathered over the whole launch:
//       line 0 (every bounce): w0-2 o   w3-5 d   w6 rng   w7 {orig, flags}   w8-10 beta   w11-13 L   w14-15 spare
//       line 1 (only paths with pending direct-light terms): w16-18 A   w19-21 Q   w22-24 K   w25-31 spare
// Fields are 8-byte words (a vec3 = three consecutive words / array elements); the fast mode keeps its binary32 values
// in the low half of each.  Two such pools ping-pong per bounce (kernels.hip).
// o = ray origin (last hit point; spawn_ray adds no offset), d = extension direction, sp = sampled light point (shadow
// ray target), pd = MIS probe direction, A / Q = pending light-sample / bsdf-sample terms (f * Le * w / pdf), K = beta at
// the vertex that produced them, orig = film staging slot (sample_local * n_pixels + pixel_local).
struct PathState {
    char* rec;
    // (ox .. faz: fifteen arrays of one slab, equally spaced in this order -- the kernels address them as ox + k * (oy - ox))
    double *ox, *oy, *oz;     // ray origin (also in the record: the shading kernels read it there)
    double *dx, *dy, *dz;     // extension direction (likewise)
    double *spx, *spy, *spz;  // shadow ray target
    double *pdx, *pdy, *pdz;  // MIS probe direction
    double *fax, *fay, *faz;  // a path whose only pending term is the light sample: what that term adds to L if the shadow
                              // ray finds the light, (A * n_lights) (*) K, complete -- such a path has no line 1
    int32_t* sh_prim;         // closest prim along the shadow ray (Q13)
    int32_t* pr_prim;         // closest prim along the probe ray
    uint64_t* rng0;           // RNG state of a camera sample after its camera draws (k_generate; see kEntFresh)
};
constexpr uint32_t kRecBytes = 256;
constexpr int kWO = 0, kWD = 3, kWRng = 6, kWMeta = 7, kWBeta = 8, kWL = 11, kWA = 16, kWQ = 19, kWK = 22;
// flags
constexpr uint32_t kBounceMask = 0xffu;
constexpr uint32_t kSpecular = 1u << 8;
constexpr uint32_t kFoldOnly = 1u << 9;
constexpr uint32_t kHasShadow = 1u << 10;
constexpr uint32_t kHasProbe = 1u << 11;
constexpr uint32_t kLine1 = 1u << 12;  // the pending terms are in line 1 (A, Q, K); else only a light sample is pending: fa* words
constexpr uint32_t kLightShift = 16;

// queue entry = slot | kQPending | kind << 30; kRayNone fills the unused end of a wave's queue chunk.  kQPending (extension
// entries): the path carries pending light terms, i.e. its shading kernel will want its fa* words or line 1 of the record -- the traversal
// kernel hands the bit on in the list entry (kEntPending), so that the loads of line 1 are issued with those of line 0
constexpr uint32_t kRayExt = 0u, kRayShadow = 1u, kRayProbe = 2u, kRayNone = 3u;
constexpr uint32_t kSlotMask = 0x1fffffffu, kQPending = 1u << 29;
constexpr uint32_t kNullEntry = 0xffffffffu;  // unused queue / list entry
constexpr uint32_t kEntPending = 1u << 30;    // in ListEnt::slot
// A camera sample has no record until its first vertex is shaded: its ray is in the ray arrays, the state of its RNG in
// rng0[], its film slot = its index in the batch = Ctl::gen_ring's first sample + (slot - first slot); beta = 1, L = 0.  k_generate
// writes 60 B per sample instead of 180, and the shading kernels do not fetch 128 B that mostly say "one" and "zero"
// (the lists keep the queue's order, so a wave's camera samples sit in nearly consecutive slots: these reads coalesce).
constexpr uint32_t kEntFresh = 1u << 29;

// ---- vertex classes (round 4).  Every primitive belongs to a class = (smallest shading-kernel instance that covers its
// material, kind of hit record: mesh slot / sphere-rect / generic); class 0 = the extension ray escaped.  A leaf's class
// rides in bits 27-29 of its leaf_prim word (and so in the traversal's best_slot), the traversal kernel appends every
// finished extension ray to the list of its class, and one shading kernel per class runs on waves that are class-pure
// over the whole launch.
constexpr int kMaxCls = 8;  // (the traversal kernel keeps a list cursor per class in SGPRs)
constexpr uint32_t kClsShift = 27;
constexpr uint32_t kIdxMask = (1u << kClsShift) - 1u;  // leaf slot / primitive index part of a leaf_prim or hit word
// hit word of a list entry: class bits, kLeafOther, and the LEAF SLOT of a mesh hit or the PRIMITIVE INDEX of a sphere /
// rect hit -- what the record of that kind of hit is rebuilt from (geom.h: tri_record_slot / prim_intersects)
constexpr int kKindAny = 0, kKindMesh = 1, kKindOther = 2, kKindNone = 3;
struct ClsDesc {
    uint8_t variant;  // index into kFeatVariants
    uint8_t kind;     // kKind*
};
struct alignas(8) ListEnt {
    uint32_t slot, hit;  // slot | kEntPending | kEntFresh; hit word of the extension ray.  (The shadow / probe results of a
                         // path with pending terms are read at its slot by the shading kernel: 8 B per entry, not 16)
};
struct Lists {
    ListEnt* ent;      // one arena of `cap` entries: the lists of an iteration back to back, class c from Ctl::cls_base[..][c]
    uint32_t* fold[2];  // slots of fold-only paths, written by the shading kernels of iteration it for it + 1
    uint32_t cap;      // entries in the arena / in a fold list
    uint32_t n_cls;
};

struct DevStats {  // one shard = two 64-B lines; kStatShards shards, summed by the host
    unsigned long long paths, r1, r2, r3, vertices, nodes, tris, others;
    unsigned long long pad[12];  // diagnostics of the instrumented build
    unsigned long long tail_rays, tail_nodes, tail_tris, tail_others;  // k_tail's share (rt_stats.tail_*)
};
constexpr int kStatShards = 64;

// ------------------------------------------------------------ kernel control blocks (kernels.hip)
constexpr uint32_t kRing = 512;  // per-iteration counters live in a ring indexed by it % kRing

constexpr int kGroupCursorWord = 16;
struct Ctl {
    uint32_t n_active[kRing];
    uint32_t n_rays[kRing];
    uint32_t head[kRing];
    // written by k_plan for the k_generate that follows it
    uint32_t gen_count, gen_first, gen_slot, gen_q;
    // {gen_first, gen_slot} of an iteration (ring of 4) for the kernels that shade its camera samples -- the light kernel
    // of iteration i still runs when k_plan(i + 1) rewrites the four words above
    uint32_t gen_ring[4][2];
    // (RT_XCD_QUEUE) one queue head per XCD and iteration (ring of 4), each on its own 128-B line
    uint32_t xhead[4][8][32];
    // lengths of the class lists / the fold list of an iteration (ring of 4), each counter on its own 128-B line
    uint32_t cls_count[4][kMaxCls][32];  // [..][0]: the length; [..][kGroupCursorWord]: the class kernel's group cursor
    uint32_t cls_base[4][kMaxCls];  // where the list of a class starts in Lists::ent (k_classify_scan)
    uint32_t fold_count[4][32];
};

// The batch being rendered, shared by both lanes.
struct BatchCtl {
    unsigned long long next;  // next camera sample (path index inside the batch) to generate
};

// Host-visible copy of the per-iteration counters (pinned, mapped memory).  k_trace(it) publishes
// {n_active[it], n_rays[it]} when it STARTS; the host sizes the grids of iteration it+2 from it
// (counts never grow) and stops launching once a published n_active is zero -- no stream sync.
struct MirrorEntry {
    uint32_t n_active, n_rays, seq, remaining;  // remaining: camera samples of the batch not yet generated (saturated)
};

struct ChunkDesc {
    uint32_t n_pixels;     // pixels in this batch (PB)
    uint32_t n_samples;    // samples per pixel in this batch
    uint32_t pixel_base;   // offset into pix_list
    uint32_t sample_base;  // first sample index
    uint32_t width, height;
    uint64_t seed;
};

struct TraceTune {
    int refill_lanes;  // refill when at least this many lanes are idle (<= 64)
    int node_bias;     // a node step runs when lanes_at_nodes * node_bias >= lanes_at_leaves * 4 (4 = plain majority)
    int unused;
    int reserve;       // queue entries a wave reserves per atomic (refills are served from the reservation)
};

}  // namespace rtc

namespace rtd {
using namespace rtc;
}
namespace rtd32 {
using namespace rtc;
}

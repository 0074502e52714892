/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Bit-level PARITY UNPINNED; pinned statistically to the reference's
 * examples/cornell_statue.png and to independent lobe tables (see oracle.cpp header).
 * C entry points of liboracle.so, loaded by tests/, smoke() and bench.py's
 * cpu_baseline leg through ctypes.  Scenes arrive as the same flattened POD
 * arrays the product's C ABI takes (include/rt_abi.h).                        */
#ifndef ORACLE_API_H
#define ORACLE_API_H

#include "../include/rt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ORACLE_TRAVERSAL_EXHAUSTIVE = 0, /* reference-shaped: hittable.rs:591-634, no pruning */
    ORACLE_TRAVERSAL_ORDERED = 1,    /* same answer, tmax-shrinking stack traversal       */
    ORACLE_TRAVERSAL_BRUTE = 2       /* loop over every primitive (ground truth)          */
};

typedef struct oracle_scene oracle_scene;

typedef struct oracle_hit_record {
    int32_t hit;
    int32_t front;
    double t;
    double uv[2];
    double p[3], n[3], sh_n[3], sh_dpdu[3], sh_dpdv[3];
} oracle_hit_record;

int oracle_scene_create(const rt_scene_desc* desc, oracle_scene** out);
int oracle_scene_destroy(oracle_scene* s);
int oracle_render(const oracle_scene* s, const rt_camera* cam, const rt_render_cfg* cfg, int traversal_mode,
                  int n_threads, double* rgb_sum, uint32_t* n, rt_stats* stats);
int oracle_intersect_batch(const oracle_scene* s, const rt_ray* rays, uint64_t n, int traversal_mode,
                           rt_hit* hits);
int oracle_prim_intersect(const oracle_scene* s, int32_t prim, const rt_ray* ray, oracle_hit_record* out);
int oracle_sample(const oracle_scene* s, const rt_camera* cam, const rt_render_cfg* cfg, uint32_t px, uint32_t py,
                  uint32_t sample, int traversal_mode, double* rgb, rt_stats* stats);

int64_t oracle_sample_rays(const oracle_scene* s, const rt_camera* cam, const rt_render_cfg* cfg, uint32_t px,
                           uint32_t py, uint32_t sample, int traversal_mode, rt_ray* rays, rt_hit* hits,
                           uint64_t capacity);

double oracle_fr_dielectric(double cos_theta_i, double eta_i, double eta_t);
void oracle_fr_conductor(double cos_theta_i, const double* eta, const double* k, double* out);
double oracle_power_heuristic(int nf, double f_pdf, int ng, double g_pdf);
double oracle_tr_d(double ax, double ay, const double* wh);
double oracle_tr_lambda(double ax, double ay, const double* w);
double oracle_tr_g(double ax, double ay, const double* wo, const double* wi);
double oracle_tr_pdf(double ax, double ay, const double* wo, const double* wh);
void oracle_tr_sample_wh(double ax, double ay, const double* wo, double u0, double u1, double* out);
double oracle_tr_roughness_to_alpha(double roughness);
void oracle_concentric_sample_disk(double u0, double u1, double* out);
void oracle_rand_cosine_dir(double r1, double r2, double* out);
void oracle_rng_draws(uint64_t seed, uint64_t pixel, uint64_t sample, uint32_t n, double* out);
int oracle_box_intersects(const double* bmin, const double* bmax, const rt_ray* ray);
int oracle_refract(const double* v, const double* n, double eta, double* out);
void oracle_lambert_f_pdf(const double* color, const double* wo, const double* wi, double* f, double* pdf);
void oracle_microfacet_f_pdf(double ax, double ay, const double* eta, const double* k, const double* wo,
                             const double* wi, double* f, double* pdf);
void oracle_micro_trans(double ax, double ay, double eta, const double* color, const double* wo, const double* wi,
                        double u0, double u1, double* f, double* pdf, double* s_wi, double* s_f, double* s_pdf);
/* One BxDF lobe in the local shading frame.  kind: 0 LambertianReflection, 1 MicrofacetReflection
 * (Trowbridge-Reitz, visible-area sampling), 2 FresnelSpecular, 3 SpecularReflection,
 * 4 MicrofacetTransmission; fresnel: 0 FresnelDielectric{eta_i, eta_t}, 1 FresnelConductor{eta, k},
 * 2 FresnelNoOp.  alpha_x / alpha_y are the values AFTER make_trowbridge_reitz's max(1e-3) clamp.   */
typedef struct oracle_lobe {
    int32_t kind, fresnel;
    double color[3], t[3];
    double eta_i, eta_t;
    double eta[3], k[3];
    double alpha_x, alpha_y;
    double eta_a, eta_b;
} oracle_lobe;
void oracle_lobe_eval(const oracle_lobe* lobe, const double* wo, const double* wi, double* f, double* pdf);
void oracle_lobe_sample(const oracle_lobe* lobe, const double* wo, double u0, double u1, const uint64_t* rng_key,
                        double* f, double* wi, double* pdf);
int oracle_env(const oracle_scene* s, int what, const double* in, double* out);
double oracle_prim_area(const oracle_scene* s, int32_t prim);
double oracle_prim_pdf(const oracle_scene* s, int32_t prim, const double* p, const double* dir);
void oracle_texture_value(const oracle_scene* s, uint32_t tex, double u, double v, double* out);
void oracle_detmath(int fn, const double* x, const double* y, uint64_t n, double* out);
void oracle_resolve_rgb8(const double* rgb_sum, const uint32_t* n, uint64_t npix, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif

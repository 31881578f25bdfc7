/* ORACLE — TEST INFRASTRUCTURE ONLY.  C ABI of the CPU restatement (liboracle.so), used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker.  It shares the POD
 * records of include/rt_api.h so the same flat scene feeds both the oracle and the HIP library.
 * PARITY UNPINNED w.r.t. the original binary; pinned by tests/golden/halton_known_answers.json. */
#ifndef ORACLE_API_H
#define ORACLE_API_H
#include "../include/rt_api.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

/* LIST: Sphere::Intersect over the whole list -- the contract.  BVH: the reference's BvhNode (its slab test loses grazing hits
   the list finds, ~2 in 10^7 paths on 10^4 spheres).  PADDED_LIST: the list scan's result, found through a tree with
   provably conservative (padded, binary64) boxes -- what the full-size differentials use (rt_oracle.h PaddedListTree). */
enum { ORC_ACCEL_LIST = 0, ORC_ACCEL_BVH = 1, ORC_ACCEL_PADDED_LIST = 2 };

int orc_create(orc_ctx** out);
void orc_destroy(orc_ctx* ctx);
const char* orc_last_error(void);

/* Scene generators (InitScene/InitCamera restated): name in {"cover","three","grid10k"}.
 * cap = capacity of the spheres/materials arrays; *n receives the count (even when cap is too
 * small, in which case nothing is written and 1 is returned). */
int orc_build_scene(const char* name, uint64_t seed, float aspect, float aperture_override, uint32_t cap,
                    rt_sphere* spheres, rt_material* materials, uint32_t* n, rt_camera* camera, rt_light* sun,
                    rt_material* sky, float* exposure_scale);

int orc_scene_upload(orc_ctx* ctx, const rt_sphere* spheres, const rt_material* materials, uint32_t n,
                     const rt_camera* camera, const rt_light* lights, uint32_t n_lights, const rt_material* sky, float exposure_scale);
int orc_render(orc_ctx* ctx, uint32_t W, uint32_t H, rt_rowset rs, uint32_t s0, uint32_t s1, uint32_t max_depth,
               uint64_t seed, int accel, int threads, rt_stats* out_stats);
/* N3: drive material draws from the reference's per-material Halton counters (serial renders only) */
void orc_use_reference_halton_counters(int on);
/* N3: sampler variants behind flags (RT_SAMPLER_*, include/rt_api.h); process-wide, 0 = the reference's mappings */
void orc_set_sampler(uint32_t flags);
/* CPU-only diagnostic: GetHitColor in the reference's nesting L = (E+S) + a*(...) (spheres-app.cpp:249-251) instead of
 * the forward radiance += throughput*(E+S) form that the path's contract (and the HIP kernel) uses */
void orc_use_nested_radiance(int on);
void orc_use_reference_bvh_tie_rule(int on);  /* default 1: ORC_ACCEL_BVH gives exact ties to the right child (the reference's rule); 0 (diagnostic): to the lower list index, like the list scan */
int orc_clear(orc_ctx* ctx);
int orc_resolve(orc_ctx* ctx, uint32_t n_samples);
int orc_download(orc_ctx* ctx, float* hdr_rgb, uint8_t* ldr_rgb);
uint32_t orc_rowset_local_rows(rt_rowset rs);
uint32_t orc_rowset_global_row(rt_rowset rs, uint32_t local_row);

/* unit-level functions (one per restated reference function) */
float orc_halton(uint64_t index, uint32_t base);
void orc_halton_disk(uint64_t index, uint32_t b1, uint32_t b2, float out[2]);
void orc_halton_hemisphere(uint64_t index, uint32_t b1, uint32_t b2, float out[3]);
int orc_unit_halton(const uint32_t* index, uint32_t base, uint32_t n, float* out);
int orc_unit_math(uint32_t op, const float* x, const float* y, uint32_t n, float* out);
void orc_camera_make(const float origin[3], const float look_at[3], float vfov, float aspect, float focal, float aperture,
                     rt_camera* out);
int orc_unit_primary_rays(orc_ctx* ctx, uint32_t W, uint32_t H, const uint32_t* ijs, uint32_t n, float* out_rays);
int orc_unit_closest_hit(orc_ctx* ctx, const float* rays, uint32_t n, int accel, float* out_hits);
int orc_unit_trace(orc_ctx* ctx, uint32_t W, uint32_t H, const uint32_t* ijs, uint32_t n, uint32_t max_depth, uint64_t seed,
                   int accel, float* out_rgb, uint32_t* out_traversals);
/* Diagnostic: one sample's path with every accelerator query recorded -- 8 floats per query: origin xyz, direction xyz, kind
   (0 closest hit, 1 sun occlusion), result (closest: t or -1 for a miss; occlusion: 1 occluded, 0 visible).  *n_queries = queries
   made (only the first `cap` are stored). */
int orc_unit_trace_path(orc_ctx* ctx, uint32_t W, uint32_t H, uint32_t i, uint32_t j, uint32_t s, uint32_t max_depth, uint64_t seed,
                        int accel, float* out_queries, uint32_t cap, uint32_t* n_queries, float out_rgb[3]);
float orc_fresnel_term(float cos_incident, float ior);
void orc_refract(const float incident[3], const float normal[3], float eta, float out[3]);
void orc_reflect(const float incident[3], const float normal[3], float out[3]);
uint32_t orc_color_pack(float r, float g, float b, float a);
void orc_color_load(uint32_t argb, float out[4]);
/* scatter one hit: material record + incoming ray + hit (pos, normal, uv) + 3 uniforms to consume in order.
 * returns scattered flag; out: attenuation[3], dir[3], n_draws */
int orc_unit_scatter(const rt_material* m, const float ray_dir[3], const float pos[3], const float normal[3], const float uv[2],
                     const float draws[3], float out_atten[3], float out_dir[3], uint32_t* n_draws);
/* Material::Emit + Material::Shade with an unoccluded sun (light.cpp:21-40) for one hit */
void orc_unit_emit_shade(const rt_material* m, const rt_light* sun, const float view_origin[3], const float pos[3],
                         const float normal[3], const float uv[2], float out_local[3]);
void orc_xoshiro_seed(uint64_t seed, uint32_t pixel_id, uint32_t sample, uint32_t out_state[4]);
void orc_xoshiro_draws(uint64_t seed, uint32_t pixel_id, uint32_t sample, uint32_t n, float* out);
void orc_tonemap(const float hdr_rgb[3], uint32_t n_samples, uint8_t out_rgb[3]);

#ifdef __cplusplus
}
#endif
#endif

/*
 * rt_api.h — C-ABI drop-in boundary for the render-loop hot path of
 * SakibSaikia/CPURayTracer (src/spheres), MI355X-native implementation.
 *
 * The reference has no FFI: the replaceable statements are the two Parallel-STL
 * algorithms inside SpheresApp::DrawBitmap
 *     std::for_each(par, rays, hdr[id] += GetHitColor(ray,0) * exposure)   spheres-app.cpp:177-184
 *     std::transform(par, hdr -> ldr: /n, ACES, gamma, XMStoreColor)       spheres-app.cpp:196-214
 * plus the serial ray generation feeding them (GenerateRays, spheres-app.cpp:132-161).
 * This header is what a binding for that path binds: plain C types, caller-owned
 * host buffers, device memory owned by the handle.  Every entry point returns an
 * int status (RT_OK == 0); rt_last_error() returns the message for the calling
 * thread.  Calls on one handle are serialised by the caller; one handle per device.
 *
 * The library behind this header is librt_hip.so (cpuraytracer_amd/csrc/): hand
 * written HIP for gfx950.  There is NO CPU fallback: rt_create fails loudly when
 * no HIP device is present.  The CPU restatement used for parity checks lives in
 * oracle/ behind oracle/oracle_api.h and shares only the POD records below.
 */
#ifndef RT_API_H
#define RT_API_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_API_VERSION 2 /* 2: rt_scene_upload takes the LIST of lights (spheres-app.h:38 m_lights), not one sun */

enum rt_status {
    RT_OK = 0,
    RT_ERR_NO_DEVICE = 1,     /* no HIP device / ordinal out of range              */
    RT_ERR_INVALID_ARG = 2,   /* null pointer, zero size, bad range                */
    RT_ERR_NO_SCENE = 3,      /* rt_render before rt_scene_upload                  */
    RT_ERR_HIP = 4,           /* a HIP runtime call failed (see rt_last_error)     */
    RT_ERR_OUT_OF_MEMORY = 5,
    RT_ERR_SEQUENCE = 6       /* sample range does not continue the accumulation   */
};

/* ------------------------------------------------------------------ records */

/* Sphere — common-lib/ray-tracing.h:53-57 (center, radius); material i belongs to sphere i. */
typedef struct rt_sphere {
    float cx, cy, cz, r;
} rt_sphere; /* 16 B */

/* Material kinds — common-lib/material.h:21,37,54,70 */
enum rt_material_type {
    RT_MAT_DIELECTRIC_OPAQUE = 0,
    RT_MAT_METAL = 1,
    RT_MAT_DIELECTRIC_TRANSPARENT = 2,
    RT_MAT_EMISSIVE = 3
};

/* Texture kinds — common-lib/texture.h:12,22 */
enum rt_texture_type {
    RT_TEX_CONST = 0,
    RT_TEX_CHECKER = 1
};

/* Material + its (single) texture, flattened.  Colours are the values the
 * reference holds after XMLoadColor(XMCOLOR) (texture.cpp:5,16-17): byte * (1/255). */
typedef struct rt_material {
    uint32_t type;       /* rt_material_type                                         */
    uint32_t tex_type;   /* rt_texture_type (opaque: albedo, metal: reflectance)     */
    float smoothness;    /* m_smoothness (replicated scalar)                         */
    float ior;           /* DielectricTransparent::m_ior                             */
    float tiling;        /* CheckerTexture::m_tilingScale                            */
    float rgb0[3];       /* ConstTexture colour / checker colour 0                   */
    float rgb1[3];       /* checker colour 1                                         */
    float luminance;     /* Emissive::m_luminance                                    */
} rt_material; /* 48 B */

/* Camera members after Camera::Camera — common-lib/camera.h:14-19, camera.cpp:3-28 */
typedef struct rt_camera {
    float origin[4];              /* m_origin            */
    float x[4];                   /* m_x = halfWidth*u   */
    float y[4];                   /* m_y = halfHeight*v  */
    float origin_image_plane[4];  /* m_originImagePlane  */
    float aperture;               /* m_aperture          */
    float focal_length;           /* m_focalLength       */
} rt_camera; /* 72 B */

/* DirectionalLight members — common-lib/light.h:18-21, light.cpp:4-9 */
typedef struct rt_light {
    float direction[3];  /* normalised */
    float color[3];      /* XMLoadColor of the ctor colour */
    float luminance;
} rt_light; /* 28 B */

/* Rows of the W x H image this call renders.  The row range
 * [first_row, first_row+num_rows) is cut into blocks of block_rows rows; the
 * call owns the blocks b with b % nshards == shard (cyclic: cost is strongly
 * row dependent).  Owned rows, in increasing order, are the "local rows" of the
 * HDR/LDR strip.  {0, H, H, 0, 1} is the whole image. */
typedef struct rt_rowset {
    uint32_t first_row;
    uint32_t num_rows;
    uint32_t block_rows;
    uint32_t shard;
    uint32_t nshards;
} rt_rowset;

/* Counters a render call returns; traversals must equal the oracle's count. */
typedef struct rt_stats {
    uint64_t samples;      /* (pixel, s) paths traced                               */
    uint64_t traversals;   /* ray-vs-whole-list scans: closest-hit + shadow         */
    uint64_t segments;     /* closest-hit scans only (= ray segments)               */
    double ms_render;      /* HIP-event time of the trace kernel(s)                 */
    double ms_accumulate;  /* HIP-event time of the ordered accumulate kernel       */
    double ms_resolve;     /* HIP-event time of the last rt_resolve                 */
    uint32_t local_rows;   /* rows in this shard's strip                            */
    uint32_t passes;       /* sample-range passes the call was split into           */
} rt_stats;

typedef struct rt_ctx rt_ctx;

/* ------------------------------------------------------------- life cycle */

const char* rt_last_error(void);
int rt_api_version(void);

/* One context per device ordinal.  Fails with RT_ERR_NO_DEVICE when there is no GPU. */
int rt_create(int device_ordinal, rt_ctx** out);
void rt_destroy(rt_ctx* ctx);

/* Launch on a caller-provided hipStream_t (e.g. torch's current stream); NULL restores
 * the context's own stream. */
int rt_set_stream(rt_ctx* ctx, void* hip_stream);

/* Upper bound for the per-sample workspace in bytes (default: min(64 GiB, half of the
 * device's free memory) — sized for 288 GB of HBM, so BASELINE configs 3 and 5 are one
 * pass on one GPU; env RT_WORKSPACE_GIB overrides).  A render whose W*rows*(s1-s0)*12 bytes
 * exceed it is split into sample-range passes that run back to back on the stream. */
int rt_set_workspace_limit(rt_ctx* ctx, uint64_t bytes);

/* ------------------------------------------------------------------ scene */

/* Replaces SpheresApp::InitScene's products (spheres-app.cpp:51-130): n spheres with
 * their materials, the camera (InitCamera, :35-49), the sun, the sky Emissive
 * material and exposureAdjustment = 2^m_exposure (:174). */
/* Limit: the clustered scan table (groups of four, padded) must stay below 65,536 entries
 * — about 65,000 spheres — because work lists, the shadow index and the closest-hit keys
 * carry entry ids in 16 bits; larger scenes are rejected here with RT_ERR_INVALID_ARG. */
/* lights: SpheresApp::m_lights (spheres-app.h:38, filled at spheres-app.cpp:129) as flat records, in list order -- the order
 * Material::Shade adds their contributions in (material.cpp:4-13); 0 <= n_lights <= RT_MAX_LIGHTS (lights may be NULL when
 * n_lights == 0: Shade then returns zero and no shadow ray is cast).  Every light answers its any-hit shadow query through
 * an exact footprint index of its own; a light below the horizon of a surface contributes nDotL = 0 exactly as in
 * the reference.  One light (the reference's scene) runs the single-light kernel path unchanged. */
#define RT_MAX_LIGHTS 8
int rt_scene_upload(rt_ctx* ctx, const rt_sphere* spheres, const rt_material* materials, uint32_t n,
                    const rt_camera* camera, const rt_light* lights, uint32_t n_lights, const rt_material* sky,
                    float exposure_scale);

/* ----------------------------------------------------------------- render */

/* Replaces GenerateRays + the for_each(par) trace loop for sample indices s in
 * [s0, s1) (1-based like m_sampleCount, spheres-app.cpp:168) over the rows in rs.
 * hdr[pixel] += GetHitColor(ray(i,j,s), 0) * exposure, added in increasing s
 * (spheres-app.cpp:182-183).  The HDR strip persists in the context: a call with
 * s0 == 1 (or after rt_clear) starts a new accumulation, a call whose s0 equals
 * the previous s1 continues it (progressive refinement, app.h:26 + spheres-app.h:42).
 * Material random draws come from a per-(pixel,s) xoshiro128** stream seeded from
 * (seed, global pixel id, s) — see DESIGN.md "RNG contract".  With out_stats == NULL the call only enqueues work
 * on the context's stream (no host wait); with out_stats it waits for the kernels to read their timers.
 * A call that fails part-way voids the accumulation: the next call must start again at s0 == 1. */
int rt_render(rt_ctx* ctx, uint32_t W, uint32_t H, rt_rowset rs, uint32_t s0, uint32_t s1,
              uint32_t max_depth, uint64_t seed, rt_stats* out_stats);

/* Sampler variants.  Default 0 = the reference's own mappings: HaltonSampleHemisphere is uniform
 * in solid angle (quasi-random.cpp:36-50: z = u1, r = sqrt(1 - u1^2); NOT cosine weighted, and the caller applies
 * no pdf) and HaltonSampleDisk takes r = u (quasi-random.cpp:52-61: NOT sqrt'ed, centre-weighted lens).  The flags
 * select the textbook mappings instead, with the same draws in the same order:
 *   RT_SAMPLER_COSINE_HEMISPHERE  r = sqrt(u1), z = sqrt(1 - u1), phi = 2 pi u2   (DielectricOpaque diffuse bounce)
 *   RT_SAMPLER_SQRT_DISK          r = sqrt(u)                                     (lens offsets, GenerateRays :152)
 * Takes effect at the next accumulation (rt_render with s0 == 1); changing it voids a running one. */
#define RT_SAMPLER_COSINE_HEMISPHERE 1u
#define RT_SAMPLER_SQRT_DISK 2u
int rt_set_sampler(rt_ctx* ctx, uint32_t flags);

/* Frame pipelining for progressive use — the reference's only mode: OnRender adds ONE sample per pixel per frame
 * (spheres-app.cpp:163-184, app.cpp:56-76).  A 1-spp frame is 0.13 ms of work followed by the tail of its few 51-segment
 * paths.  With depth > 0, an rt_render call that passes out_stats == NULL returns with up to `depth` of the newest calls'
 * sample planes still in flight: its kernel ends when nothing is left to start and carries the unfinished paths into the
 * next call's kernel.  Planes are added to the HDR strip strictly in sample order as they complete, so after a flush the
 * strip equals the one-shot render bit for bit.  While frames are in flight rt_resolve(ctx, 0) divides by the number of
 * samples actually in the strip (rt_committed_samples), and rt_download / rt_copy_to_device hand out that strip.
 * rt_synchronize, a call with statistics and rt_set_frame_pipelining itself settle everything first.  depth 0 = off.
 * SCOPE: the carrying kernel exists for scenes whose tables fit LDS with the matrix-core filter (up to 128 groups of four
 * spheres, i.e. about 500 spheres: the reference's cover scene and everything smaller).  Larger scenes -- the cell-grid and
 * bounds-hierarchy scans (10,000-sphere class) -- and contexts created under non-default launch knobs render every call
 * UNPIPELINED: same results, each call runs its own paths to their end before the next one starts. */
int rt_set_frame_pipelining(rt_ctx* ctx, uint32_t depth);
/* Frame batching for progressive use: a 1-spp frame is less work than the three launches it takes.  With frames > 1, rt_render
 * calls that pass out_stats == NULL and continue each other (same image, row set, depth and seed; s0 == the previous s1) are
 * only RECORDED until `frames` sample planes are pending; then ONE launch renders them all, and the planes are added to the
 * HDR strip in sample order as always -- the strip equals what the separate calls would have produced, bit for bit.  Until
 * then rt_resolve / rt_download / rt_copy_to_device hand out the samples committed so far (rt_committed_samples; when nothing
 * is committed yet they render the pending frames first).  rt_synchronize, a call with statistics, a call that does not
 * continue the batch, rt_scene_upload, rt_set_stream and rt_set_frame_batch itself render what is pending first; errors of
 * deferred calls surface there.  Not combined with rt_set_frame_pipelining (depth > 0 takes precedence).  frames == 1: off. */
int rt_set_frame_batch(rt_ctx* ctx, uint32_t frames);
/* Render-AHEAD for progressive use (round 4): with frames > 1, an rt_render call that passes out_stats == NULL traces, with its ONE
 * launch, its own sample planes AND the planes of the next calls (`frames` planes in all), but adds only its own to the HDR strip;
 * the calls that continue it (same image, row set, depth and seed; s0 == the previous s1) find their planes traced and only add
 * them, in sample order -- one small kernel.  The strip always holds EXACTLY the samples of the calls made (display lag 0: what
 * rt_resolve / rt_download hand out after call f is the one-shot render of samples 1..f, bit for bit), the launch and its tail are
 * paid once per `frames` calls, and the first call of each group takes the group's time.  A call that does not continue, a call
 * with statistics, rt_scene_upload, rt_clear, rt_set_stream drop what was traced ahead (nothing of it ever reached the strip).
 * Why not one frame per launch at full rate: a path is up to 51 dependent scans of ~10 us, five 1200x800 1-spp frames' worth of
 * bulk-rate work, so a frame's last sample cannot exist before ~5 later frames have been started (interactive mode).
 * Takes precedence below rt_set_frame_batch and rt_set_frame_pipelining (either of them on: ignored).  frames == 1: off. */
int rt_set_frame_lookahead(rt_ctx* ctx, uint32_t frames);
/* Samples per pixel in the HDR strip right now (waits for the stream). */
int rt_committed_samples(rt_ctx* ctx, uint32_t* out);

/* Forget the accumulated HDR strip and sample count. */
int rt_clear(rt_ctx* ctx);

/* Replaces the transform(par) tonemap (spheres-app.cpp:186-214): hdr / n_samples,
 * ACES fit, gamma 1/2.2, XMStoreColor.  n_samples == 0 uses the accumulated count. */
int rt_resolve(rt_ctx* ctx, uint32_t n_samples);
/* HIP-event time of the last rt_resolve kernel, milliseconds. */
double rt_last_resolve_ms(rt_ctx* ctx);

/* Copy the strip to host.  hdr_rgb: W*local_rows*3 floats; ldr_rgb: W*local_rows*3
 * bytes (R,G,B of the XMCOLOR).  Either pointer may be NULL. */
int rt_download(rt_ctx* ctx, float* hdr_rgb, uint8_t* ldr_rgb);

/* Same, device to device into caller-owned device memory (e.g. a torch tensor that
 * feeds the RCCL gather).  Asynchronous on the context's stream. */
int rt_copy_to_device(rt_ctx* ctx, void* dev_hdr_rgb, void* dev_ldr_rgb);

/* Block until everything queued on the context's stream has finished. */
int rt_synchronize(rt_ctx* ctx);

/* Global row index of local row lr under rs (pure host arithmetic; no GPU). */
uint32_t rt_rowset_local_rows(rt_rowset rs);
uint32_t rt_rowset_global_row(rt_rowset rs, uint32_t local_row);

/* --------------------------------------------------- unit-level entry points
 * Batched, device-evaluated pieces of the path, so that each reference function
 * has a GPU-vs-oracle known-answer test (tests/test_gpu_parity.py).  All buffers
 * are host memory; n elements. */

/* Random::HaltonSample — quasi-random.cpp:3-16 */
int rt_unit_halton(rt_ctx* ctx, const uint32_t* index, uint32_t base, uint32_t n, float* out);
/* op 0: sin, 1: cos, 2: pow(x,y), 3: tan — the shared elementary-function contract; 4: the refined reciprocal and
 * 5: the guarded Markstein quotient x/y of the hit processing (both must equal IEEE division); 6: plain x/y on the device;
 * 7: the path's square root (guarded fast form), 8: its fast form alone (x in [2^-80, inf)), 9: the compiler's sqrtf on the
 * device -- all three must equal the IEEE square root */
int rt_unit_math(rt_ctx* ctx, uint32_t op, const float* x, const float* y, uint32_t n, float* out);
/* GenerateRays for chosen pixels: (i,j,s) -> origin xyz, direction xyz (6 floats each) */
int rt_unit_primary_rays(rt_ctx* ctx, uint32_t W, uint32_t H, const uint32_t* ijs /*3 per ray*/,
                         uint32_t n, float* out_rays);
/* Closest hit over the uploaded scene: rays (6 floats) -> t, index (as float bits), pos xyz,
 * normal xyz, uv (10 floats each; index < 0 == miss) */
int rt_unit_closest_hit(rt_ctx* ctx, const float* rays, uint32_t n, float* out_hits);
/* Per-sample radiance*exposure for chosen (i,j,s): 3 floats each + traversal count */
int rt_unit_trace(rt_ctx* ctx, uint32_t W, uint32_t H, const uint32_t* ijs, uint32_t n,
                  uint32_t max_depth, uint64_t seed, float* out_rgb, uint32_t* out_traversals);

/* Camera::GetRay (camera.cpp:30-48) for (uv.x, uv.y, offset.x, offset.y) quadruples -> origin xyz, direction xyz */
int rt_unit_camera_rays(rt_ctx* ctx, const rt_camera* camera, const float* uv_offset, uint32_t n, float* out_rays);
/* Material::Scatter (material.cpp:20-164) + Emit + unoccluded DirectionalLight::Shade (light.cpp:21-40) for one
 * material record.  in: ray direction 3, hit pos 3, hit normal 3, three uniforms consumed in call order (12 floats);
 * out: scattered flag, attenuation 3, scattered direction 3, draws consumed, Emit+Shade 3 (11 floats). */
int rt_unit_scatter(rt_ctx* ctx, const rt_material* material, const rt_light* sun, const float view_origin[3],
                    const float* in, uint32_t n, float* out);  /* uses the context's sampler flags */
/* Host-only view of the clustered storage rt_scene_upload builds for the scan (groups of four spheres with a
 * conservative bounding sphere each).  orig: 4 entries per group, 0xffffffff = padding;
 * bounds: Cx, Cy, Cz, |C|^2 - Rf^2 per group.  cap_groups == 0 only queries *n_groups.  Needs no GPU. */
int rt_unit_layout(const rt_sphere* spheres, uint32_t n, uint32_t cap_groups, uint32_t* n_groups, uint32_t* orig, float* bounds);
/* ... and which closest-hit scan that storage is laid out for: out[0] = 0 flat matrix-core filter (<= 128 groups), 1 cell grid,
 * 2 bounds hierarchy; out[1], out[2] = grid cells along its two axes (0 otherwise); out[3] = big spheres tested for every ray;
 * out[4] = levels of bounds.  Needs no GPU. */
int rt_unit_layout_info(const rt_sphere* spheres, uint32_t n, uint32_t out[5]);
/* The cell-grid scan's walk as SHIPPED (csrc/rt_scan.h grid_segment_slope + grid_slab_rows, compiled for the host from the same
 * source the kernels use): for n segments (su, sv, eu, ev, D: 5 floats each, grid coordinates) and slabs iu, the rows [r0, r1] of
 * slab iu the walk visits (r0 > r1: none) and the entry parameter sEnter.  out_rows: 2 ints per query.  Needs no GPU. */
int rt_unit_grid_rows(const float* segments, const int32_t* iu, uint32_t n, int32_t nv, int32_t* out_rows, float* out_s_enter);
/* The resolve of spheres-app.cpp:196-214 for given HDR triples -> R,G,B bytes */
int rt_unit_tonemap(rt_ctx* ctx, const float* hdr_rgb, uint32_t n, uint32_t n_samples, uint8_t* out_rgb);

#ifdef __cplusplus
}
#endif
#endif /* RT_API_H */

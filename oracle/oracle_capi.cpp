// ORACLE — TEST INFRASTRUCTURE ONLY.  C ABI over rt_oracle.{h,cpp}; see oracle_api.h.
#include "oracle_api.h"

#include <cstring>
#include <string>

#include "rt_oracle.h"

using namespace orc;

struct orc_ctx {
    SpheresApp app;
};

static thread_local std::string g_err;
static int Fail(const char* msg) {
    g_err = msg;
    return 1;
}

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }

int orc_create(orc_ctx** out) {
    if (!out) return Fail("orc_create: null out");
    *out = new orc_ctx();
    return 0;
}
void orc_destroy(orc_ctx* ctx) { delete ctx; }

int orc_build_scene(const char* name, uint64_t seed, float aspect, float aperture_override, uint32_t cap, rt_sphere* spheres,
                    rt_material* materials, uint32_t* n, rt_camera* camera, rt_light* sun, rt_material* sky,
                    float* exposure_scale) {
    FlatScene fs;
    if (!BuildNamedScene(name, seed, aspect, aperture_override, fs)) return Fail("orc_build_scene: unknown scene name");
    if (n) *n = (uint32_t)fs.spheres.size();
    if (fs.spheres.size() > cap) return Fail("orc_build_scene: capacity too small");
    if (spheres) std::memcpy(spheres, fs.spheres.data(), fs.spheres.size() * sizeof(rt_sphere));
    if (materials) std::memcpy(materials, fs.materials.data(), fs.materials.size() * sizeof(rt_material));
    if (camera) *camera = fs.camera;
    if (sun) *sun = fs.sun;
    if (sky) *sky = fs.sky;
    if (exposure_scale) *exposure_scale = fs.exposureScale;
    return 0;
}

int orc_scene_upload(orc_ctx* ctx, const rt_sphere* spheres, const rt_material* materials, uint32_t n, const rt_camera* camera,
                     const rt_light* lights, uint32_t n_lights, const rt_material* sky, float exposure_scale) {
    if (!ctx || !spheres || !materials || !camera || (!lights && n_lights != 0) || !sky || n == 0) return Fail("orc_scene_upload: invalid argument");
    FlatScene fs;
    fs.spheres.assign(spheres, spheres + n);
    fs.materials.assign(materials, materials + n);
    fs.camera = *camera;
    fs.lightsGiven = true;
    fs.lights.assign(lights, lights + n_lights);
    if (n_lights) fs.sun = lights[0];
    fs.sky = *sky;
    fs.exposureScale = exposure_scale;
    ctx->app.LoadScene(fs, 1);
    return 0;
}

static SpheresApp::Accel AccelOf(int accel) {
    return accel == ORC_ACCEL_BVH ? SpheresApp::Accel::Bvh : (accel == ORC_ACCEL_PADDED_LIST ? SpheresApp::Accel::PaddedList : SpheresApp::Accel::List);
}
int orc_render(orc_ctx* ctx, uint32_t W, uint32_t H, rt_rowset rs, uint32_t s0, uint32_t s1, uint32_t max_depth, uint64_t seed,
               int accel, int threads, rt_stats* out_stats) {
    if (!ctx || W == 0 || H == 0 || s1 <= s0 || s0 == 0) return Fail("orc_render: invalid argument");
    if (!ctx->app.HasScene()) return Fail("orc_render: no scene");
    if (RowsetLocalRows(rs) == 0 || rs.first_row + rs.num_rows > H) return Fail("orc_render: bad row set");
    if (Random::ReferenceHaltonCounters() && threads > 1) return Fail("orc_render: the reference-counter sampler is serial only");
    if (s0 == 1) ctx->app.Clear();
    else if (ctx->app.SampleCount() + 1 != s0) return Fail("orc_render: sample range does not continue the accumulation");
    RenderCounters rc;
    ctx->app.Render(W, H, rs, s0, s1, max_depth, seed, AccelOf(accel),
                    threads, rc);
    if (out_stats) {
        std::memset(out_stats, 0, sizeof(*out_stats));
        out_stats->samples = rc.samples;
        out_stats->traversals = rc.traversals;
        out_stats->segments = rc.segments;
        out_stats->local_rows = ctx->app.StripRows();
        out_stats->passes = 1;
    }
    return 0;
}

void orc_use_reference_halton_counters(int on) { Random::UseReferenceHaltonCounters(on != 0); }
void orc_set_sampler(uint32_t flags) { Random::SetSamplerFlags(flags); }
void orc_use_nested_radiance(int on) { UseNestedRadiance(on != 0); }
void orc_use_reference_bvh_tie_rule(int on) { UseReferenceBvhTieRule(on != 0); }

int orc_clear(orc_ctx* ctx) {
    if (!ctx) return Fail("orc_clear: null ctx");
    ctx->app.Clear();
    return 0;
}

int orc_resolve(orc_ctx* ctx, uint32_t n_samples) {
    if (!ctx) return Fail("orc_resolve: null ctx");
    if (n_samples == 0 && ctx->app.SampleCount() == 0) return Fail("orc_resolve: nothing accumulated");
    ctx->app.Resolve(n_samples);
    return 0;
}

int orc_download(orc_ctx* ctx, float* hdr_rgb, uint8_t* ldr_rgb) {
    if (!ctx) return Fail("orc_download: null ctx");
    const auto& hdr = ctx->app.Hdr();
    const auto& ldr = ctx->app.Ldr();
    for (size_t k = 0; k < hdr.size(); ++k) {
        if (hdr_rgb) {
            hdr_rgb[3 * k + 0] = hdr[k].x;
            hdr_rgb[3 * k + 1] = hdr[k].y;
            hdr_rgb[3 * k + 2] = hdr[k].z;
        }
        if (ldr_rgb) {
            ldr_rgb[3 * k + 0] = (uint8_t)((ldr[k].c >> 16) & 0xff);
            ldr_rgb[3 * k + 1] = (uint8_t)((ldr[k].c >> 8) & 0xff);
            ldr_rgb[3 * k + 2] = (uint8_t)(ldr[k].c & 0xff);
        }
    }
    return 0;
}

uint32_t orc_rowset_local_rows(rt_rowset rs) { return RowsetLocalRows(rs); }
uint32_t orc_rowset_global_row(rt_rowset rs, uint32_t local_row) { return RowsetGlobalRow(rs, local_row); }

// ------------------------------------------------------------------------- units
float orc_halton(uint64_t index, uint32_t base) { return Random::HaltonSample(index, base); }
void orc_halton_disk(uint64_t index, uint32_t b1, uint32_t b2, float out[2]) {
    const XMFLOAT2 v = Random::HaltonSampleDisk(index, b1, b2);
    out[0] = v.x; out[1] = v.y;
}
void orc_halton_hemisphere(uint64_t index, uint32_t b1, uint32_t b2, float out[3]) {
    const XMFLOAT3 v = Random::HaltonSampleHemisphere(index, b1, b2);
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
}
int orc_unit_halton(const uint32_t* index, uint32_t base, uint32_t n, float* out) {
    for (uint32_t k = 0; k < n; ++k) out[k] = Random::HaltonSample(index[k], base);
    return 0;
}
int orc_unit_math(uint32_t op, const float* x, const float* y, uint32_t n, float* out) {
    for (uint32_t k = 0; k < n; ++k) {
        switch (op) {
            case 0: out[k] = rt_sinf(x[k]); break;
            case 1: out[k] = rt_cosf(x[k]); break;
            case 2: out[k] = rt_powf(x[k], y[k]); break;
            case 3: out[k] = rt_tanf(x[k]); break;
            default: return Fail("orc_unit_math: bad op");
        }
    }
    return 0;
}
void orc_camera_make(const float origin[3], const float look_at[3], float vfov, float aspect, float focal, float aperture,
                     rt_camera* out) {
    Camera cam(XMVectorSet(origin[0], origin[1], origin[2], 1.f), XMVectorSet(look_at[0], look_at[1], look_at[2], 1.f), vfov, aspect,
               focal, aperture);
    *out = cam.Flatten();
}
int orc_unit_primary_rays(orc_ctx* ctx, uint32_t W, uint32_t H, const uint32_t* ijs, uint32_t n, float* out_rays) {
    if (!ctx || !ctx->app.HasScene()) return Fail("orc_unit_primary_rays: no scene");
    for (uint32_t k = 0; k < n; ++k) {
        const Ray r = ctx->app.GeneratePrimaryRay(W, H, ijs[3 * k], ijs[3 * k + 1], ijs[3 * k + 2]);
        float* o = out_rays + 6 * k;
        o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
        o[3] = r.direction.x; o[4] = r.direction.y; o[5] = r.direction.z;
    }
    return 0;
}
int orc_unit_closest_hit(orc_ctx* ctx, const float* rays, uint32_t n, int accel, float* out_hits) {
    if (!ctx || !ctx->app.HasScene()) return Fail("orc_unit_closest_hit: no scene");
    // route through TraceSample's accel selection by a zero-sample trick: select accel directly
    for (uint32_t k = 0; k < n; ++k) {
        const float* r = rays + 6 * k;
        Ray ray{XMVectorSet(r[0], r[1], r[2], 1.f), XMVectorSet(r[3], r[4], r[5], 0.f)};
        float* o = out_hits + 10 * k;
        const auto hit = ctx->app.ClosestHitWith(ray, AccelOf(accel));
        if (hit) {
            const Payload& p = *hit;
            int32_t idx = p.index;
            o[0] = p.t.x;
            std::memcpy(&o[1], &idx, 4);
            o[2] = p.pos.x; o[3] = p.pos.y; o[4] = p.pos.z;
            o[5] = p.normal.x; o[6] = p.normal.y; o[7] = p.normal.z;
            o[8] = p.uv.x; o[9] = p.uv.y;
        } else {
            int32_t idx = -1;
            for (int q = 0; q < 10; ++q) o[q] = 0.f;
            std::memcpy(&o[1], &idx, 4);
        }
    }
    return 0;
}
int orc_unit_trace(orc_ctx* ctx, uint32_t W, uint32_t H, const uint32_t* ijs, uint32_t n, uint32_t max_depth, uint64_t seed, int accel,
                   float* out_rgb, uint32_t* out_traversals) {
    if (!ctx || !ctx->app.HasScene()) return Fail("orc_unit_trace: no scene");
    for (uint32_t k = 0; k < n; ++k) {
        uint32_t trav = 0;
        const XMVECTOR c = ctx->app.TraceSample(W, H, ijs[3 * k], ijs[3 * k + 1], ijs[3 * k + 2], max_depth, seed,
                                                AccelOf(accel), &trav);
        out_rgb[3 * k] = c.x; out_rgb[3 * k + 1] = c.y; out_rgb[3 * k + 2] = c.z;
        if (out_traversals) out_traversals[k] = trav;
    }
    return 0;
}
int orc_unit_trace_path(orc_ctx* ctx, uint32_t W, uint32_t H, uint32_t i, uint32_t j, uint32_t s, uint32_t max_depth, uint64_t seed, int accel,
                        float* out_queries, uint32_t cap, uint32_t* n_queries, float out_rgb[3]) {
    if (!ctx || !ctx->app.HasScene()) return Fail("orc_unit_trace_path: no scene");
    std::vector<float> rays;
    const XMVECTOR c = ctx->app.TraceSampleRecorded(W, H, i, j, s, max_depth, seed,
                                                    AccelOf(accel), rays);
    const uint32_t n = (uint32_t)(rays.size() / 8);
    if (n_queries) *n_queries = n;
    for (uint32_t k = 0; k < n && k < cap; ++k)
        for (int q = 0; q < 8; ++q) out_queries[8 * k + q] = rays[8 * (size_t)k + q];
    if (out_rgb) { out_rgb[0] = c.x; out_rgb[1] = c.y; out_rgb[2] = c.z; }
    return 0;
}
float orc_fresnel_term(float c, float ior) { return FresnelTerm1(c, ior); }
void orc_refract(const float i[3], const float nrm[3], float eta, float out[3]) {
    const XMVECTOR r = XMVector3RefractV(XMVectorSet(i[0], i[1], i[2], 0.f), XMVectorSet(nrm[0], nrm[1], nrm[2], 0.f), XMVectorReplicate(eta));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_reflect(const float i[3], const float nrm[3], float out[3]) {
    const XMVECTOR r = XMVector3Reflect(XMVectorSet(i[0], i[1], i[2], 0.f), XMVectorSet(nrm[0], nrm[1], nrm[2], 0.f));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
uint32_t orc_color_pack(float r, float g, float b, float a) { return XMCOLOR(r, g, b, a).c; }
void orc_color_load(uint32_t argb, float out[4]) {
    XMCOLOR c; c.c = argb;
    const XMVECTOR v = XMLoadColor(&c);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}

namespace {
// A stream that replays three caller-provided uniforms (for orc_unit_scatter).
struct ReplayStream : Xoshiro128 {};
}  // namespace

int orc_unit_scatter(const rt_material* m, const float ray_dir[3], const float pos[3], const float normal[3], const float uv[2],
                     const float draws[3], float out_atten[3], float out_dir[3], uint32_t* n_draws) {
    // Build the one material; feed draws through a scripted stream by temporarily overriding
    // the path stream with a recorder: xoshiro cannot be scripted, so use the scripted hook.
    FlatScene fs;
    fs.spheres.push_back({0.f, 0.f, 0.f, 1.f});
    fs.materials.push_back(*m);
    fs.camera = rt_camera{};
    fs.sun = rt_light{};
    fs.sky = rt_material{};
    SpheresApp app;
    app.LoadScene(fs, 1);
    Payload hit{};
    hit.pos = XMVectorSet(pos[0], pos[1], pos[2], 0.f);
    hit.normal = XMVectorSet(normal[0], normal[1], normal[2], 0.f);
    hit.uv = XMFLOAT2(uv[0], uv[1]);
    Ray ray{XMVectorSet(0.f, 0.f, 0.f, 1.f), XMVectorSet(ray_dir[0], ray_dir[1], ray_dir[2], 0.f)};
    XMVECTOR atten = ORC_XM_Zero;
    Ray outRay{ORC_XM_Zero, ORC_XM_Zero};
    Random::ScriptDraws(draws, 3);
    const bool scattered = app.MaterialOf(0)->Scatter(ray, hit, atten, outRay);
    const uint32_t used = Random::ScriptDrawsUsed();
    Random::ScriptDraws(nullptr, 0);
    out_atten[0] = atten.x; out_atten[1] = atten.y; out_atten[2] = atten.z;
    out_dir[0] = outRay.direction.x; out_dir[1] = outRay.direction.y; out_dir[2] = outRay.direction.z;
    if (n_draws) *n_draws = used;
    return scattered ? 1 : 0;
}

void orc_unit_emit_shade(const rt_material* m, const rt_light* sun, const float view_origin[3], const float pos[3],
                         const float normal[3], const float uv[2], float out_local[3]) {
    FlatScene fs;
    fs.spheres.push_back({0.f, 0.f, 0.f, 1.f});
    fs.materials.push_back(*m);
    fs.camera = rt_camera{};
    fs.sun = *sun;
    fs.sky = rt_material{};
    SpheresApp app;
    app.LoadScene(fs, 1);
    std::vector<std::unique_ptr<Light>> lights;
    lights.push_back(std::make_unique<DirectionalLight>(*sun, [](const Ray&) { return false; }));
    Payload hit{};
    hit.pos = XMVectorSet(pos[0], pos[1], pos[2], 0.f);
    hit.normal = XMVectorSet(normal[0], normal[1], normal[2], 0.f);
    hit.uv = XMFLOAT2(uv[0], uv[1]);
    const Material* mat = app.MaterialOf(0);
    const XMVECTOR local = mat->Emit(hit) + mat->Shade(hit, lights, XMVectorSet(view_origin[0], view_origin[1], view_origin[2], 1.f));
    out_local[0] = local.x; out_local[1] = local.y; out_local[2] = local.z;
}

void orc_xoshiro_seed(uint64_t seed, uint32_t pixel_id, uint32_t sample, uint32_t out_state[4]) {
    Xoshiro128 x;
    x.Seed(seed, pixel_id, sample);
    std::memcpy(out_state, x.s, 16);
}
void orc_xoshiro_draws(uint64_t seed, uint32_t pixel_id, uint32_t sample, uint32_t n, float* out) {
    Xoshiro128 x;
    x.Seed(seed, pixel_id, sample);
    for (uint32_t k = 0; k < n; ++k) out[k] = x.NextUniform();
}
void orc_tonemap(const float hdr_rgb[3], uint32_t n_samples, uint8_t out_rgb[3]) {
    SpheresApp::TonemapPixel(hdr_rgb, n_samples, out_rgb);
}

}  // extern "C"

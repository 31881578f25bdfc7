// rt_host.cpp — host mirror of the reference's class API (see rt_host.h).  Constructors and getters
// run on the host (once per scene); every per-ray method calls librt_hip.so.
#include "rt_host.h"

#include <chrono>
#include <cstring>
#include <string>

namespace {

[[noreturn]] void ThrowRt(const char* where) { throw std::runtime_error(std::string(where) + ": " + rt_last_error()); }
#define RT_CALL(expr)                         \
    do {                                      \
        if ((expr) != RT_OK) ThrowRt(#expr);  \
    } while (0)

rt_ctx* g_ctx = nullptr;
int g_device = 0;
const void* g_sceneOwner = nullptr;  // which hitable's spheres are currently uploaded
thread_local float t_draws[3] = {0.f, 0.f, 0.f};

rt_camera DummyCamera() {
    rt_camera c{};
    c.x[0] = 1.f; c.y[1] = 1.f; c.origin_image_plane[2] = 1.f; c.focal_length = 1.f;
    return c;
}
rt_light DummyLight() {
    rt_light l{};
    l.direction[1] = 1.f;
    return l;
}

// Upload h's spheres as the device scene (cached by owner) and return material owners in list order.
const std::vector<const Material*>& UploadFor(const Hitable* h) {
    static std::vector<const Material*> owners;
    if (g_sceneOwner == h) return owners;
    std::vector<rt_sphere> s;
    std::vector<rt_material> m;
    owners.clear();
    h->Flatten(s, m, &owners);
    const rt_camera cam = DummyCamera();
    const rt_light sun = DummyLight();
    rt_material sky{};
    sky.type = RT_MAT_EMISSIVE;
    RT_CALL(rt_scene_upload(DeviceEval::Context(), s.data(), m.data(), (uint32_t)s.size(), &cam, &sun, 1u, &sky, 1.f));
    g_sceneOwner = h;
    return owners;
}

bool ClosestHit(const Hitable* h, const Ray& ray, Payload& payload) {
    const auto& owners = UploadFor(h);
    const float r[6] = {ray.origin.x, ray.origin.y, ray.origin.z, ray.direction.x, ray.direction.y, ray.direction.z};
    float o[10];
    RT_CALL(rt_unit_closest_hit(DeviceEval::Context(), r, 1, o));
    int32_t idx;
    std::memcpy(&idx, &o[1], 4);
    if (idx < 0) return false;
    payload.t = XMVectorReplicate(o[0]);
    payload.pos = XMVectorSet(o[2], o[3], o[4], 0.f);
    payload.normal = XMVectorSet(o[5], o[6], o[7], 0.f);
    payload.uv = XMFLOAT2(o[8], o[9]);
    payload.material = owners[(size_t)idx];
    return true;
}

// Material::Scatter + Emit + unoccluded sun shade on the device for one hit.
struct ScatterOut {
    bool scattered;
    XMVECTOR atten, dir, local;
};
ScatterOut DeviceScatter(const rt_material& m, const rt_light& sun, const XMVECTOR& viewOrigin, const XMVECTOR& rayDir, const Payload& hit) {
    const float in[12] = {rayDir.x, rayDir.y, rayDir.z, hit.pos.x, hit.pos.y, hit.pos.z, hit.normal.x, hit.normal.y, hit.normal.z,
                          t_draws[0], t_draws[1], t_draws[2]};
    const float vo[3] = {viewOrigin.x, viewOrigin.y, viewOrigin.z};
    float o[11];
    RT_CALL(rt_unit_scatter(DeviceEval::Context(), &m, &sun, vo, in, 1, o));
    return ScatterOut{o[0] != 0.f, XMVectorSet(o[1], o[2], o[3], 0.f), XMVectorSet(o[4], o[5], o[6], 0.f), XMVectorSet(o[8], o[9], o[10], 0.f)};
}

float DeviceMath(uint32_t op, float x, float y = 0.f) {
    float out = 0.f;
    RT_CALL(rt_unit_math(DeviceEval::Context(), op, &x, &y, 1, &out));
    return out;
}

}  // namespace

namespace DeviceEval {
rt_ctx* Context() {
    if (!g_ctx) {
        if (rt_create(g_device, &g_ctx) != RT_OK) ThrowRt("rt_create");
    }
    return g_ctx;
}
void SetDevice(int ordinal) { g_device = ordinal; }
void Shutdown() {
    if (g_ctx) rt_destroy(g_ctx);
    g_ctx = nullptr;
    g_sceneOwner = nullptr;
}
}  // namespace DeviceEval

// ================================================================ ray-tracing.cpp mirror
Sphere::Sphere(const XMVECTOR& c, const float r, std::unique_ptr<Material>&& mat) noexcept  // ray-tracing.cpp:21-24
    : center{c}, radius{r}, material{std::move(mat)} {}

AABB Sphere::GetAABB() const {  // ray-tracing.cpp:86-94
    return AABB{XMFLOAT3(center.x, center.y, center.z), XMFLOAT3(radius, radius, radius)};
}
bool Sphere::Intersect(const Ray& ray, Payload& payload) const { return ClosestHit(this, ray, payload); }  // ray-tracing.cpp:42-84
void Sphere::Flatten(std::vector<rt_sphere>& spheres, std::vector<rt_material>& materials, std::vector<const Material*>* owners) const {
    spheres.push_back(rt_sphere{center.x, center.y, center.z, radius});
    materials.push_back(material->Describe());
    if (owners) owners->push_back(material.get());
}

BvhNode::BvhNode(BvhNode::Iter begin, BvhNode::Iter end) {  // ray-tracing.cpp:107-167
    bool first = true;
    float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    for (auto it = begin; it != end; ++it) {
        const AABB b = (*it)->GetAABB();
        const float c[3] = {b.center.x, b.center.y, b.center.z}, e[3] = {b.extents.x, b.extents.y, b.extents.z};
        for (int k = 0; k < 3; ++k) {
            const float lo = c[k] - e[k], hi = c[k] + e[k];
            if (first || lo < mn[k]) mn[k] = lo;
            if (first || hi > mx[k]) mx[k] = hi;
        }
        first = false;
        m_items.push_back(std::move(*it));  // ownership moves out of the caller's vector, as in the reference
    }
    m_aabb = AABB{XMFLOAT3((mn[0] + mx[0]) * 0.5f, (mn[1] + mx[1]) * 0.5f, (mn[2] + mx[2]) * 0.5f),
                  XMFLOAT3((mx[0] - mn[0]) * 0.5f, (mx[1] - mn[1]) * 0.5f, (mx[2] - mn[2]) * 0.5f)};
}
AABB BvhNode::GetAABB() const { return m_aabb; }                                                              // ray-tracing.cpp:169-172
bool BvhNode::Intersect(const Ray& ray, Payload& payload) const { return ClosestHit(this, ray, payload); }  // ray-tracing.cpp:174-214
void BvhNode::Flatten(std::vector<rt_sphere>& spheres, std::vector<rt_material>& materials, std::vector<const Material*>* owners) const {
    for (const auto& h : m_items) h->Flatten(spheres, materials, owners);
}

// ==================================================================== texture.cpp mirror
ConstTexture::ConstTexture(const XMCOLOR& color) { m_color = XMLoadColor(&color); }  // texture.cpp:3-6
XMVECTOR ConstTexture::Evaluate(XMFLOAT2) const { return m_color; }                  // texture.cpp:8-11
void ConstTexture::Describe(rt_material& m) const {
    m.tex_type = RT_TEX_CONST;
    m.rgb0[0] = m_color.x; m.rgb0[1] = m_color.y; m.rgb0[2] = m_color.z;
}
CheckerTexture::CheckerTexture(const XMCOLOR& color0, const XMCOLOR& color1, float tiling) : m_tilingScale{tiling} {  // texture.cpp:13-18
    m_checkerColors[0] = XMLoadColor(&color0);
    m_checkerColors[1] = XMLoadColor(&color1);
}
XMVECTOR CheckerTexture::Evaluate(XMFLOAT2 uv) const {  // texture.cpp:20-33 (integer parity of the truncated tile coordinates)
    const auto u = static_cast<int>(m_tilingScale * uv.x);
    const auto v = static_cast<int>(m_tilingScale * uv.y);
    return (u % 2 == v % 2) ? m_checkerColors[0] : m_checkerColors[1];
}
void CheckerTexture::Describe(rt_material& m) const {
    m.tex_type = RT_TEX_CHECKER;
    m.tiling = m_tilingScale;
    m.rgb0[0] = m_checkerColors[0].x; m.rgb0[1] = m_checkerColors[0].y; m.rgb0[2] = m_checkerColors[0].z;
    m.rgb1[0] = m_checkerColors[1].x; m.rgb1[1] = m_checkerColors[1].y; m.rgb1[2] = m_checkerColors[1].z;
}

// ====================================================================== light.cpp mirror
DirectionalLight::DirectionalLight(const XMVECTOR& dir, const XMCOLOR& color, const float luminance,
                                   std::function<bool(const Ray& ray)> lightOcclusionTest)  // light.cpp:4-9
    : m_luminance{luminance}, IsOccluded{std::move(lightOcclusionTest)} {
    m_direction = XMVector3Normalize(dir);
    m_color = XMLoadColor(&color);
}
rt_light DirectionalLight::Describe() const {
    rt_light l{};
    l.direction[0] = m_direction.x; l.direction[1] = m_direction.y; l.direction[2] = m_direction.z;
    l.color[0] = m_color.x; l.color[1] = m_color.y; l.color[2] = m_color.z;
    l.luminance = m_luminance;
    return l;
}
XMVECTOR DirectionalLight::Shade(const Material* material, const Payload& payload, const XMVECTOR& viewOrigin) const {  // light.cpp:11-42
    Ray shadowRay{payload.pos, m_direction};
    if (IsOccluded && IsOccluded(shadowRay)) return XM_Zero;
    rt_material m = material->Describe();
    m.luminance = 0.f;  // the device returns Emit + Shade; with zero luminance that is 0 + Shade
    return DeviceScatter(m, Describe(), viewOrigin, XMVectorSet(0.f, 0.f, 1.f, 0.f), payload).local;
}

// =================================================================== material.cpp mirror
void Material::SetScatterDraws(float u0, float u1, float u2) { t_draws[0] = u0; t_draws[1] = u1; t_draws[2] = u2; }

bool Material::Scatter(const Ray& ray, const Payload& payload, XMVECTOR& outAttenuation, Ray& outRay) const {  // material.cpp:20-164
    const ScatterOut o = DeviceScatter(Describe(), DummyLight(), XM_Zero, ray.direction, payload);
    outAttenuation = o.atten;
    outRay = Ray{payload.pos, o.dir};
    return o.scattered;
}
XMVECTOR Material::Shade(const Payload& payload, const std::vector<std::unique_ptr<Light>>& lights, const XMVECTOR& viewOrigin) const {  // material.cpp:4-13
    XMVECTOR directLighting = XM_Zero;
    for (const auto& light : lights) directLighting = directLighting + light->Shade(this, payload, viewOrigin);
    return directLighting;
}
Metal::Metal(const Texture* reflectance, const XMVECTOR& smoothness) : m_reflectance{reflectance}, m_smoothness{smoothness} {}  // material.cpp:67-70
rt_material Metal::Describe() const {
    rt_material m{};
    m.type = RT_MAT_METAL;
    m.smoothness = m_smoothness.x;
    m_reflectance->Describe(m);
    return m;
}
DielectricOpaque::DielectricOpaque(const Texture* albedo, const XMVECTOR& smoothness) : m_albedo{albedo}, m_smoothness{smoothness} {}  // material.cpp:15-18
rt_material DielectricOpaque::Describe() const {
    rt_material m{};
    m.type = RT_MAT_DIELECTRIC_OPAQUE;
    m.smoothness = m_smoothness.x;
    m_albedo->Describe(m);
    return m;
}
DielectricTransparent::DielectricTransparent(const XMVECTOR& smoothness, const float ior) : m_smoothness{smoothness} {  // material.cpp:105-109
    m_ior = XMVectorReplicate(ior);
}
rt_material DielectricTransparent::Describe() const {
    rt_material m{};
    m.type = RT_MAT_DIELECTRIC_TRANSPARENT;
    m.tex_type = RT_TEX_CONST;
    m.smoothness = m_smoothness.x;
    m.ior = m_ior.x;
    return m;
}
Emissive::Emissive(const float luminance, const Texture* color) : m_color{color}, m_luminance{luminance} {}  // material.cpp:166-170
rt_material Emissive::Describe() const {
    rt_material m{};
    m.type = RT_MAT_EMISSIVE;
    m.luminance = m_luminance;
    m_color->Describe(m);
    return m;
}
XMVECTOR Emissive::Emit(const Payload& payload) const {  // material.cpp:172-175 — luminance * colour(uv), a getter-level product
    return m_luminance * m_color->Evaluate(payload.uv);
}

// ===================================================================== camera.cpp mirror
Camera::Camera(const XMVECTOR origin, const XMVECTOR lookAt, const float verticalFOV, const float aspectRatio, const float focalLength,
               const float aperture)  // camera.cpp:3-28 — once per scene, host side
    : m_origin{origin}, m_aperture{aperture}, m_focalLength{focalLength} {
    const float theta = verticalFOV * XM_PI / 180.f;
    const float halfHeight = rtd::rt_tanf(theta / 2.f);
    const float halfWidth = aspectRatio * halfHeight;
    const XMVECTORF32 up{0.f, 1.f, 0.f, 0.f};
    const XMVECTOR w = XMVector3Normalize(lookAt - origin);
    const XMVECTOR u = XMVector3Normalize(XMVector3Cross(up, w));
    const XMVECTOR v = XMVector3Cross(w, u);
    const float imagePlaneOffset = 1.f;
    m_originImagePlane = origin + imagePlaneOffset * w;
    m_x = halfWidth * u;
    m_y = halfHeight * v;
}
rt_camera Camera::Describe() const {
    rt_camera c{};
    const XMVECTOR* src[4] = {&m_origin, &m_x, &m_y, &m_originImagePlane};
    float* dst[4] = {c.origin, c.x, c.y, c.origin_image_plane};
    for (int k = 0; k < 4; ++k) {
        dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; dst[k][3] = src[k]->w;
    }
    c.aperture = m_aperture;
    c.focal_length = m_focalLength;
    return c;
}
Ray Camera::GetRay(XMFLOAT2 uv, XMFLOAT2 offset) const {  // camera.cpp:30-48
    const rt_camera c = Describe();
    const float in[4] = {uv.x, uv.y, offset.x, offset.y};
    float o[6];
    RT_CALL(rt_unit_camera_rays(DeviceEval::Context(), &c, in, 1, o));
    return Ray{XMVectorSet(o[0], o[1], o[2], 1.f), XMVectorSet(o[3], o[4], o[5], 0.f)};
}
XMVECTOR Camera::GetOrigin() const { return m_origin; }  // camera.cpp:50-53

// =============================================================== quasi-random.cpp mirror
namespace Random {
float HaltonSample(uint64_t sampleIndex, uint32_t base) {  // quasi-random.cpp:3-16
    if (sampleIndex > 0xffffffffull) throw std::runtime_error("Random::HaltonSample: index beyond 2^32 is off the render path");
    const uint32_t idx = (uint32_t)sampleIndex;
    float out = 0.f;
    RT_CALL(rt_unit_halton(DeviceEval::Context(), &idx, base, 1, &out));
    return out;
}
XMFLOAT2 HaltonSample2D(uint64_t sampleIndex, uint32_t base1, uint32_t base2) {  // :18-24
    return XMFLOAT2(HaltonSample(sampleIndex, base1), HaltonSample(sampleIndex, base2));
}
XMFLOAT2 HaltonSampleRing(uint64_t sampleIndex, uint32_t base) {  // :26-34
    const float theta = 2.f * XM_PI * HaltonSample(sampleIndex, base);
    return XMFLOAT2(DeviceMath(1, theta), DeviceMath(0, theta));
}
XMFLOAT3 HaltonSampleHemisphere(uint64_t sampleIndex, uint32_t base1, uint32_t base2) {  // :36-50
    const float u1 = HaltonSample(sampleIndex, base1);
    const float u2 = HaltonSample(sampleIndex, base2);
    const float r = __builtin_sqrtf(1.f - u1 * u1);
    const float phi = 2 * XM_PI * u2;
    return XMFLOAT3(r * DeviceMath(1, phi), r * DeviceMath(0, phi), u1);
}
XMFLOAT2 HaltonSampleDisk(uint64_t sampleIndex, uint32_t base1, uint32_t base2) {  // :52-61
    const float theta = 2.f * XM_PI * HaltonSample(sampleIndex, base1);
    const float r = HaltonSample(sampleIndex, base2);
    return XMFLOAT2(r * DeviceMath(1, theta), r * DeviceMath(0, theta));
}
uint64_t Xorshift() {  // :65-76
    static auto startTime = std::chrono::high_resolution_clock::now();
    const auto timeNow = std::chrono::high_resolution_clock::now();
    uint64_t x = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(timeNow - startTime).count();
    x ^= x >> 12;
    x ^= x << 25;
    x ^= x >> 27;
    return x;
}
}  // namespace Random

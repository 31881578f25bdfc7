// ORACLE — TEST INFRASTRUCTURE ONLY (see dxmath_restate.h header).  PARITY UNPINNED with respect
// to the original Windows binary (it cannot be built here and is non-deterministic, SURVEY.md §0
// F1/F2); pinned instead by the Halton known answers captured from the reference's own
// quasi-random.cpp (SURVEY.md §8(c) -> tests/golden/halton_known_answers.json) and by closed-form
// checks of every restated function.
//
// rt_oracle.h — CPU restatement of the reference's render-loop hot path, class for class:
//   Ray, Payload, AABB, Hitable, Sphere, BvhNode      common-lib/ray-tracing.{h,cpp}
//   Texture, ConstTexture, CheckerTexture             common-lib/texture.{h,cpp}
//   Light, DirectionalLight                           common-lib/light.{h,cpp}
//   Material, Metal, DielectricOpaque,
//   DielectricTransparent, Emissive                   common-lib/material.{h,cpp}
//   Camera                                            common-lib/camera.{h,cpp}
//   Random::Halton*                                   common-lib/quasi-random.{h,cpp}
//   SpheresApp (headless)                             spheres/spheres-app.{h,cpp}
// Deviations from the reference, all forced by determinism:
//   * material random draws come from a per-(pixel,s) xoshiro128** stream instead of the racy
//     per-material global Halton counters (material.h:34,50-51,67);
//   * GetHitColor is written iteratively (throughput/radiance), algebraically equal to the
//     recursion at spheres-app.cpp:238-257;
//   * the closest hit may be found by a linear list scan (tie: lower index) instead of BvhNode;
//   * XMVectorReciprocalEst is the exact reciprocal; sinf/cosf/powf/tanf are the f64 kernels of
//     dxmath_restate.h.
#pragma once

#include <array>
#include <atomic>
#include <functional>
#include <memory>
#include <optional>
#include <vector>

#include "dxmath_restate.h"
#include "../include/rt_api.h"

namespace orc {

// ------------------------------------------------------------------ RNG contract (A9)
inline uint64_t splitmix64_mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
struct Xoshiro128 {
    uint32_t s[4];
    void Seed(uint64_t seed, uint32_t pixelId, uint32_t sample) {
        const uint64_t key = ((uint64_t)pixelId << 32) | (uint64_t)sample;
        const uint64_t a = splitmix64_mix(splitmix64_mix(seed) ^ key);
        const uint64_t b = splitmix64_mix(a);
        s[0] = (uint32_t)a; s[1] = (uint32_t)(a >> 32); s[2] = (uint32_t)b; s[3] = (uint32_t)(b >> 32);
        if ((s[0] | s[1] | s[2] | s[3]) == 0u) s[0] = 1u;
    }
    static uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
    uint32_t Next() {  // xoshiro128**
        const uint32_t result = rotl(s[1] * 5u, 7) * 9u;
        const uint32_t t = s[1] << 9;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 11);
        return result;
    }
    float NextUniform() { return (float)(Next() >> 8) * 0x1p-24f; }
};

namespace Random {
// quasi-random.h:5-14
float HaltonSample(uint64_t sampleIndex, uint32_t base);
XMFLOAT2 HaltonSample2D(uint64_t sampleIndex, uint32_t base1, uint32_t base2);
XMFLOAT2 HaltonSampleRing(uint64_t sampleIndex, uint32_t base);
XMFLOAT2 HaltonSampleDisk(uint64_t sampleIndex, uint32_t base1, uint32_t base2);
XMFLOAT3 HaltonSampleHemisphere(uint64_t sampleIndex, uint32_t base1, uint32_t base2);
// uniform-hemisphere mapping of HaltonSampleHemisphere applied to two given uniforms
XMFLOAT3 HemisphereFromUniforms(float u1, float u2);
// The path's material draw stream (replaces `counter++` -> HaltonSample at material.cpp:29,44,82,151).
void BindPathStream(Xoshiro128* stream);
float NextMaterialDraw();
// Reference-faithful sampler mode: when enabled, a material draw is
// HaltonSample(counter++, base) on the material's own counter exactly as material.cpp:29,44,82,151 do —
// only meaningful for a SERIAL render (the reference's counters are racy under its thread pool).  Used by the
// statistical parity test; the deterministic per-path stream stays the contract for the GPU parity tests.
void SetSamplerFlags(uint32_t flags);  // RT_SAMPLER_* (include/rt_api.h); 0 = the reference's mappings
uint32_t SamplerFlags();
void UseReferenceHaltonCounters(bool on);
bool ReferenceHaltonCounters();
float Draw(uint64_t& counter, uint32_t base);                                      // one uniform
XMFLOAT3 DrawHemisphere(uint64_t& counter, uint32_t base1, uint32_t base2);        // quasi-random.cpp:36-50
// Unit tests only: replay caller-provided uniforms instead of the bound stream.
void ScriptDraws(const float* draws, uint32_t n);
uint32_t ScriptDrawsUsed();
}  // namespace Random

// CPU-only diagnostic: evaluate GetHitColor in the reference's own nesting (spheres-app.cpp:249-251) instead of the
// forward throughput form that is the path's contract (see GetHitColorNested in rt_oracle.cpp).
void UseNestedRadiance(bool on);
// BvhNode::Intersect gives exact ties to the right child, as the reference does (default, true); false (CPU diagnostic): to the
// lower list index, the list scan's rule (see rt_oracle.cpp).
void UseReferenceBvhTieRule(bool on);
bool NestedRadiance();

// ------------------------------------------------------------------ geometry
struct alignas(16) Payload {  // ray-tracing.h:5-13
    XMVECTOR t;
    XMVECTOR pos;
    XMVECTOR normal;
    XMFLOAT2 uv;
    const class Material* material;
    int index;  // list index of the sphere (oracle addition: tie-break + tests)
};

struct alignas(16) Ray {  // ray-tracing.h:15-24
    XMVECTOR origin;
    XMVECTOR direction;
    Ray() = default;
    Ray(const XMVECTOR& o, const XMVECTOR& d) noexcept : origin{o}, direction{d} {}
    XMVECTOR Evaluate(float t) const;
};

struct AABB {  // ray-tracing.h:26-33
    BoundingBox m_box;
    AABB() = default;
    AABB(const XMFLOAT3& center, const XMFLOAT3& extents);
    bool Intersect(const Ray& ray) const;
};

struct Hitable {  // ray-tracing.h:35-39 (+ virtual dtor)
    virtual ~Hitable() = default;
    virtual AABB GetAABB() const = 0;
    virtual bool Intersect(const Ray& ray, Payload& payload) const = 0;
};

struct BvhNode : public Hitable {  // ray-tracing.h:41-51
    AABB m_aabb;
    std::unique_ptr<Hitable> m_left;
    std::unique_ptr<Hitable> m_right;
    using Iter = std::vector<std::unique_ptr<Hitable>>::iterator;
    BvhNode(Iter begin, Iter end);
    AABB GetAABB() const override;
    bool Intersect(const Ray& ray, Payload& payload) const override;
};

struct Sphere : public Hitable {  // ray-tracing.h:53-65
    alignas(16) XMVECTOR center;
    float radius;
    std::shared_ptr<const class Material> material;  // shared so list and BVH copies see one object
    int index = -1;
    Sphere(const XMVECTOR& c, const float r, std::shared_ptr<const class Material> mat) noexcept;
    AABB GetAABB() const override;
    bool Intersect(const Ray& ray, Payload& payload) const override;

private:
    XMFLOAT2 ComputeUV(const XMVECTOR& worldPos) const;
};

// Linear scan stand-in for BvhNode: smaller t wins, equal t -> lower index.
struct HitableList : public Hitable {
    std::vector<std::unique_ptr<Hitable>> items;
    AABB GetAABB() const override { return AABB{}; }
    bool Intersect(const Ray& ray, Payload& payload) const override;
};

// ------------------------------------------------------------------ textures
class Texture {  // texture.h:6-10
public:
    virtual ~Texture() = default;
    virtual XMVECTOR Evaluate(XMFLOAT2 uv) const = 0;
};
class ConstTexture : public Texture {  // texture.h:12-20
public:
    explicit ConstTexture(const XMCOLOR& color);
    XMVECTOR Evaluate(XMFLOAT2 uv) const override;
private:
    XMVECTOR m_color;
};
class CheckerTexture : public Texture {  // texture.h:22-31
public:
    CheckerTexture(const XMCOLOR& color0, const XMCOLOR& color1, float tiling);
    XMVECTOR Evaluate(XMFLOAT2 uv) const override;
private:
    std::array<XMVECTOR, 2> m_checkerColors;
    float m_tilingScale;
};

// -------------------------------------------------------------------- lights
class Light {  // light.h:6-10
public:
    virtual ~Light() = default;
    virtual XMVECTOR Shade(const class Material* material, const Payload& payload, const XMVECTOR& viewOrigin) const = 0;
};
class DirectionalLight : public Light {  // light.h:12-22
public:
    DirectionalLight(const XMVECTOR& dir, const XMCOLOR& color, const float luminance,
                     std::function<bool(const Ray& ray)> lightOcclusionTest);
    // flat-table form: direction already normalised, colour already loaded (rt_light)
    DirectionalLight(const rt_light& flat, std::function<bool(const Ray& ray)> lightOcclusionTest);
    XMVECTOR Shade(const class Material* material, const Payload& payload, const XMVECTOR& viewOrigin) const override;
    XMVECTOR GetDirection() const { return m_direction; }
    XMVECTOR GetColor() const { return m_color; }
    float GetLuminance() const { return m_luminance; }
private:
    XMVECTOR m_direction;
    XMVECTOR m_color;
    float m_luminance;
    std::function<bool(const Ray& ray)> IsOccluded;
};

// ----------------------------------------------------------------- materials
class Material {  // material.h:8-19
public:
    virtual ~Material() = default;
    virtual bool Scatter(const Ray& ray, const Payload& payload, XMVECTOR& outAttenuation, Ray& outRay) const = 0;
    virtual XMVECTOR Shade(const Payload& payload, const std::vector<std::unique_ptr<Light>>& lights, const XMVECTOR& viewOrigin) const;
    virtual XMVECTOR Emit(const Payload& payload) const = 0;
    virtual XMVECTOR GetAlbedo(XMFLOAT2 uv) const = 0;
    virtual XMVECTOR GetReflectance(XMFLOAT2 uv) const = 0;
    virtual XMVECTOR GetSmoothness(XMFLOAT2 uv) const = 0;
};
class Metal : public Material {  // material.h:21-35
public:
    Metal(const Texture* reflectance, const XMVECTOR& smoothness);
    bool Scatter(const Ray& ray, const Payload& payload, XMVECTOR& outAttenuation, Ray& outRay) const override;
    XMVECTOR Emit(const Payload&) const override { return ORC_XM_Zero; }
    XMVECTOR GetAlbedo(XMFLOAT2) const override { return ORC_XM_Zero; }
    XMVECTOR GetReflectance(XMFLOAT2 uv) const override { return m_reflectance->Evaluate(uv); }
    XMVECTOR GetSmoothness(XMFLOAT2) const override { return m_smoothness; }
private:
    const Texture* m_reflectance;
    XMVECTOR m_smoothness;
    mutable uint64_t m_reflectionProbabilitySampleIndex = 0u;  // material.h:34 (std::atomic there)
};
class DielectricOpaque : public Material {  // material.h:37-52
public:
    DielectricOpaque(const Texture* albedo, const XMVECTOR& smoothness);
    bool Scatter(const Ray& ray, const Payload& payload, XMVECTOR& outAttenuation, Ray& outRay) const override;
    XMVECTOR Emit(const Payload&) const override { return ORC_XM_Zero; }
    XMVECTOR GetAlbedo(XMFLOAT2 uv) const override { return m_albedo->Evaluate(uv); }
    XMVECTOR GetReflectance(XMFLOAT2) const override { return XMVECTORF32{0.04f, 0.04f, 0.04f, 1.f}; }
    XMVECTOR GetSmoothness(XMFLOAT2) const override { return m_smoothness; }
private:
    const Texture* m_albedo;
    XMVECTOR m_smoothness;
    mutable uint64_t m_sampleIndex = 0u;                       // material.h:50
    mutable uint64_t m_reflectionProbabilitySampleIndex = 0u;  // material.h:51
};
class DielectricTransparent : public Material {  // material.h:54-68
public:
    DielectricTransparent(const XMVECTOR& smoothness, float ior);
    bool Scatter(const Ray& ray, const Payload& payload, XMVECTOR& outAttenuation, Ray& outRay) const override;
    XMVECTOR Emit(const Payload&) const override { return ORC_XM_Zero; }
    XMVECTOR GetAlbedo(XMFLOAT2) const override { return ORC_XM_Zero; }
    XMVECTOR GetReflectance(XMFLOAT2) const override { return XMVECTORF32{0.04f, 0.04f, 0.04f, 1.f}; }
    XMVECTOR GetSmoothness(XMFLOAT2) const override { return m_smoothness; }
private:
    XMVECTOR m_smoothness;
    XMVECTOR m_ior;
    mutable uint64_t m_sampleIndex = 0u;  // material.h:67
};
class Emissive : public Material {  // material.h:70-84
public:
    Emissive(const float luminance, const Texture* color);
    XMVECTOR Emit(const Payload& payload) const override;
    bool Scatter(const Ray&, const Payload&, XMVECTOR&, Ray&) const override { return false; }
    XMVECTOR GetAlbedo(XMFLOAT2) const override { return ORC_XM_Zero; }
    XMVECTOR GetReflectance(XMFLOAT2) const override { return ORC_XM_Zero; }
    XMVECTOR GetSmoothness(XMFLOAT2) const override { return ORC_XM_Zero; }
private:
    const Texture* m_color;
    float m_luminance;
};

// -------------------------------------------------------------------- camera
class Camera {  // camera.h:6-20
public:
    Camera(XMVECTOR origin, XMVECTOR lookAt, float verticalFOV, float aspectRatio, float focalLength, float aperture);
    explicit Camera(const rt_camera& flat);
    Ray GetRay(XMFLOAT2 uv, XMFLOAT2 offset) const;
    XMVECTOR GetOrigin() const;
    rt_camera Flatten() const;
private:
    XMVECTOR m_origin;
    XMVECTOR m_x;
    XMVECTOR m_y;
    XMVECTOR m_originImagePlane;
    float m_aperture;
    float m_focalLength;
};

// ------------------------------------------------------- flat scene (rt_api.h records)
struct FlatScene {
    std::vector<rt_sphere> spheres;
    std::vector<rt_material> materials;
    rt_camera camera{};
    rt_light sun{};                // the scene generators' single light (spheres-app.cpp:129) ...
    std::vector<rt_light> lights;  // ... and the list the app is loaded with (spheres-app.h:38 m_lights); empty: {sun}
    bool lightsGiven = false;      // true: `lights` is the list, even when it is empty (Material::Shade then returns XM_Zero)
    rt_material sky{};
    float exposureScale = 0.f;
};

// Scene generators: InitScene/InitCamera (spheres-app.cpp:35-130) with a fixed seed, and the
// BASELINE configs' synthetic scenes.  name: "cover", "three", "grid10k".
bool BuildNamedScene(const char* name, uint64_t seed, float aspect, float apertureOverride /*<0: default*/, FlatScene& out);

// ------------------------------------------------------------- list scan, accelerated (oracle infrastructure)
// The path's contract is the LIST scan: Sphere::Intersect for every sphere, smallest t, ties to the lower index
// (SURVEY.md §8a A6).  The reference's BvhNode is not that function: its slab test (BoundingBox::Intersects in binary32) and
// the sphere test (also binary32, with a discriminant that is off by up to 16 eps a G far from the origin) disagree about
// grazing rays, so on the 10,004-sphere scene BvhNode loses about 2 hits in 10^7 paths that the list finds (and an exact tie
// goes to whichever sphere the random tree puts right).  Running the list itself on 10^9 paths x 10^4 spheres is out of
// reach, so the full-size differentials use THIS accelerator: a median-split tree over the same spheres whose boxes are
// padded, per ray, by more than any accepted root's hit point can lie outside its sphere -- the discriminant's error bound
// E <= 16 eps a G, G = 2|o|^2 + 2(|c|+r)^2 + r^2, puts that point within sqrt(2E/a) of the surface; the padding is
// sqrt(128 eps G_node) with G_node from the node's largest |c|+r and r (four times the bound), and the slab test runs in
// binary64 over the forward half-line.  Every sphere whose reference-order test accepts a root is therefore visited, each
// visited sphere is tested by Sphere::Intersect itself, and the merge is the list's (smaller t, then lower index): the result
// IS the list scan's.  tests/test_oracle_units.py checks it against the plain list on grazing and random rays.
struct PaddedListTree : public Hitable {
    struct Node {
        double lo[3], hi[3];
        double reach, rmax;  // max |c| + r and max r over the node's spheres
        int left, right;     // children, or -1
        int sphere;          // leaf: index into spheres
    };
    std::vector<Node> nodes;
    std::vector<const Sphere*> spheres;
    explicit PaddedListTree(const std::vector<const Sphere*>& list);
    AABB GetAABB() const override { return AABB{}; }
    bool Intersect(const Ray& ray, Payload& payload) const override;

private:
    int Build(std::vector<int>& ids, int begin, int end);
};

// ------------------------------------------------------------- headless SpheresApp
struct RenderCounters {
    uint64_t samples = 0, traversals = 0, segments = 0;
};

class SpheresApp {  // spheres-app.h:15-43, minus the Win32/D2D members
public:
    enum class Accel { List, Bvh, PaddedList };
    const Hitable* AccelFor(Accel a) const;
    void LoadScene(const FlatScene& flat, uint64_t bvhAxisSeed);

    // GenerateRays for one pixel (spheres-app.cpp:132-161): sample index s is 1-based (:168).
    Ray GeneratePrimaryRay(uint32_t W, uint32_t H, uint32_t i, uint32_t j, uint32_t s) const;
    std::optional<Payload> GetClosestIntersection(const Ray& ray) const;  // :224-236
    XMVECTOR GetHitColor(const Ray& ray, int depth0) const;               // :238-257 (iterative)
    XMVECTOR GetHitColorNested(const Ray& ray, int depth) const;  // the reference's nesting, CPU-only diagnostic (rt_oracle.cpp)

    // DrawBitmap's trace loop (:174-184) over a row set and sample range, accumulating into hdr.
    void Render(uint32_t W, uint32_t H, rt_rowset rs, uint32_t s0, uint32_t s1, uint32_t maxDepth, uint64_t seed,
                Accel accel, int threads, RenderCounters& counters);
    void Clear();
    // DrawBitmap's tonemap (:186-214)
    void Resolve(uint32_t nSamples);
    std::optional<Payload> ClosestHitWith(const Ray& ray, Accel accel) const;
    const Material* MaterialOf(size_t sphereIndex) const;
    static void TonemapPixel(const float hdrRgb[3], uint32_t nSamples, uint8_t outRgb[3]);
    static XMCOLOR TonemapColor(const XMVECTOR& hdrColor, uint32_t n);
    XMVECTOR TraceSample(uint32_t W, uint32_t H, uint32_t i, uint32_t j, uint32_t s, uint32_t maxDepth, uint64_t seed,
                         Accel accel, uint32_t* traversals) const;
    XMVECTOR TraceSampleRecorded(uint32_t W, uint32_t H, uint32_t i, uint32_t j, uint32_t s, uint32_t maxDepth, uint64_t seed,
                                 Accel accel, std::vector<float>& rays) const;

    const std::vector<XMVECTOR>& Hdr() const { return m_backbufferHdr; }
    const std::vector<XMCOLOR>& Ldr() const { return m_backbufferLdr; }
    uint32_t StripWidth() const { return m_stripW; }
    uint32_t StripRows() const { return m_stripRows; }
    uint32_t SampleCount() const { return m_sampleCount; }
    const Camera& GetCamera() const { return *m_camera; }
    bool HasScene() const { return (bool)m_camera; }

private:
    std::unique_ptr<Camera> m_camera;
    std::unique_ptr<HitableList> m_sceneList;
    std::vector<std::unique_ptr<Texture>> m_textures;
    std::vector<std::unique_ptr<Light>> m_lights;
    std::unique_ptr<BvhNode> m_bvh;
    std::unique_ptr<PaddedListTree> m_paddedList;
    std::unique_ptr<Material> m_skyMaterial;
    float m_exposureScale = 0.f;
    uint32_t m_sampleCount = 0;
    mutable const Hitable* m_activeAccel = nullptr;
    mutable int m_maxDepth = 50;

    std::vector<XMVECTOR> m_backbufferHdr;
    std::vector<XMCOLOR> m_backbufferLdr;
    uint32_t m_stripW = 0, m_stripRows = 0;
};

uint32_t RowsetLocalRows(rt_rowset rs);
uint32_t RowsetGlobalRow(rt_rowset rs, uint32_t localRow);

}  // namespace orc

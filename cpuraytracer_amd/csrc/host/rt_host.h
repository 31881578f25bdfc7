// rt_host.h — C++17 host mirror of the reference's class API for the render-loop path
// (SURVEY.md §8b): same names, constructor signatures, const-ness and ownership as
//   common-lib/ray-tracing.h, material.h, texture.h, light.h, camera.h, quasi-random.h
// so code written against the reference's headers builds against these.  The objects are scene
// DESCRIPTIONS: they flatten into the rt_api.h tables that librt_hip.so renders.  The per-call hot
// methods (Camera::GetRay, Hitable::Intersect, Material::Scatter/Shade/Emit) evaluate ON THE DEVICE
// through the library's unit entry points — there is no CPU implementation of them here, and they
// throw std::runtime_error when no GPU is present.
#pragma once

#include <array>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <vector>

#include "../../../include/rt_api.h"
#include "xm_types.h"

class Material;
class Texture;

// Bound device context for the per-call methods (one per process, device 0 unless set).
namespace DeviceEval {
rt_ctx* Context();           // lazily rt_create(0); throws std::runtime_error without a GPU
void SetDevice(int ordinal);  // before first use
void Shutdown();
}  // namespace DeviceEval

// ------------------------------------------------------------------ ray-tracing.h
struct alignas(16) Payload {  // ray-tracing.h:5-13
    XMVECTOR t;
    XMVECTOR pos;
    XMVECTOR normal;
    XMFLOAT2 uv;
    const Material* material;
};

struct alignas(16) Ray {  // ray-tracing.h:15-24
    XMVECTOR origin;
    XMVECTOR direction;
    Ray() = default;
    Ray(const XMVECTOR& o, const XMVECTOR& d) noexcept : origin{o}, direction{d} {}
};

struct AABB {  // ray-tracing.h:26-33 (centre / extents of DirectX::BoundingBox)
    XMFLOAT3 center{0, 0, 0};
    XMFLOAT3 extents{0, 0, 0};
    AABB() = default;
    AABB(const XMFLOAT3& c, const XMFLOAT3& e) : center(c), extents(e) {}
};

struct Hitable {  // ray-tracing.h:35-39
    virtual ~Hitable() = default;
    virtual AABB GetAABB() const = 0;
    virtual bool Intersect(const Ray& ray, Payload& payload) const = 0;  // device evaluated
    // flatten this hitable's spheres (and their materials) in list order
    virtual void Flatten(std::vector<rt_sphere>& spheres, std::vector<rt_material>& materials,
                         std::vector<const Material*>* owners = nullptr) const = 0;
};

struct Sphere : public Hitable {  // ray-tracing.h:53-65
    alignas(16) XMVECTOR center;
    float radius;
    std::unique_ptr<Material> material;
    Sphere(const XMVECTOR& c, const float r, std::unique_ptr<Material>&& mat) noexcept;
    AABB GetAABB() const override;
    bool Intersect(const Ray& ray, Payload& payload) const override;
    void Flatten(std::vector<rt_sphere>& spheres, std::vector<rt_material>& materials,
                 std::vector<const Material*>* owners = nullptr) const override;
};

// BvhNode (ray-tracing.h:41-51): takes ownership by moving out of the caller's vector exactly like
// the reference (ray-tracing.cpp:113,118-119).  On MI355X the closest hit is found by the LDS list
// scan (result-equivalent), so the node keeps its hitables as a list.
struct BvhNode : public Hitable {
    using Iter = std::vector<std::unique_ptr<Hitable>>::iterator;
    BvhNode(Iter begin, Iter end);
    AABB GetAABB() const override;
    bool Intersect(const Ray& ray, Payload& payload) const override;
    void Flatten(std::vector<rt_sphere>& spheres, std::vector<rt_material>& materials,
                 std::vector<const Material*>* owners = nullptr) const override;
    size_t Size() const { return m_items.size(); }

private:
    std::vector<std::unique_ptr<Hitable>> m_items;
    AABB m_aabb;
};

// ---------------------------------------------------------------------- texture.h
class Texture {  // texture.h:6-10
public:
    virtual ~Texture() = default;
    virtual XMVECTOR Evaluate(XMFLOAT2 uv) const = 0;
    virtual void Describe(rt_material& m) const = 0;  // fill tex_type / rgb0 / rgb1 / tiling
};
class ConstTexture : public Texture {  // texture.h:12-20
public:
    ConstTexture(const XMCOLOR& color);
    XMVECTOR Evaluate(XMFLOAT2 uv) const override;
    void Describe(rt_material& m) const override;
private:
    XMVECTOR m_color;
};
class CheckerTexture : public Texture {  // texture.h:22-31
public:
    CheckerTexture(const XMCOLOR& color0, const XMCOLOR& color1, float tiling);
    XMVECTOR Evaluate(XMFLOAT2 uv) const override;
    void Describe(rt_material& m) const override;
private:
    std::array<XMVECTOR, 2> m_checkerColors;
    float m_tilingScale;
};

// ------------------------------------------------------------------------ light.h
class Light {  // light.h:6-10
public:
    virtual ~Light() = default;
    virtual XMVECTOR Shade(const Material* material, const Payload& payload, const XMVECTOR& viewOrigin) const = 0;
    virtual rt_light Describe() const = 0;
};
class DirectionalLight : public Light {  // light.h:12-22
public:
    DirectionalLight(const XMVECTOR& dir, const XMCOLOR& color, const float luminance,
                     std::function<bool(const Ray& ray)> lightOcclusionTest);
    XMVECTOR Shade(const Material* material, const Payload& payload, const XMVECTOR& viewOrigin) const override;
    rt_light Describe() const override;
private:
    XMVECTOR m_direction;
    XMVECTOR m_color;
    float m_luminance;
    std::function<bool(const Ray& ray)> IsOccluded;
};

// --------------------------------------------------------------------- material.h
class Material {  // material.h:8-19
public:
    virtual ~Material() = default;
    virtual bool Scatter(const Ray& ray, const Payload& payload, XMVECTOR& outAttenuation, Ray& outRay) const;  // device evaluated
    virtual XMVECTOR Shade(const Payload& payload, const std::vector<std::unique_ptr<Light>>& lights, const XMVECTOR& viewOrigin) const;
    virtual XMVECTOR Emit(const Payload& payload) const = 0;
    virtual XMVECTOR GetAlbedo(XMFLOAT2 uv) const = 0;
    virtual XMVECTOR GetReflectance(XMFLOAT2 uv) const = 0;
    virtual XMVECTOR GetSmoothness(XMFLOAT2 uv) const = 0;
    virtual rt_material Describe() const = 0;
    // The per-path draw stream replaces the reference's per-material counters (material.h:34,50-51,67);
    // a single-call Scatter takes its (up to three) uniforms from here.
    static void SetScatterDraws(float u0, float u1, float u2);
};
class Metal : public Material {  // material.h:21-35
public:
    Metal(const Texture* reflectance, const XMVECTOR& smoothness);
    XMVECTOR Emit(const Payload&) const override { return XM_Zero; }
    XMVECTOR GetAlbedo(XMFLOAT2) const override { return XM_Zero; }
    XMVECTOR GetReflectance(XMFLOAT2 uv) const override { return m_reflectance->Evaluate(uv); }
    XMVECTOR GetSmoothness(XMFLOAT2) const override { return m_smoothness; }
    rt_material Describe() const override;
private:
    const Texture* m_reflectance;
    XMVECTOR m_smoothness;
};
class DielectricOpaque : public Material {  // material.h:37-52
public:
    DielectricOpaque(const Texture* albedo, const XMVECTOR& smoothness);
    XMVECTOR Emit(const Payload&) const override { return XM_Zero; }
    XMVECTOR GetAlbedo(XMFLOAT2 uv) const override { return m_albedo->Evaluate(uv); }
    XMVECTOR GetReflectance(XMFLOAT2) const override { return XMVECTORF32{0.04f, 0.04f, 0.04f, 1.f}; }
    XMVECTOR GetSmoothness(XMFLOAT2) const override { return m_smoothness; }
    rt_material Describe() const override;
private:
    const Texture* m_albedo;
    XMVECTOR m_smoothness;
};
class DielectricTransparent : public Material {  // material.h:54-68
public:
    DielectricTransparent(const XMVECTOR& smoothness, float ior);
    XMVECTOR Emit(const Payload&) const override { return XM_Zero; }
    XMVECTOR GetAlbedo(XMFLOAT2) const override { return XM_Zero; }
    XMVECTOR GetReflectance(XMFLOAT2) const override { return XMVECTORF32{0.04f, 0.04f, 0.04f, 1.f}; }
    XMVECTOR GetSmoothness(XMFLOAT2) const override { return m_smoothness; }
    rt_material Describe() const override;
private:
    XMVECTOR m_smoothness;
    XMVECTOR m_ior;
};
class Emissive : public Material {  // material.h:70-84
public:
    Emissive(const float luminance, const Texture* color);
    XMVECTOR Emit(const Payload& payload) const override;
    bool Scatter(const Ray&, const Payload&, XMVECTOR&, Ray&) const override { return false; }
    XMVECTOR GetAlbedo(XMFLOAT2) const override { return XM_Zero; }
    XMVECTOR GetReflectance(XMFLOAT2) const override { return XM_Zero; }
    XMVECTOR GetSmoothness(XMFLOAT2) const override { return XM_Zero; }
    rt_material Describe() const override;
private:
    const Texture* m_color;
    float m_luminance;
};

// ----------------------------------------------------------------------- camera.h
class Camera {  // camera.h:6-20
public:
    Camera(XMVECTOR origin, XMVECTOR lookAt, float verticalFOV, float aspectRatio, float focalLength, float aperture);
    Ray GetRay(XMFLOAT2 uv, XMFLOAT2 offset) const;  // device evaluated
    XMVECTOR GetOrigin() const;
    rt_camera Describe() const;
private:
    XMVECTOR m_origin;
    XMVECTOR m_x;
    XMVECTOR m_y;
    XMVECTOR m_originImagePlane;
    float m_aperture;
    float m_focalLength;
};

// ----------------------------------------------------------------- quasi-random.h
namespace Random {  // quasi-random.h:5-14 — batch-of-one device evaluation
float HaltonSample(uint64_t sampleIndex, uint32_t base);
XMFLOAT2 HaltonSample2D(uint64_t sampleIndex, uint32_t base1, uint32_t base2);
XMFLOAT2 HaltonSampleRing(uint64_t sampleIndex, uint32_t base);
XMFLOAT2 HaltonSampleDisk(uint64_t sampleIndex, uint32_t base1, uint32_t base2);
XMFLOAT3 HaltonSampleHemisphere(uint64_t sampleIndex, uint32_t base1, uint32_t base2);
uint64_t Xorshift();  // quasi-random.cpp:65-76 (dead code in the reference; kept for the symbol)
}  // namespace Random

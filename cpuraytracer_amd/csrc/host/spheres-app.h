// spheres-app.h — headless mirror of src/spheres/spheres-app.h + common-lib/app.h for MI355X.
// Same hooks (OnInitialize / OnRender / GetBackBufferWidth / GetBackBufferHeight, app.h:12-15) and
// the same members minus the Win32/Direct2D ones; DrawBitmap's for_each(par) and transform(par)
// (spheres-app.cpp:177-184,196-214) are one rt_render + rt_resolve call into librt_hip.so per frame.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "rt_host.h"

// spheres-app.h:5-13 — constexpr in the reference; runtime here so the CLI can set them
// (defaults are the reference's values).
struct AppSettingsT {
    int k_backbufferWidth = 1280;
    int k_backbufferHeight = 720;
    int k_recursionDepth = 50;
    float k_verticalFov = 25.f;
    float k_aperture = 0.4f;
    float AspectRatio() const { return k_backbufferWidth / static_cast<float>(k_backbufferHeight); }
    // additions (the reference has no CLI)
    std::string scene = "cover";  // cover | three | grid10k
    uint64_t sceneSeed = 1;       // replaces std::random_device (spheres-app.cpp:53-54)
    uint64_t renderSeed = 1;      // per-path xoshiro stream seed
    uint32_t samplesPerFrame = 1; // the reference adds one sample per frame (spheres-app.cpp:168)
    uint32_t framesInFlight = 0;  // rt_set_frame_pipelining: frames whose unfinished paths may ride along into later frames (0 = off)
    uint32_t framesPerLaunch = 1; // rt_set_frame_batch: quiet frames rendered per launch (1 = off)
    uint32_t samplerFlags = 0;    // RT_SAMPLER_* (rt_api.h): 0 = the reference's uniform hemisphere and linear-r lens disk
};

class RayTracingApp {  // app.h:5-30 without the window
public:
    virtual ~RayTracingApp() = default;
    virtual void Initialize(int deviceOrdinal);
    virtual int Run(uint32_t frames) noexcept;  // app.cpp:56-76: OnRender whenever idle, here `frames` times
    virtual void BeforeFrame(uint32_t frame, uint32_t frames) {}

protected:
    virtual void OnInitialize() = 0;
    virtual void OnRender() = 0;
    virtual int GetBackBufferWidth() const = 0;
    virtual int GetBackBufferHeight() const = 0;
    void InitBuffers();  // app.cpp:112-119

    std::vector<XMVECTOR> m_backbufferHdr;  // app.h:26
    std::vector<XMCOLOR> m_backbufferLdr;   // app.h:27
    rt_ctx* m_device = nullptr;
    int m_deviceOrdinal = 0;
};

class SpheresApp : public RayTracingApp {  // spheres-app.h:15-43
public:
    explicit SpheresApp(const AppSettingsT& settings) : AppSettings(settings) {}
    ~SpheresApp() override;

    // flat scene as uploaded (for tools / bindings)
    void DescribeScene(std::vector<rt_sphere>& spheres, std::vector<rt_material>& materials, rt_camera& camera, std::vector<rt_light>& lights,
                       rt_material& sky, float& exposureScale) const;
    bool WritePPM(const std::string& path) const;
    const std::vector<XMVECTOR>& Hdr() const { return m_backbufferHdr; }
    const std::vector<XMCOLOR>& Ldr() const { return m_backbufferLdr; }
    size_t SampleCount() const { return m_sampleCount; }
    double TotalSeconds() const { return m_totalSeconds; }
    const rt_stats& LastStats() const { return m_lastStats; }
    void SetRowset(rt_rowset rs) { m_rowset = rs; m_hasRowset = true; }
    void SetQuiet(bool q) { m_quiet = q; }
    void SetLastFrame(bool l) { m_lastFrame = l; }

    void BeforeFrame(uint32_t frame, uint32_t frames) override { m_lastFrame = frame + 1 == frames; }
    void OnInitialize() override;  // public so tools can build the scene without a device
    void OnRender() override;
    int GetBackBufferWidth() const override;
    int GetBackBufferHeight() const override;

private:
    void InitScene();
    void InitCamera();
    size_t DrawBitmap();
    void DisplayStats(size_t rayCount, double timeElapsed) const;

    AppSettingsT AppSettings;
    std::unique_ptr<Camera> m_camera;
    std::vector<std::unique_ptr<Hitable>> m_scene;
    std::vector<std::unique_ptr<Texture>> m_textures;
    std::vector<std::unique_ptr<Light>> m_lights;
    std::unique_ptr<BvhNode> m_bvh;
    std::unique_ptr<Material> m_skyMaterial;
    float m_exposure = 0.f;
    size_t m_sampleCount = 0;
    bool m_uploaded = false;
    rt_rowset m_rowset{};
    bool m_hasRowset = false;
    bool m_quiet = false;
    bool m_lastFrame = true;
    mutable double m_totalSeconds = 0.0;
    rt_stats m_lastStats{};
};

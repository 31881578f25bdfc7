// spheres-app.cpp — headless SpheresApp for MI355X (mirror of src/spheres/spheres-app.cpp).
#include "spheres-app.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <stdexcept>

namespace {
[[noreturn]] void ThrowRt(const char* where) { throw std::runtime_error(std::string(where) + ": " + rt_last_error()); }
#define RT_CALL(expr)                         \
    do {                                      \
        if ((expr) != RT_OK) ThrowRt(#expr);  \
    } while (0)
}  // namespace

// ------------------------------------------------------------------ RayTracingApp (app.cpp)
void RayTracingApp::Initialize(int deviceOrdinal) {  // app.cpp:16-54 minus RegisterClass/CreateWindow/InitDirect2D
    m_deviceOrdinal = deviceOrdinal;
    RT_CALL(rt_create(deviceOrdinal, &m_device));  // fails loudly without a GPU: there is no CPU render path
    InitBuffers();
    OnInitialize();
}
void RayTracingApp::InitBuffers() {  // app.cpp:112-119
    const size_t n = (size_t)GetBackBufferWidth() * GetBackBufferHeight();
    m_backbufferHdr.assign(n, XMVectorZero());
    m_backbufferLdr.assign(n, XMCOLOR(0.f, 0.f, 0.f, 0.f));
}
int RayTracingApp::Run(uint32_t frames) noexcept {  // app.cpp:56-76
    try {
        for (uint32_t f = 0; f < frames; ++f) {
            BeforeFrame(f, frames);
            OnRender();
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "spheres: %s\n", e.what());
        return 1;
    }
    return 0;
}

// ------------------------------------------------------------------------------ SpheresApp
SpheresApp::~SpheresApp() {
    if (m_device) rt_destroy(m_device);
}

void SpheresApp::OnInitialize() {  // spheres-app.cpp:4-8
    InitCamera();
    InitScene();
}

void SpheresApp::InitCamera() {  // spheres-app.cpp:35-49
    if (AppSettings.scene == "three") {  // SURVEY.md §8(d) C1 camera
        m_camera = std::make_unique<Camera>(XMVectorSet(0.f, 0.f, 0.f, 1.f), XMVectorSet(0.f, 0.f, 1.f, 1.f), AppSettings.k_verticalFov,
                                            AppSettings.AspectRatio(), 1.f, AppSettings.k_aperture);
    } else {
        XMVECTOR camOrigin = XMVectorSet(12.f, 2.f, -2.5f, 1.f);
        XMVECTOR camLookAt = XMVectorSet(0, 1, 0, 1.f);
        m_camera = std::make_unique<Camera>(camOrigin, camLookAt, AppSettings.k_verticalFov, AppSettings.AspectRatio(),
                                            XMVectorGetX(XMVector3Length(camOrigin - camLookAt)), AppSettings.k_aperture);
    }
    m_exposure = -15;
}

void SpheresApp::InitScene() {  // spheres-app.cpp:51-130
    // std::random_device seeding (:53-54) replaced by a fixed seed; uniform_real_distribution
    // (implementation defined) replaced by engine() * 2^-24.
    std::ranlux24_base generator((std::ranlux24_base::result_type)AppSettings.sceneSeed);
    auto uniformDist = [&generator]() -> float { return (float)generator() * 0x1p-24f; };

    m_scene.clear();
    m_textures.clear();
    m_lights.clear();
    m_scene.reserve(500);

    if (AppSettings.scene == "three") {
        const float cols[3][3] = {{0.5f, 0.5f, 0.5f}, {0.8f, 0.3f, 0.3f}, {0.8f, 0.8f, 0.0f}};
        const float geo[3][4] = {{0.f, 0.f, 1.f, 0.5f}, {1.f, 0.f, 1.f, 0.5f}, {0.f, -100.5f, 1.f, 100.f}};
        for (int k = 0; k < 3; ++k) {
            m_textures.push_back(std::make_unique<ConstTexture>(XMCOLOR{cols[k][0], cols[k][1], cols[k][2], 1.f}));
            m_scene.push_back(std::make_unique<Sphere>(XMVECTORF32{geo[k][0], geo[k][1], geo[k][2], 0.f}, geo[k][3],
                                                       std::make_unique<DielectricOpaque>(m_textures.back().get(), XMVectorReplicate(16.f))));
        }
    } else {
        const int lo = AppSettings.scene == "grid10k" ? -50 : -11;
        const int hi = AppSettings.scene == "grid10k" ? 50 : 11;
        // Floor
        m_textures.push_back(std::make_unique<CheckerTexture>(XMCOLOR{0.9f, 0.9f, 0.9f, 1.f}, XMCOLOR{0.2f, 0.3f, 0.1f, 1.f}, 2500.f));
        m_scene.push_back(std::make_unique<Sphere>(XMVECTORF32{0, -1000, 0, 0}, 1000.f,
                                                   std::make_unique<DielectricOpaque>(m_textures.back().get(), XMVectorReplicate(16.f))));
        // Random small spheres
        for (int a = lo; a < hi; ++a) {
            for (int b = lo; b < hi; ++b) {
                const float chooseMat = uniformDist();
                const float cx = a + 0.9f * uniformDist();
                const float cz = b + 0.9f * uniformDist();
                XMVECTORF32 center{cx, 0.2f, cz, 0.f};
                if (chooseMat < 0.8f) {
                    const float r0 = uniformDist(), r1 = uniformDist(), g0 = uniformDist(), g1 = uniformDist(), b0 = uniformDist(), b1 = uniformDist();
                    m_textures.push_back(std::make_unique<ConstTexture>(XMCOLOR{r0 * r1, g0 * g1, b0 * b1, 1.f}));
                    float smoothness = 8.f * (4.f + uniformDist());
                    m_scene.push_back(std::make_unique<Sphere>(center, 0.2f, std::make_unique<DielectricOpaque>(m_textures.back().get(), XMVectorReplicate(smoothness))));
                } else if (chooseMat < 0.95f) {
                    const float r = uniformDist(), g = uniformDist(), bl = uniformDist();
                    m_textures.push_back(std::make_unique<ConstTexture>(XMCOLOR{0.5f * (1.f + r), 0.5f * (1.f + g), 0.5f * (1.f + bl), 1.f}));
                    m_scene.push_back(std::make_unique<Sphere>(center, 0.2f, std::make_unique<Metal>(m_textures.back().get(), XM_Zero)));
                } else {
                    float smoothness = 8.f * (4.f + uniformDist());
                    m_scene.emplace_back(std::make_unique<Sphere>(center, 0.2f, std::make_unique<DielectricTransparent>(XMVectorReplicate(smoothness), 1.5f)));
                }
            }
        }
        // Large spheres
        m_scene.push_back(std::make_unique<Sphere>(XMVECTORF32{0, 1, 0, 0}, 1.f, std::make_unique<DielectricTransparent>(XMVectorReplicate(16.f), 1.5f)));
        m_textures.push_back(std::make_unique<ConstTexture>(XMCOLOR{0.4f, 0.2f, 0.1f, 1.f}));
        m_scene.push_back(std::make_unique<Sphere>(XMVECTORF32{-4, 1, 0, 0}, 1.f, std::make_unique<DielectricOpaque>(m_textures.back().get(), XMVectorReplicate(16.f))));
        m_textures.push_back(std::make_unique<ConstTexture>(XMCOLOR{0.7f, 0.6f, 0.5f, 1.f}));
        m_scene.push_back(std::make_unique<Sphere>(XMVECTORF32{4, 1, 0, 0}, 1.f, std::make_unique<Metal>(m_textures.back().get(), XM_Zero)));
    }

    // Construct BVH (moves the unique_ptrs out of m_scene, as the reference does)
    m_bvh = std::make_unique<BvhNode>(m_scene.begin(), m_scene.end());

    // Sky
    m_textures.push_back(std::make_unique<ConstTexture>(XMCOLOR{0.85f, 0.91f, 0.98f, 1.f}));
    m_skyMaterial = std::make_unique<Emissive>(8000.f, m_textures.back().get());

    // Sun
    auto lightOcclusionTest = [this](const Ray& ray) -> bool {
        Payload dummy{};
        return m_bvh->Intersect(ray, dummy);
    };
    m_lights.push_back(std::make_unique<DirectionalLight>(XMVECTORF32{1.f, 1.f, 1.f, 0.f}, XMCOLOR{1.f, 0.97f, 0.88f, 1.f}, 40000.f, lightOcclusionTest));
    m_uploaded = false;
}

void SpheresApp::DescribeScene(std::vector<rt_sphere>& spheres, std::vector<rt_material>& materials, rt_camera& camera,
                               std::vector<rt_light>& lights, rt_material& sky, float& exposureScale) const {
    spheres.clear();
    materials.clear();
    m_bvh->Flatten(spheres, materials);
    camera = m_camera->Describe();
    lights.clear();  // every light of the list, in list order (Material::Shade adds them in that order, material.cpp:4-13)
    for (const auto& l : m_lights) lights.push_back(l->Describe());
    sky = m_skyMaterial->Describe();
    exposureScale = static_cast<float>(std::pow(2, m_exposure));  // spheres-app.cpp:174
}

void SpheresApp::OnRender() {  // spheres-app.cpp:10-33 minus BeginPaint/BeginDraw
    const auto start = std::chrono::high_resolution_clock::now();
    const size_t rayCount = DrawBitmap();
    const auto stop = std::chrono::high_resolution_clock::now();
    const std::chrono::duration<double, std::micro> duration = stop - start;
    DisplayStats(rayCount, duration.count());
}

size_t SpheresApp::DrawBitmap() {  // spheres-app.cpp:163-222
    const uint32_t W = (uint32_t)GetBackBufferWidth(), H = (uint32_t)GetBackBufferHeight();
    if (!m_uploaded) {
        std::vector<rt_sphere> spheres;
        std::vector<rt_material> materials;
        rt_camera camera;
        std::vector<rt_light> lights;
        rt_material sky;
        float exposureAdjustment;
        DescribeScene(spheres, materials, camera, lights, sky, exposureAdjustment);
        RT_CALL(rt_scene_upload(m_device, spheres.data(), materials.data(), (uint32_t)spheres.size(), &camera, lights.data(), (uint32_t)lights.size(), &sky,
                                exposureAdjustment));
        RT_CALL(rt_set_sampler(m_device, AppSettings.samplerFlags));
        RT_CALL(rt_set_frame_pipelining(m_device, AppSettings.framesInFlight));
        RT_CALL(rt_set_frame_batch(m_device, AppSettings.framesPerLaunch ? AppSettings.framesPerLaunch : 1u));
        m_uploaded = true;
    }
    const rt_rowset rs = m_hasRowset ? m_rowset : rt_rowset{0, H, H, 0, 1};
    const uint32_t s0 = (uint32_t)m_sampleCount + 1;  // ++m_sampleCount (:168): first frame uses index 1
    const uint32_t s1 = s0 + AppSettings.samplesPerFrame;
    // GenerateRays + trace (:171-184) and the tonemap (:196-214) on the device
    // per-frame statistics cost a host wait; a quiet progressive run stays asynchronous until the last frame
    rt_stats* wantStats = (m_quiet && !m_lastFrame) ? nullptr : &m_lastStats;
    RT_CALL(rt_render(m_device, W, H, rs, s0, s1, (uint32_t)AppSettings.k_recursionDepth, AppSettings.renderSeed, wantStats));
    if (!wantStats) m_lastStats.local_rows = rt_rowset_local_rows(rs);
    m_sampleCount += AppSettings.samplesPerFrame;
    if (wantStats) {  // the tonemap feeds the display in the reference; headless and quiet, only the last frame needs it
        RT_CALL(rt_resolve(m_device, (uint32_t)m_sampleCount));
        m_lastStats.ms_resolve = rt_last_resolve_ms(m_device);
    }
    return (size_t)W * m_lastStats.local_rows * AppSettings.samplesPerFrame;
}

bool SpheresApp::WritePPM(const std::string& path) const {
    const uint32_t W = (uint32_t)GetBackBufferWidth();
    const uint32_t rows = m_lastStats.local_rows;
    std::vector<uint8_t> rgb((size_t)W * rows * 3);
    if (rt_download(m_device, nullptr, rgb.data()) != RT_OK) return false;
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%u %u\n255\n", W, rows);
    const bool ok = std::fwrite(rgb.data(), 1, rgb.size(), f) == rgb.size();
    std::fclose(f);
    return ok;
}

void SpheresApp::DisplayStats(const size_t rayCount, const double timeElapsed) const {  // spheres-app.cpp:259-272 -> stdout JSON
    m_totalSeconds += timeElapsed * std::pow(10, -6);
    const double mraysPerSecond = static_cast<double>(rayCount) / timeElapsed;
    if (m_quiet) return;
    std::printf("{\"Mrays_per_s\": %.3f, \"spp\": %zu, \"time_s\": %.6f, \"kernel_ms\": %.3f, \"traversals_per_sample\": %.4f}\n", mraysPerSecond,
                m_sampleCount, m_totalSeconds, m_lastStats.ms_render + m_lastStats.ms_accumulate + m_lastStats.ms_resolve,
                m_lastStats.samples ? (double)m_lastStats.traversals / (double)m_lastStats.samples : 0.0);
}

int SpheresApp::GetBackBufferWidth() const { return AppSettings.k_backbufferWidth; }    // spheres-app.cpp:274-277
int SpheresApp::GetBackBufferHeight() const { return AppSettings.k_backbufferHeight; }  // spheres-app.cpp:279-282

// main.cpp — headless replacement for src/spheres/main.cpp (WinMain, main.cpp:5-10): builds the
// app, renders `--spp` samples on MI355X through librt_hip.so and writes a PPM.  The reference has
// no command line; the flags below are new surface with AppSettings defaults.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "multi_gpu.h"
#include "spheres-app.h"

static void Usage() {
    std::puts("usage: spheres [--width N] [--height N] [--spp N] [--frame-spp N] [--depth N] [--fov F] [--aperture F]\n"
              "               [--scene cover|three|grid10k] [--scene-seed N] [--seed N] [--device N] [--gpus N] [--out file.ppm] [--quiet]\n"
              "               [--sampler reference|cosine|sqrtdisk|cosine+sqrtdisk]   (default: the reference's mappings)\n"
              "               [--pipeline N]   frames in flight for quiet progressive runs (--frame-spp 1 --quiet); 0 = off\n"
              "               [--batch N]      quiet progressive frames rendered per launch (rt_set_frame_batch); 1 = off");
}

int main(int argc, char** argv) {
    AppSettingsT st;
    uint32_t spp = 16;
    int device = 0, gpus = 0;
    std::string out = "out.ppm";
    bool quiet = false, fovSet = false, apSet = false;
    for (int a = 1; a < argc; ++a) {
        const std::string k = argv[a];
        auto val = [&]() -> const char* {
            if (a + 1 >= argc) { Usage(); std::exit(2); }
            return argv[++a];
        };
        if (k == "--width") st.k_backbufferWidth = std::atoi(val());
        else if (k == "--height") st.k_backbufferHeight = std::atoi(val());
        else if (k == "--spp") spp = (uint32_t)std::atoi(val());
        else if (k == "--frame-spp") st.samplesPerFrame = (uint32_t)std::atoi(val());
        else if (k == "--depth") st.k_recursionDepth = std::atoi(val());
        else if (k == "--fov") { st.k_verticalFov = (float)std::atof(val()); fovSet = true; }
        else if (k == "--aperture") { st.k_aperture = (float)std::atof(val()); apSet = true; }
        else if (k == "--scene") st.scene = val();
        else if (k == "--scene-seed") st.sceneSeed = std::strtoull(val(), nullptr, 10);
        else if (k == "--seed") st.renderSeed = std::strtoull(val(), nullptr, 10);
        else if (k == "--device") device = std::atoi(val());
        else if (k == "--gpus") gpus = std::atoi(val());
        else if (k == "--out") out = val();
        else if (k == "--quiet") quiet = true;
        else if (k == "--pipeline") st.framesInFlight = (uint32_t)std::atoi(val());
        else if (k == "--batch") st.framesPerLaunch = (uint32_t)std::atoi(val());
        else if (k == "--sampler") {
            const std::string v = val();
            st.samplerFlags = (v.find("cosine") != std::string::npos ? RT_SAMPLER_COSINE_HEMISPHERE : 0u) |
                              (v.find("sqrtdisk") != std::string::npos ? RT_SAMPLER_SQRT_DISK : 0u);
            if (st.samplerFlags == 0u && v != "reference") { Usage(); return 2; }
        }
        else { Usage(); return k == "--help" ? 0 : 2; }
    }
    if (st.scene == "three") {  // C1 defaults
        if (!fovSet) st.k_verticalFov = 90.f;
        if (!apSet) st.k_aperture = 0.f;
    }
    if (st.samplesPerFrame == 0 || spp == 0 || st.k_backbufferWidth <= 0 || st.k_backbufferHeight <= 0) { Usage(); return 2; }
    if (spp % st.samplesPerFrame != 0) st.samplesPerFrame = 1;
    if (gpus > 0) {  // rows sharded over `gpus` devices, RCCL gather to device 0
        MultiGpuResult res;
        std::string err;
        if (RenderMultiGpu(st, gpus, spp, res, &err) != 0) {
            std::fprintf(stderr, "spheres: %s\n", err.c_str());
            return 1;
        }
        FILE* f = std::fopen(out.c_str(), "wb");
        if (!f) {
            std::fprintf(stderr, "spheres: cannot write %s\n", out.c_str());
            return 1;
        }
        std::fprintf(f, "P6\n%d %d\n255\n", st.k_backbufferWidth, st.k_backbufferHeight);
        std::fwrite(res.ldr.data(), 1, res.ldr.size(), f);
        std::fclose(f);
        std::printf("{\"out\": \"%s\", \"gpus\": %d, \"spp\": %u, \"render_s\": %.4f, \"Msamples_per_s\": %.2f, \"traversals_per_sample\": %.4f}\n",
                    out.c_str(), gpus, spp, res.renderSeconds, (double)res.samples / res.renderSeconds / 1e6,
                    res.samples ? (double)res.traversals / (double)res.samples : 0.0);
        return 0;
    }
    SpheresApp app(st);
    app.SetQuiet(quiet);
    try {
        app.Initialize(device);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "spheres: %s\n", e.what());
        return 1;
    }
    const int rc = app.Run(spp / st.samplesPerFrame);
    if (rc != 0) return rc;
    if (!app.WritePPM(out)) {
        std::fprintf(stderr, "spheres: cannot write %s\n", out.c_str());
        return 1;
    }
    const rt_stats& ls = app.LastStats();
    std::printf("{\"out\": \"%s\", \"spp\": %zu, \"total_s\": %.4f, \"last_frame_kernel_ms\": %.3f}\n", out.c_str(), app.SampleCount(),
                app.TotalSeconds(), ls.ms_render + ls.ms_accumulate + ls.ms_resolve);
    return 0;
}

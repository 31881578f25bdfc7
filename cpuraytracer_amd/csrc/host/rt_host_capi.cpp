// rt_host_capi.cpp — small C surface over the host mirror for the Python plumbing (tests, bench.py):
// the product's own InitScene/InitCamera (SpheresApp) flattened into rt_api.h tables.
#include <cstring>
#include <string>

#include "spheres-app.h"

extern "C" {

// Same contract as the oracle's generator: cap = capacity; *n = count; returns 0 on success.
int rth_build_scene(const char* name, uint64_t seed, int width, int height, float vfov, float aperture, uint32_t cap, rt_sphere* spheres,
                    rt_material* materials, uint32_t* n, rt_camera* camera, rt_light* sun, rt_material* sky, float* exposure_scale) {
    try {
        AppSettingsT st;
        st.scene = name ? name : "cover";
        st.sceneSeed = seed;
        st.k_backbufferWidth = width;
        st.k_backbufferHeight = height;
        if (vfov > 0.f) st.k_verticalFov = vfov;
        else if (st.scene == "three") st.k_verticalFov = 90.f;
        if (aperture >= 0.f) st.k_aperture = aperture;
        else if (st.scene == "three") st.k_aperture = 0.f;
        if (st.scene != "cover" && st.scene != "three" && st.scene != "grid10k") return 2;
        SpheresApp app(st);
        app.OnInitialize();
        std::vector<rt_sphere> s;
        std::vector<rt_material> m;
        rt_camera c;
        std::vector<rt_light> ls;
        rt_material k;
        float e;
        app.DescribeScene(s, m, c, ls, k, e);
        const rt_light l = ls.empty() ? rt_light{} : ls.front();  // (the generators' scenes have one light, spheres-app.cpp:129)
        if (n) *n = (uint32_t)s.size();
        if (s.size() > cap) return 1;
        if (spheres) std::memcpy(spheres, s.data(), s.size() * sizeof(rt_sphere));
        if (materials) std::memcpy(materials, m.data(), m.size() * sizeof(rt_material));
        if (camera) *camera = c;
        if (sun) *sun = l;
        if (sky) *sky = k;
        if (exposure_scale) *exposure_scale = e;
        return 0;
    } catch (...) {
        return 3;
    }
}

}  // extern "C"

// api_selftest.cpp — exercises the host mirror exactly the way code written against the reference's headers would
// (common-lib/ray-tracing.h, material.h, texture.h, light.h, camera.h, quasi-random.h) and prints the results as one
// JSON object; tests/test_gpu_parity.py::test_host_mirror_api compares them with the oracle.  Every per-ray call below
// runs on the GPU through librt_hip.so.
#include <cstdio>
#include <memory>
#include <vector>

#include "rt_host.h"

static void P3(const char* k, XMVECTOR v, bool comma = true) { std::printf("\"%s\": [%.9g, %.9g, %.9g]%s", k, v.x, v.y, v.z, comma ? ", " : ""); }

int main() {
    try {
        std::printf("{");
        // quasi-random.h
        std::printf("\"halton_100_3\": %.9g, ", Random::HaltonSample(100, 3));
        const XMFLOAT2 dk = Random::HaltonSampleDisk(7, 4, 5);
        std::printf("\"disk_7\": [%.9g, %.9g], ", dk.x, dk.y);
        const XMFLOAT3 hs = Random::HaltonSampleHemisphere(3, 5, 7);
        std::printf("\"hemi_3\": [%.9g, %.9g, %.9g], ", hs.x, hs.y, hs.z);

        // camera.h — InitCamera's camera (spheres-app.cpp:35-49) at 1200x800
        const XMVECTOR camOrigin = XMVectorSet(12.f, 2.f, -2.5f, 1.f), camLookAt = XMVectorSet(0, 1, 0, 1.f);
        Camera camera(camOrigin, camLookAt, 25.f, 1.5f, XMVectorGetX(XMVector3Length(camOrigin - camLookAt)), 0.4f);
        const Ray pr = camera.GetRay(XMFLOAT2(0.3f, 0.6f), XMFLOAT2(0.1f, -0.2f));
        P3("cam_ray_o", pr.origin);
        P3("cam_ray_d", pr.direction);

        // texture.h / material.h / ray-tracing.h — the C1 scene (three DielectricOpaque spheres)
        std::vector<std::unique_ptr<Texture>> textures;
        std::vector<std::unique_ptr<Hitable>> scene;
        const float cols[3][3] = {{0.5f, 0.5f, 0.5f}, {0.8f, 0.3f, 0.3f}, {0.8f, 0.8f, 0.0f}};
        const float geo[3][4] = {{0.f, 0.f, 1.f, 0.5f}, {1.f, 0.f, 1.f, 0.5f}, {0.f, -100.5f, 1.f, 100.f}};
        for (int k = 0; k < 3; ++k) {
            textures.push_back(std::make_unique<ConstTexture>(XMCOLOR{cols[k][0], cols[k][1], cols[k][2], 1.f}));
            scene.push_back(std::make_unique<Sphere>(XMVECTORF32{geo[k][0], geo[k][1], geo[k][2], 0.f}, geo[k][3],
                                                     std::make_unique<DielectricOpaque>(textures.back().get(), XMVectorReplicate(16.f))));
        }
        // one sphere on its own
        Payload hit{};
        const Ray ray{XMVectorSet(0.1f, 0.05f, 0.f, 1.f), XMVector3Normalize(XMVectorSet(0.05f, -0.02f, 1.f, 0.f))};
        const bool hitOne = scene[0]->Intersect(ray, hit);
        std::printf("\"sphere_hit\": %d, \"sphere_t\": %.9g, ", hitOne ? 1 : 0, hit.t.x);
        P3("sphere_pos", hit.pos);
        P3("sphere_normal", hit.normal);
        std::printf("\"sphere_uv\": [%.9g, %.9g], ", hit.uv.x, hit.uv.y);

        // Scatter with given draws, then Shade with an unoccluded and an occluded sun
        Material::SetScatterDraws(0.5f, 0.3f, 0.7f);
        XMVECTOR attenuation;
        Ray scattered;
        const bool sc = hit.material->Scatter(ray, hit, attenuation, scattered);
        std::printf("\"scattered\": %d, ", sc ? 1 : 0);
        P3("attenuation", attenuation);
        P3("scatter_dir", scattered.direction);

        // BvhNode takes ownership by moving out of the vector (ray-tracing.cpp:113,118-119)
        BvhNode bvh(scene.begin(), scene.end());
        std::printf("\"scene_moved_out\": %d, ", (scene[0] == nullptr && scene[2] == nullptr) ? 1 : 0);
        std::vector<std::unique_ptr<Light>> lights;
        lights.push_back(std::make_unique<DirectionalLight>(XMVECTORF32{1.f, 1.f, 1.f, 0.f}, XMCOLOR{1.f, 0.97f, 0.88f, 1.f}, 40000.f,
                                                            [&bvh](const Ray& r) { Payload d{}; return bvh.Intersect(r, d); }));
        P3("shade", hit.material->Shade(hit, lights, camera.GetOrigin()));
        Payload floorHit{};
        const Ray down{XMVectorSet(0.45f, 1.f, 0.55f, 1.f), XMVectorSet(0.f, -1.f, 0.f, 0.f)};
        const bool hitBvh = bvh.Intersect(down, floorHit);
        std::printf("\"bvh_hit\": %d, \"bvh_t\": %.9g, ", hitBvh ? 1 : 0, floorHit.t.x);
        P3("bvh_normal", floorHit.normal);
        // a point on the floor in the shadow of sphere 0 (sun direction (1,1,1)): Shade must return 0
        Payload shadowed{};
        const Ray toShadow{XMVectorSet(-0.35f, 1.f, 0.65f, 1.f), XMVectorSet(0.f, -1.f, 0.f, 0.f)};
        bvh.Intersect(toShadow, shadowed);
        P3("shade_in_shadow", shadowed.material->Shade(shadowed, lights, camera.GetOrigin()), false);
        std::printf("}\n");
        DeviceEval::Shutdown();
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "api_selftest: %s\n", e.what());
        return 1;
    }
}

// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-alone driver for sanitizer runs (AddressSanitizer + UBSan on the CPU
// build; GPU sanitizers are unavailable on the pool): exercises scene generation, list and BVH traversal, every
// material, threads, sharding, resolve.  Exit code 0 = ran clean and the two accelerators agreed.
#include <cstdio>
#include <cstring>
#include <vector>

#include "oracle_api.h"

int main() {
    const uint32_t cap = 10100;
    std::vector<rt_sphere> sp(cap);
    std::vector<rt_material> mt(cap);
    uint32_t n = 0;
    rt_camera cam;
    rt_light sun;
    rt_material sky;
    float exposure = 0.f;
    int bad = 0;
    for (const char* name : {"three", "cover", "grid10k"}) {
        if (orc_build_scene(name, 1, 1.5f, -1.f, cap, sp.data(), mt.data(), &n, &cam, &sun, &sky, &exposure) != 0) return 2;
        orc_ctx* ctx = nullptr;
        if (orc_create(&ctx) != 0) return 2;
        if (orc_scene_upload(ctx, sp.data(), mt.data(), n, &cam, &sun, 1u, &sky, exposure) != 0) return 2;
        const uint32_t W = 48, H = 30;
        const rt_rowset whole{0, H, H, 0, 1};
        rt_stats a{}, b{};
        std::vector<float> h1(W * H * 3), h2(W * H * 3);
        std::vector<uint8_t> l1(W * H * 3);
        if (orc_render(ctx, W, H, whole, 1, 3, 50, 1, ORC_ACCEL_BVH, 3, &a) != 0) return 2;
        orc_resolve(ctx, 0);
        orc_download(ctx, h1.data(), l1.data());
        if (std::strcmp(name, "grid10k") != 0) {  // the exhaustive list scan of 10k spheres is slow under ASan
            if (orc_render(ctx, W, H, whole, 1, 3, 50, 1, ORC_ACCEL_LIST, 1, &b) != 0) return 2;
            orc_download(ctx, h2.data(), nullptr);
            if (std::memcmp(h1.data(), h2.data(), h1.size() * sizeof(float)) != 0 || a.traversals != b.traversals) {
                std::fprintf(stderr, "%s: BVH and list scan disagree\n", name);
                bad = 1;
            }
        }
        const rt_rowset shard{0, H, 4, 1, 3};
        if (orc_render(ctx, W, H, shard, 1, 2, 8, 7, ORC_ACCEL_BVH, 2, &a) != 0) return 2;
        std::printf("%s: n=%u traversals=%llu shard rows=%u\n", name, n, (unsigned long long)a.traversals, a.local_rows);
        orc_destroy(ctx);
    }
    float o2[2], o3[3];
    orc_halton_disk(12345, 4, 5, o2);
    orc_halton_hemisphere(54321, 5, 7, o3);
    return bad;
}

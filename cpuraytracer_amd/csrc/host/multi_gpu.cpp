// multi_gpu.cpp — see multi_gpu.h.  Host C++17 + the HIP runtime API + RCCL (no kernels here).
#include "multi_gpu.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <thread>

namespace {

constexpr uint32_t kBlockRows = 1;  // single rows: the slowest rank is 0.5 % above the mean (4-row blocks: 2 %)

uint32_t LocalRows(uint32_t H, uint32_t rank, uint32_t world) {
    const uint32_t nblocks = (H + kBlockRows - 1) / kBlockRows;
    uint32_t rows = 0;
    for (uint32_t b = rank; b < nblocks; b += world) rows += std::min(kBlockRows, H - b * kBlockRows);
    return rows;
}
uint32_t GlobalRow(uint32_t lr, uint32_t rank, uint32_t world) { return ((lr / kBlockRows) * world + rank) * kBlockRows + lr % kBlockRows; }

struct Rank {
    int device = 0;
    rt_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    float* hdrStrip = nullptr;    // [rowsMax * W * 3] padded to the largest shard: one equal-count gather
    uint8_t* ldrStrip = nullptr;
    rt_stats stats{};
    std::string error;
};

#define HIP_OK(call)                                                                  \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            rk.error = std::string(#call) + ": " + hipGetErrorString(e_);             \
            return;                                                                   \
        }                                                                             \
    } while (0)

}  // namespace

int RenderMultiGpu(const AppSettingsT& st, int nGpus, uint32_t spp, MultiGpuResult& out, std::string* err) {
    auto fail = [&](const std::string& m) {
        if (err) *err = m;
        return 1;
    };
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail("no HIP device; this program has no CPU render path");
    if (nGpus < 1 || nGpus > count) return fail("--gpus exceeds the " + std::to_string(count) + " visible device(s)");
    const uint32_t W = (uint32_t)st.k_backbufferWidth, H = (uint32_t)st.k_backbufferHeight, G = (uint32_t)nGpus;
    uint32_t rowsMax = 0;
    for (uint32_t r = 0; r < G; ++r) rowsMax = std::max(rowsMax, LocalRows(H, r, G));
    const size_t stripPix = (size_t)rowsMax * W;

    // the scene is built once on the host and uploaded to every device (<= 0.5 MB even for 10k spheres)
    SpheresApp builder(st);
    builder.OnInitialize();
    std::vector<rt_sphere> spheres;
    std::vector<rt_material> materials;
    rt_camera camera;
    std::vector<rt_light> lights;
    rt_material sky;
    float exposure;
    builder.DescribeScene(spheres, materials, camera, lights, sky, exposure);

    std::vector<Rank> ranks(G);
    std::vector<ncclComm_t> comms(G);
    std::vector<int> devs(G);
    for (uint32_t r = 0; r < G; ++r) devs[r] = (int)r;
    if (ncclCommInitAll(comms.data(), (int)G, devs.data()) != ncclSuccess) return fail("ncclCommInitAll failed");

    float* hdrAll = nullptr;  // device 0: [G][rowsMax * W * 3]
    uint8_t* ldrAll = nullptr;

    // ---- set up every device
    for (uint32_t r = 0; r < G; ++r) {
        Rank& rk = ranks[r];
        rk.device = (int)r;
        [&]() {
            HIP_OK(hipSetDevice(rk.device));
            HIP_OK(hipStreamCreateWithFlags(&rk.stream, hipStreamNonBlocking));
            HIP_OK(hipMalloc(reinterpret_cast<void**>(&rk.hdrStrip), stripPix * 3 * sizeof(float)));
            HIP_OK(hipMalloc(reinterpret_cast<void**>(&rk.ldrStrip), stripPix * 3));
            HIP_OK(hipMemset(rk.hdrStrip, 0, stripPix * 3 * sizeof(float)));
            HIP_OK(hipMemset(rk.ldrStrip, 0, stripPix * 3));
            if (rt_create(rk.device, &rk.ctx) != RT_OK || rt_set_stream(rk.ctx, rk.stream) != RT_OK ||
                rt_scene_upload(rk.ctx, spheres.data(), materials.data(), (uint32_t)spheres.size(), &camera, lights.data(), (uint32_t)lights.size(), &sky, exposure) != RT_OK ||
                rt_set_sampler(rk.ctx, st.samplerFlags) != RT_OK)
                rk.error = rt_last_error();
        }();
        if (!rk.error.empty()) return fail("device " + std::to_string(r) + ": " + rk.error);
    }
    if (hipSetDevice(0) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&hdrAll), (size_t)G * stripPix * 3 * sizeof(float)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&ldrAll), (size_t)G * stripPix * 3) != hipSuccess)
        return fail("gather buffers on device 0: out of memory");

    // ---- render: one thread per device; the gather is one grouped collective issued from this thread
    const auto t0 = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> pool;
        for (uint32_t r = 0; r < G; ++r) {
            pool.emplace_back([&, r]() {
                Rank& rk = ranks[r];
                if (hipSetDevice(rk.device) != hipSuccess) {
                    rk.error = "hipSetDevice failed";
                    return;
                }
                const rt_rowset rs{0, H, kBlockRows, r, G};
                if (rt_render(rk.ctx, W, H, rs, 1, 1 + spp, (uint32_t)st.k_recursionDepth, st.renderSeed, &rk.stats) != RT_OK ||
                    rt_resolve(rk.ctx, spp) != RT_OK || rt_copy_to_device(rk.ctx, rk.hdrStrip, rk.ldrStrip) != RT_OK)
                    rk.error = rt_last_error();
            });
        }
        for (auto& th : pool) th.join();
    }
    for (uint32_t r = 0; r < G; ++r)
        if (!ranks[r].error.empty()) return fail("device " + std::to_string(r) + ": " + ranks[r].error);

    // RCCL gather of the strips to rank 0 (equal counts: strips are padded to the largest shard)
    bool ncclOk = ncclGroupStart() == ncclSuccess;
    for (uint32_t r = 0; r < G && ncclOk; ++r) {
        ncclOk = ncclGather(ranks[r].hdrStrip, hdrAll, stripPix * 3, ncclFloat, 0, comms[r], ranks[r].stream) == ncclSuccess &&
                 ncclGather(ranks[r].ldrStrip, ldrAll, stripPix * 3, ncclUint8, 0, comms[r], ranks[r].stream) == ncclSuccess;
    }
    ncclOk = ncclOk && ncclGroupEnd() == ncclSuccess;
    if (!ncclOk) return fail("ncclGather failed");
    for (uint32_t r = 0; r < G; ++r) {
        if (hipSetDevice(ranks[r].device) != hipSuccess || hipStreamSynchronize(ranks[r].stream) != hipSuccess) return fail("stream sync failed");
    }
    out.renderSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    // ---- de-interleave on the host (outside the timed region, like the PPM write)
    std::vector<float> hdrHost((size_t)G * stripPix * 3);
    std::vector<uint8_t> ldrHost((size_t)G * stripPix * 3);
    if (hipSetDevice(0) != hipSuccess || hipMemcpy(hdrHost.data(), hdrAll, hdrHost.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(ldrHost.data(), ldrAll, ldrHost.size(), hipMemcpyDeviceToHost) != hipSuccess)
        return fail("download of the gathered image failed");
    out.hdr.assign((size_t)W * H * 3, 0.f);
    out.ldr.assign((size_t)W * H * 3, 0);
    out.traversals = out.samples = 0;
    for (uint32_t r = 0; r < G; ++r) {
        const uint32_t rows = LocalRows(H, r, G);
        for (uint32_t lr = 0; lr < rows; ++lr) {
            const uint32_t j = GlobalRow(lr, r, G);
            std::memcpy(&out.hdr[(size_t)j * W * 3], &hdrHost[((size_t)r * stripPix + (size_t)lr * W) * 3], (size_t)W * 3 * sizeof(float));
            std::memcpy(&out.ldr[(size_t)j * W * 3], &ldrHost[((size_t)r * stripPix + (size_t)lr * W) * 3], (size_t)W * 3);
        }
        out.traversals += ranks[r].stats.traversals;
        out.samples += ranks[r].stats.samples;
    }
    for (uint32_t r = 0; r < G; ++r) {
        (void)hipSetDevice(ranks[r].device);
        rt_destroy(ranks[r].ctx);
        (void)hipFree(ranks[r].hdrStrip);
        (void)hipFree(ranks[r].ldrStrip);
        (void)hipStreamDestroy(ranks[r].stream);
        ncclCommDestroy(comms[r]);
    }
    (void)hipSetDevice(0);
    (void)hipFree(hdrAll);
    (void)hipFree(ldrAll);
    return 0;
}

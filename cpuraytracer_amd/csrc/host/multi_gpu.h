// multi_gpu.h — single-process, one-thread-per-GPU driver for the headless app: rank r renders the
// 4-row blocks b = r (mod G) of the same image on device r; one RCCL gather over xGMI moves the finished strips to
// device 0, which de-interleaves them.  (bench.py does the same with one PROCESS per GPU through torch.distributed.)
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "spheres-app.h"

struct MultiGpuResult {
    std::vector<uint8_t> ldr;  // H x W x 3, assembled on the host from device 0's gather buffer
    std::vector<float> hdr;    // H x W x 3
    double renderSeconds = 0;  // barrier-to-barrier wall time of render + resolve + gather (max over devices)
    uint64_t traversals = 0, samples = 0;
};

// Returns 0 on success; message in *err otherwise.  nGpus must not exceed hipGetDeviceCount().
int RenderMultiGpu(const AppSettingsT& settings, int nGpus, uint32_t spp, MultiGpuResult& out, std::string* err);

// MFMA issue-rate microbenchmark for gfx950: f32 32x32x2 (the filter's first form) vs bf16 32x32x16 vs f32 16x16x4.
// Reports SIMD cycles per MFMA instruction with 4 independent accumulator chains per wave, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 256;

template <int MODE>
__global__ void __launch_bounds__(256) k_rate(float* out, float sg, unsigned long long* clk) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    f32x4 q0 = {0}, q1 = {0}, q2 = {0}, q3 = {0};
    const float a = sg + threadIdx.x, b = sg * 0.5f;
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(a + i); hb[i] = (__bf16)(b + i); }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
        } else if (MODE == 1) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, c3, 0, 0, 0);
        } else {
            q0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q0, 0, 0, 0);
            q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q1, 0, 0, 0);
            q2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q2, 0, 0, 0);
            q3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q3, 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 4; ++i) r += q0[i] + q1[i] + q2[i] + q3[i];
    if (r == 123.456f) out[0] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
int run(const char* name, int waves_per_simd, double macs, float* d_out, unsigned long long* d_clk) {
    int blocks = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = 1 per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) k_rate<MODE><<<blocks, 256>>>(d_out, 1.0f, d_clk);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0));
        k_rate<MODE><<<blocks, 256>>>(d_out, 1.0f, d_clk);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    unsigned long long h[2]; CHECK(hipMemcpy(h, d_clk, 16, hipMemcpyDeviceToHost));
    double ghz = (double)h[0] / ((double)h[1] * 10.0);
    double instr_per_simd = (double)waves_per_simd * ITERS * 4;
    double total_instr = (double)blocks * 4 * ITERS * 4;
    printf("{\"bench\":\"%s\",\"waves_per_simd\":%d,\"ms\":%.4f,\"cyc_per_mfma_per_simd\":%.1f,\"TFLOPs\":%.1f,\"clock_GHz\":%.3f}\n", name, waves_per_simd, best,
           best * 1e-3 * ghz * 1e9 / instr_per_simd, 2.0 * macs * total_instr / (best * 1e-3) / 1e12, ghz);
    return 0;
}

int main() {
    float* d_out; unsigned long long* d_clk;
    CHECK(hipMalloc(&d_out, 1024)); CHECK(hipMalloc(&d_clk, 64));
    for (int w : {1, 4}) {
        if (run<0>("mfma_f32_32x32x2_f32", w, 32.0 * 32 * 2, d_out, d_clk)) return 1;
        if (run<1>("mfma_f32_32x32x16_bf16", w, 32.0 * 32 * 16, d_out, d_clk)) return 1;
        if (run<2>("mfma_f32_16x16x4_f32", w, 16.0 * 16 * 4, d_out, d_clk)) return 1;
    }
    return 0;
}

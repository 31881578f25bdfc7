// VALU encoding/operand microbenchmark #2 for gfx950 (follow-up to valu_rate.hip):
// is the ~4-cycle issue of v_fma_f32 / SGPR-operand ops a VOP3 (64-bit encoding) effect?
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 512;

#define I8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define X4(S) S S S S

#define ASM8(str) asm volatile(X4(str) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "s"(sg), "v"(d))

template <int MODE>
__global__ void __launch_bounds__(256) k_rate(float* out, float sg, unsigned long long* clk) {
    float a0 = sg + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float c = 1.0000001f, d = 0.9999999f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 0) ASM8("v_mul_f32_e32 %0, %0, %8\n v_mul_f32_e32 %1, %1, %8\n v_mul_f32_e32 %2, %2, %8\n v_mul_f32_e32 %3, %3, %8\n v_mul_f32_e32 %4, %4, %8\n v_mul_f32_e32 %5, %5, %8\n v_mul_f32_e32 %6, %6, %8\n v_mul_f32_e32 %7, %7, %8\n");
        if (MODE == 1) ASM8("v_mul_f32_e32 %0, %9, %0\n v_mul_f32_e32 %1, %9, %1\n v_mul_f32_e32 %2, %9, %2\n v_mul_f32_e32 %3, %9, %3\n v_mul_f32_e32 %4, %9, %4\n v_mul_f32_e32 %5, %9, %5\n v_mul_f32_e32 %6, %9, %6\n v_mul_f32_e32 %7, %9, %7\n");
        if (MODE == 2) ASM8("v_mul_f32_e64 %0, %0, %8\n v_mul_f32_e64 %1, %1, %8\n v_mul_f32_e64 %2, %2, %8\n v_mul_f32_e64 %3, %3, %8\n v_mul_f32_e64 %4, %4, %8\n v_mul_f32_e64 %5, %5, %8\n v_mul_f32_e64 %6, %6, %8\n v_mul_f32_e64 %7, %7, %8\n");
        if (MODE == 3) ASM8("v_fmac_f32_e32 %0, %8, %10\n v_fmac_f32_e32 %1, %8, %10\n v_fmac_f32_e32 %2, %8, %10\n v_fmac_f32_e32 %3, %8, %10\n v_fmac_f32_e32 %4, %8, %10\n v_fmac_f32_e32 %5, %8, %10\n v_fmac_f32_e32 %6, %8, %10\n v_fmac_f32_e32 %7, %8, %10\n");
        if (MODE == 4) ASM8("v_fma_f32 %0, %0, %8, %10\n v_fma_f32 %1, %1, %8, %10\n v_fma_f32 %2, %2, %8, %10\n v_fma_f32 %3, %3, %8, %10\n v_fma_f32 %4, %4, %8, %10\n v_fma_f32 %5, %5, %8, %10\n v_fma_f32 %6, %6, %8, %10\n v_fma_f32 %7, %7, %8, %10\n");
        if (MODE == 5) ASM8("v_subrev_f32_e32 %0, %9, %0\n v_subrev_f32_e32 %1, %9, %1\n v_subrev_f32_e32 %2, %9, %2\n v_subrev_f32_e32 %3, %9, %3\n v_subrev_f32_e32 %4, %9, %4\n v_subrev_f32_e32 %5, %9, %5\n v_subrev_f32_e32 %6, %9, %6\n v_subrev_f32_e32 %7, %9, %7\n");
        if (MODE == 6) ASM8("v_add_f32_e32 %0, %0, %1\n v_mul_f32_e32 %1, %1, %2\n v_add_f32_e32 %2, %2, %3\n v_mul_f32_e32 %3, %3, %4\n v_add_f32_e32 %4, %4, %5\n v_mul_f32_e32 %5, %5, %6\n v_add_f32_e32 %6, %6, %7\n v_mul_f32_e32 %7, %7, %8\n");
        if (MODE == 7) ASM8("v_fmac_f32_e32 %0, %9, %10\n v_fmac_f32_e32 %1, %9, %10\n v_fmac_f32_e32 %2, %9, %10\n v_fmac_f32_e32 %3, %9, %10\n v_fmac_f32_e32 %4, %9, %10\n v_fmac_f32_e32 %5, %9, %10\n v_fmac_f32_e32 %6, %9, %10\n v_fmac_f32_e32 %7, %9, %10\n");
        if (MODE == 8) ASM8("v_cmp_lt_f32_e32 vcc, %0, %8\n v_cmp_lt_f32_e32 vcc, %1, %8\n v_cmp_lt_f32_e32 vcc, %2, %8\n v_cmp_lt_f32_e32 vcc, %3, %8\n v_cmp_lt_f32_e32 vcc, %4, %8\n v_cmp_lt_f32_e32 vcc, %5, %8\n v_cmp_lt_f32_e32 vcc, %6, %8\n v_cmp_lt_f32_e32 vcc, %7, %8\n");
        if (MODE == 9) ASM8("v_max_f32_e32 %0, %0, %8\n v_min_f32_e32 %1, %1, %8\n v_max_f32_e32 %2, %2, %8\n v_min_f32_e32 %3, %3, %8\n v_max_f32_e32 %4, %4, %8\n v_min_f32_e32 %5, %5, %8\n v_max_f32_e32 %6, %6, %8\n v_min_f32_e32 %7, %7, %8\n");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (r == 123.456f) out[0] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
int run(const char* name, int waves_per_simd, float* d_out, unsigned long long* d_clk) {
    int blocks = 256 * waves_per_simd;
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
    double ghz = (double)h[0] / ((double)h[1] * 10.0) ;  // memrealtime ticks at 100 MHz -> ns*0.1
    double per_simd_instr = (double)waves_per_simd * ITERS * 32;
    double cyc_in_kernel = (double)h[0] / per_simd_instr;  // shader cycles (block 0 wave 0 span) per instr per SIMD
    printf("{\"bench\":\"%s\",\"waves_per_simd\":%d,\"ms\":%.4f,\"Tlane_ops_per_s\":%.2f,\"clock_GHz\":%.3f,\"cyc_per_instr_per_simd\":%.2f}\n",
           name, waves_per_simd, best, (double)blocks * 4 * ITERS * 32 * 64 / (best * 1e-3) / 1e12, ghz, cyc_in_kernel);
    return 0;
}

int main() {
    float* d_out; CHECK(hipMalloc(&d_out, 1024));
    unsigned long long* d_clk; CHECK(hipMalloc(&d_clk, 64));
    for (int w : {1, 2, 4}) {
        run<0>("v_mul_e32_vgpr", w, d_out, d_clk);
        run<1>("v_mul_e32_sgpr_src0", w, d_out, d_clk);
        run<2>("v_mul_e64_vgpr", w, d_out, d_clk);
        run<3>("v_fmac_e32_vgpr", w, d_out, d_clk);
        run<4>("v_fma_vop3", w, d_out, d_clk);
        run<5>("v_subrev_e32_sgpr_src0", w, d_out, d_clk);
        run<6>("mix_add_mul_e32_2src_distinct", w, d_out, d_clk);
        run<7>("v_fmac_e32_sgpr_src0", w, d_out, d_clk);
        run<8>("v_cmp_e32", w, d_out, d_clk);
        run<9>("v_minmax_e32", w, d_out, d_clk);
    }
    return 0;
}

// VALU issue-rate microbenchmark for gfx950: decides whether the sphere-scan
// inner loop should use scalar f32 ops or packed (v_pk_*_f32) ops.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int MODE>
__global__ void __launch_bounds__(256) k_rate(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    float c = 1.0000001f;
    f2 c2 = {c, c};
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 0) {  // v_mul_f32 x8, independent
            asm volatile(
                "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (MODE == 1) {  // v_fma_f32 x8
            asm volatile(
                "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (MODE == 2) {  // v_pk_mul_f32 x8
            asm volatile(
                "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));
        } else if (MODE == 3) {  // v_pk_fma_f32 x8
            asm volatile(
                "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));
        } else if (MODE == 4) {  // v_pk_add_f32 x8
            asm volatile(
                "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));
        } else if (MODE == 5) {  // v_sub_f32 with SGPR operand x8
            asm volatile(
                "v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                "v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed));
        } else if (MODE == 6) {  // dependent chain v_mul -> v_add (1 chain per lane)
            asm volatile(
                "v_mul_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n"
                "v_mul_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (MODE == 7) {  // v_cmp + v_cndmask pairs x4
            asm volatile(
                "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %0, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %2, vcc\n"
                "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %4, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %6, vcc\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
        } else if (MODE == 8) {  // f64 fma x8 (for the pow/sin contract cost)
            asm volatile(
                "v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n"
                "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));
        } else if (MODE == 9) {  // v_sqrt_f32 x8 (transcendental rate)
            asm volatile(
                "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (r == 123.456f) out[0] = r;  // keep live
}

template <int MODE>
int run(const char* name, int lanes_per_instr, int waves_per_simd, float* d_out) {
    int cus = 256;
    int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 wave per SIMD per block
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_rate<MODE><<<blocks, 256>>>(d_out, 1.0f);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0));
        k_rate<MODE><<<blocks, 256>>>(d_out, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double instr = (double)blocks * 4 /*waves*/ * ITERS * 8;
    double lane_ops = instr * lanes_per_instr;
    // cycles per wave-instruction per SIMD, assuming 2.4 GHz nominal
    double per_simd_instr = (double)waves_per_simd * ITERS * 8;
    double cyc = best * 1e-3 * 2.4e9 / per_simd_instr;
    printf("{\"bench\":\"%s\",\"waves_per_simd\":%d,\"ms\":%.4f,\"Tlane_ops_per_s\":%.2f,\"cycles_per_instr_per_simd_at_2.4GHz\":%.2f}\n",
           name, waves_per_simd, best, lane_ops / (best * 1e-3) / 1e12, cyc);
    return 0;
}

int main() {
    float* d_out; CHECK(hipMalloc(&d_out, 1024));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("{\"device\":\"%s\",\"cus\":%d,\"clock_khz\":%d}\n", p.name, p.multiProcessorCount, p.clockRate);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_mul_f32", 64, w, d_out);
        run<1>("v_fma_f32", 64, w, d_out);
        run<2>("v_pk_mul_f32", 128, w, d_out);
        run<3>("v_pk_fma_f32", 128, w, d_out);
        run<4>("v_pk_add_f32", 128, w, d_out);
        run<5>("v_sub_f32_sgpr", 64, w, d_out);
        run<6>("dep_chain_mul_add", 64, w, d_out);
        run<7>("v_cmp+v_cndmask", 64, w, d_out);
        run<8>("v_fma_f64", 64, w, d_out);
        run<9>("v_sqrt_f32", 64, w, d_out);
    }
    return 0;
}

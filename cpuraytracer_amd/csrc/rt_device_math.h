// rt_device_math.h — device-side arithmetic contract of the render-loop kernels (gfx950).
//
// Every function below is the HIP implementation of one reference function (file:line cited) or
// of one DirectXMath 3.11 SSE2-path function the reference calls (semantics per SURVEY.md §8(c)).
// The evaluation order is the reference's: separate multiply and add, left to right.  The file is
// compiled with -ffp-contract=off and correctly rounded divide/sqrt so results are IEEE-exact and
// reproducible; the CPU oracle (oracle/) restates the same functions independently and the unit
// tests compare the two bit for bit.
//
// Elementary functions: the reference calls MSVC sinf/cosf/powf (quasi-random.cpp:45-47,58-59,
// XMVectorPow).  Neither those bits nor ocml's are reproducible across CPU and GPU, so the path
// defines binary64 polynomial kernels rounded once to binary32 (spec: DESIGN.md "Elementary
// functions").  MI355X runs f64 VALU at half the f32 rate, and these are called a handful of
// times per hit against ~10^4 f32 ops per list scan, so the cost is noise.
#pragma once

#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_DEV __host__ __device__ __forceinline__
#else
// The host mirror (csrc/host/) compiles the same arithmetic with g++ -ffp-contract=off for the
// once-per-scene work (Camera::Camera, InitScene); nothing on the render path runs on the CPU.
#define RT_DEV inline
#endif
#include "rt_sites.h"

namespace rtd {

struct V3 {
    float x, y, z;
};

RT_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RT_DEV V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DEV V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_DEV V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_DEV V3 operator*(float s, V3 a) { return {a.x * s, a.y * s, a.z * s}; }
RT_DEV V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
RT_DEV V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }  // true divide per lane
RT_DEV V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }

// XMVector3Dot: (x1*x2 + y1*y2) + z1*z2
RT_DEV float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

RT_DEV float sat1(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }  // XMVectorSaturate per lane

// ---------------------------------------------------------------- correctly rounded division, cheaper
// x / b as the compiler expands it (v_div_scale x2, v_rcp, five fma, v_div_fmas, v_div_fixup) is 11 VALU operations, and the
// hit processing divides ~20 times per hit, mostly several numerators by one denominator.  Markstein's theorem: with
// y = RN(1/b) and q0 = RN(x y), the residual r = x - b q0 is exact in an fma and q1 = RN(q0 + r y) = RN(x / b), provided
// nothing leaves the normal range.  recip_rn refines the hardware's 1-ulp v_rcp_f32 with one Newton step (3 operations);
// div_rn is mul + fma + fma + a sign transfer (x = -0 would otherwise give +0).  Preconditions, checked by the callers
// with div_exponents_ok(): b in [2^-20, 2^100] and every numerator zero or with x / b >= 2^-80 in magnitude (then
// |x| >= 2^-100, the quotient is normal and r, a multiple of 2^(e_x - 46), is representable).  Outside them the plain
// division runs.  tests/test_gpu_parity.py::test_markstein_division_is_ieee_division runs both on the device for all 2^23
// significands of b against 1/b and for millions of quotients incl. the guard's edges; a CPU brute force over 1.9e10
// (x, b) pairs of the same sequence found no mismatch.  Host builds (the C++ host mirror) use the plain division.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_NO_MARKSTEIN)
#define RT_MARKSTEIN 1
RT_DEV float recip_rn(float b) {
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
RT_DEV float div_rn(float x, float b, float y) {
    const float q0 = x * y;
    const float r = __builtin_fmaf(-b, q0, x);
    return __builtin_copysignf(__builtin_fmaf(r, y, q0), x);  // b > 0 at every call site
}
RT_DEV int float_exponent(float x) { return __builtin_amdgcn_frexp_expf(x); }  // 0 for x == 0, else floor(log2|x|) + 1
// one denominator, three numerators (a zero numerator has exponent 0: fine unless b > 2^80, where the slow path is taken)
RT_DEV bool div_exponents_ok(float x0, float x1, float x2, float b) {
    const int e0 = float_exponent(x0), e1 = float_exponent(x1), e2 = float_exponent(x2), eb = float_exponent(b);
    const int m = e0 < e1 ? (e0 < e2 ? e0 : e2) : (e1 < e2 ? e1 : e2);
    return (unsigned)(eb + 19) <= 119u && m - eb >= -80;
}
#else
#define RT_MARKSTEIN 0
#endif

// ---------------------------------------------------------------- correctly rounded square root, cheaper
// sqrtf as the compiler expands it (scale test, v_sqrt_f32, the two neighbours with an fma residual each, two compare-and-select
// pairs with their VCC wait states, unscale, class test) is 16 VALU operations and 4 s_nop.  Markstein's correction again: with
// s0 = v_sqrt_f32(x) (1 ulp) and h = 0.5 v_rsq_f32(x) (1 ulp), the residual d = x - s0^2 is exact in an fma and
// RN(s0 + d h) = RN(sqrt x): four operations behind the two hardware instructions.  Valid for x in [2^-80, inf) -- normal, away
// from the flush of the two instructions; tests/test_gpu_parity.py::test_fast_square_root_is_ieee_square_root runs it on the
// device for EVERY float of that range (1.74e9 patterns) against the IEEE root.  Anything else (zero, tiny, negative, inf, NaN
// in ANY active lane of the wave: one uniform branch, not taken in practice) takes the compiler's sequence for the whole wave.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_NO_FAST_SQRT)
#define RT_FAST_SQRT 1
RT_DEV float sqrt_rn_core(float x) {
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float d = __builtin_fmaf(-s0, s0, x);
    return __builtin_fmaf(d, h, s0);
}
RT_DEV bool sqrt_fast_range(float x) { return (__float_as_uint(x) - 0x17800000u) < 0x68000000u; }  // 2^-80 <= x < inf
RT_DEV float sqrt_rn(float x) {
    if (__builtin_expect(__ballot(!sqrt_fast_range(x)) == 0ull, 1)) return sqrt_rn_core(x);
    {
        RT_SITE(M_SQRT_SLOW);
        return __builtin_sqrtf(x);
    }
}
#else
#define RT_FAST_SQRT 0
RT_DEV float sqrt_rn(float x) { return __builtin_sqrtf(x); }
#endif

// v / b per component (true divide), b > 0.
RT_DEV V3 div3(V3 v, float b) {
#if RT_MARKSTEIN
    if (__builtin_expect(div_exponents_ok(v.x, v.y, v.z, b), 1)) {
        const float y = recip_rn(b);
        return {div_rn(v.x, b, y), div_rn(v.y, b, y), div_rn(v.z, b, y)};
    }
#endif
    {
        RT_SITE(M_DIV_SLOW);
        return {v.x / b, v.y / b, v.z / b};
    }
}

// XMVector3Normalize (SSE2): zero length -> 0, infinite length -> QNaN, else true divide.
RT_DEV V3 normalize3(V3 v) {
    const float lenSq = dot3(v, v);
    const float len = sqrt_rn(lenSq);
    if (lenSq == __builtin_inff()) {
        const float q = __builtin_nanf("");
        return {q, q, q};
    }
    if (len == 0.f) return {0.f, 0.f, 0.f};
    return div3(v, len);
}

// XMVector3Cross
RT_DEV V3 cross3(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// XMVector3Reflect: s = dot(I,N); s = s + s; I - s*N
RT_DEV V3 reflect3(V3 I, V3 N) {
    float s = dot3(I, N);
    s = s + s;
    return {I.x - s * N.x, I.y - s * N.y, I.z - s * N.z};
}

// XMVector3RefractV: d = dot(I,N); k = 1 - ((1 - d*d)*eta)*eta; k <= 0 -> zero; eta*I - (sqrt(k) + eta*d)*N
RT_DEV V3 refract3(V3 I, V3 N, float e) {
    const float d = dot3(I, N);
    float k = d * d;
    k = 1.f - k;
    k = k * e;
    k = k * e;
    k = 1.f - k;
    if (k <= 0.f) return {0.f, 0.f, 0.f};
    float r = sqrt_rn(k);
    r = r + e * d;
    return {e * I.x - r * N.x, e * I.y - r * N.y, e * I.z - r * N.z};
}

// XMFresnelTerm(c, n)
RT_DEV float fresnel_term(float c, float n) {
    float g = n * n;
    float t = c * c;
    g = g - 1.f;
    t = t + g;
    g = __builtin_fabsf(t);
    g = sqrt_rn(g);
    float gAddC = g + c;
    float gSubC = g - c;
    float res = gSubC * gSubC;
    t = gAddC * gAddC;
    res = res * 0.5f;
    res = res / t;
    gAddC = gAddC * c;
    gSubC = gSubC * c;
    gAddC = gAddC - 1.f;
    gSubC = gSubC + 1.f;
    gAddC = gAddC * gAddC;
    gSubC = gSubC * gSubC;
    gAddC = gAddC / gSubC;
    gAddC = gAddC + 1.f;
    res = res * gAddC;
    res = res < 0.f ? 0.f : res;
    res = res > 1.f ? 1.f : res;
    return res;
}

// ------------------------------------------------------------- elementary functions (f64 kernels)
// Contract (oracle/dxmath_restate.h restates it independently): binary64 kernels with EXPLICIT fused
// multiply-adds (__builtin_fma: one rounding, identical on gfx950's v_fma_f64 and on the host's FMA unit / libm fma),
// table lookups into rt_math_tables.inc (generated, tools/gen_math_tables.py) and one final rounding to binary32.
// The FIRST step of every Horner chain is an unfused multiply + add: a fused one would have two literal operands, and
// gfx950's VOP3 encoding reads at most one scalar/literal per instruction -- the second constant would sit in a VGPR
// pair for the whole persistent loop (measured: 10 spilled VGPRs).
// gfx950 runs f64 VALU at half the f32 rate, so the kernels are short: sincos 24 f64 operations (64-entry table of
// sin/cos(j pi/32), degree-7/8 remainders), pow 36 (47-entry reciprocal/log2 table, degree-9 log1p; 33-entry 2^(j/32)
// table, degree-6 exp) -- the round-1 kernels (long division, degree-21/13 series without fma) were 76 and 110.
RT_DEV double f64_of_bits(unsigned long long b) { return __builtin_bit_cast(double, b); }

// A binary64 literal for the Horner chains below.  gfx950's VOP3 encoding has no 64-bit literals, so the compiler keeps
// each coefficient in a register pair; left alone it hoists a dozen of them into VGPR pairs for the whole persistent loop
// (spills) and copies one into the destination before every v_fmac_f64.  The empty asm pins the coefficient to a scalar
// pair right where it is used: two s_mov_b32 on the scalar unit, then one v_fma_f64 with a scalar addend.
#if defined(__HIP_DEVICE_COMPILE__)
RT_DEV double kf64(double c) {
    asm volatile("" : "+s"(c));
    return c;
}
#else
RT_DEV double kf64(double c) { return c; }
#endif

// Where a caller reads the contract's tables from: the constant copy in the code object (default), or the copy the trace
// kernel stages into LDS (an L2 round trip per lookup would otherwise sit in front of every pow and sincos).
struct MathTabs {
    const unsigned long long* log;     // [47][2]  r = RN(1 / (1 + idx/64)), l = RN(-log2 r)
    const unsigned long long* exp2;    // [33]     RN(2^(j/32)), j = -16 .. 16
    const unsigned long long* sincos;  // [64][2]  RN(sin(j pi/32)), RN(cos(j pi/32))
};
constexpr unsigned kMathTabWords = 47 * 2 + 33 + 64 * 2;  // 255 64-bit words
RT_DEV MathTabs default_math_tabs() {
#include "rt_math_tables.inc"
    return MathTabs{&kLogTabBits[0][0], &kExp2TabBits[0], &kSinCosTabBits[0][0]};
}

// x >= 0.  j = rint(x * 32/pi); r = x - j pi/32 (two-term); sin/cos(x) = S_j cos r + C_j sin r, C_j cos r - S_j sin r.
// At the multiples of pi/2 the table holds exact 0 / +-1, so results near the zeros keep full relative accuracy.
RT_DEV void sincos_f64(float xf, double& s_out, double& c_out, const MathTabs& T) {
    const double x = (double)xf;
    const double jd = __builtin_rint(x * 10.185916357881301);  // 32/pi
    const int j = (int)jd & 63;
    double r = __builtin_fma(-jd, 0.09817477042468103, x);       // pi/32 hi = 0x3FB921FB54442D18
    r = __builtin_fma(-jd, 3.8270212473354788e-18, r);           // pi/32 lo = 0x3C51A62633145C07
    const double z = r * r;
    double ps = z * -1.984126984126984e-04 + 8.333333333333333e-03;  // -1/5040, 1/120
    ps = __builtin_fma(z, ps, kf64(-1.6666666666666666e-01));                            // -1/6
    const double sr = __builtin_fma(r * z, ps, r);
    double pc = z * 2.48015873015873e-05 + -1.388888888888889e-03;   // 1/40320, -1/720
    pc = __builtin_fma(z, pc, kf64(4.1666666666666664e-02));                            // 1/24
    pc = __builtin_fma(z, pc, -0.5);
    const double cr = __builtin_fma(z, pc, 1.0);
    const double S = f64_of_bits(T.sincos[2 * j]), C = f64_of_bits(T.sincos[2 * j + 1]);
    s_out = __builtin_fma(C, sr, S * cr);
    c_out = __builtin_fma(-S, sr, C * cr);
}
RT_DEV void sincos_f64(float xf, double& s_out, double& c_out) { sincos_f64(xf, s_out, c_out, default_math_tabs()); }
RT_DEV float rt_sinf(float x) { double s, c; sincos_f64(x, s, c); return (float)s; }
RT_DEV float rt_cosf(float x) { double s, c; sincos_f64(x, s, c); return (float)c; }
RT_DEV float rt_tanf(float x) { double s, c; sincos_f64(x, s, c); return (float)(s / c); }

// pow(x,y) = 2^(y*log2 x), x >= 0.
RT_DEV float rt_powf(float xf, float yf, const MathTabs& T) {
    if (yf == 0.f) return 1.f;
    if (!(xf > 0.f)) return 0.f;
    if (xf == 1.f) return 1.f;
    const double x = (double)xf;
    if (yf == 5.f) {  // Schlick's (1 - n.v)^5 (material.cpp:27,79; light.cpp:37): three exact-order f64 products
        const double x2 = x * x;
        const double x4 = x2 * x2;
        return (float)(x4 * x);
    }
    // log2 x = e + log2 m, m in [sqrt(1/2), sqrt(2)); idx = rint((m - 1) 64), r = RN(1 / (1 + idx/64)), u = m r - 1 (|u| < 0.0113),
    // log2 m = -log2 r + log1p(u) / ln 2.  idx == 0 has r = 1 exactly: no cancellation for x near 1.
    const uint64_t bits = __builtin_bit_cast(uint64_t, x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    double m = __builtin_bit_cast(double, (uint64_t)((bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL));
    if (m > 1.41421356237309514547) {
        m = m * 0.5;
        e = e + 1;
    }
    const int idx = (int)__builtin_rint((m - 1.0) * 64.0);
    const double rj = f64_of_bits(T.log[2 * (idx + 19)]), lj = f64_of_bits(T.log[2 * (idx + 19) + 1]);
    const double u = __builtin_fma(m, rj, -1.0);
    double p = u * 1.1111111111111111e-01 + -0.125;      // u^9/9, -u^8/8
    p = __builtin_fma(u, p, kf64(1.4285714285714285e-01));                   // 1/7
    p = __builtin_fma(u, p, kf64(-1.6666666666666666e-01));                  // -1/6
    p = __builtin_fma(u, p, kf64(0.2));
    p = __builtin_fma(u, p, kf64(-0.25));
    p = __builtin_fma(u, p, kf64(3.3333333333333331e-01));                   // 1/3
    p = __builtin_fma(u, p, -0.5);
    const double lp = __builtin_fma(u * u, p, u);                      // log1p(u)
    const double log2x = __builtin_fma(lp, 1.4426950408889634, (double)e + lj);  // 1/ln 2
    const double t = (double)yf * log2x;
    if (t < -160.0) return 0.f;
    if (t > 160.0) return __builtin_inff();
    // 2^t = 2^k 2^(j/32) e^(h ln 2), k = rint(t), j = rint(32 (t - k)) in [-16, 16], |h| <= 1/64
    const double kd = __builtin_rint(t);
    const double f = t - kd;
    const double jd = __builtin_rint(f * 32.0);
    const double h = __builtin_fma(jd, -0.03125, f);
    const double g = h * 0.69314718055994531;  // ln 2
    double q = g * 1.3888888888888889e-03 + 8.3333333333333332e-03;  // 1/720, 1/120
    q = __builtin_fma(g, q, kf64(4.1666666666666664e-02));                                // 1/24
    q = __builtin_fma(g, q, kf64(1.6666666666666666e-01));                                // 1/6
    q = __builtin_fma(g, q, 0.5);
    q = __builtin_fma(g, q, 1.0);
    q = __builtin_fma(g, q, 1.0);
    const double w = f64_of_bits(T.exp2[(int)jd + 16]) * q;
    return (float)__builtin_ldexp(w, (int)kd);
}
RT_DEV float rt_powf(float xf, float yf) { return rt_powf(xf, yf, default_math_tabs()); }

// ----------------------------------------------------------------------- quasi-random.cpp
// Random::HaltonSample (quasi-random.cpp:3-16) with a 32-bit index: every index on the path is
// sampleCount + i + j < 2^32 (SURVEY.md §7 "Integer division").
RT_DEV float halton(uint32_t index, uint32_t base) {
    float result = 0.f;
    float f = 1.f;
    const float fb = (float)base;
    while (index > 0) {
        f = f / fb;
        result += f * (float)(index % base);
        index = index / base;
    }
    return result;
}

// ------------------------------------------------------------------------- RNG contract (A9)
RT_DEV uint64_t splitmix64_mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
struct Rng {
    uint32_t s0, s1, s2, s3;  // xoshiro128** state, lives in VGPRs
};
RT_DEV Rng rng_seed(uint64_t seed, uint32_t pixelId, uint32_t sample) {
    const uint64_t key = ((uint64_t)pixelId << 32) | (uint64_t)sample;
    const uint64_t a = splitmix64_mix(splitmix64_mix(seed) ^ key);
    const uint64_t b = splitmix64_mix(a);
    Rng r{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    if ((r.s0 | r.s1 | r.s2 | r.s3) == 0u) r.s0 = 1u;
    return r;
}
RT_DEV uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
RT_DEV float rng_uniform(Rng& r) {
    const uint32_t result = rotl32(r.s1 * 5u, 7) * 9u;
    const uint32_t t = r.s1 << 9;
    r.s2 ^= r.s0;
    r.s3 ^= r.s1;
    r.s1 ^= r.s2;
    r.s0 ^= r.s3;
    r.s2 ^= t;
    r.s3 = rotl32(r.s3, 11);
    return (float)(result >> 8) * 0x1p-24f;
}

// XMStoreColor channel: saturate, *255, round to nearest even.
RT_DEV uint32_t rne_u8(float c) {
    c = sat1(c);
    c = c * 255.0f;
    return (uint32_t)__builtin_rintf(c);
}

}  // namespace rtd

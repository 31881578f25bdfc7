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

// XMVector3Normalize (SSE2): zero length -> 0, infinite length -> QNaN, else true divide.
RT_DEV V3 normalize3(V3 v) {
    const float lenSq = dot3(v, v);
    const float len = __builtin_sqrtf(lenSq);
    if (lenSq == __builtin_inff()) {
        const float q = __builtin_nanf("");
        return {q, q, q};
    }
    if (len == 0.f) return {0.f, 0.f, 0.f};
    return {v.x / len, v.y / len, v.z / len};
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
    float r = __builtin_sqrtf(k);
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
    g = __builtin_sqrtf(g);
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
RT_DEV void sincos_f64(float xf, double& s_out, double& c_out) {
    const double TWO_OVER_PI = 0.63661977236758138243;
    const double PIO2_HI = 1.57079632679489655800e+00;
    const double PIO2_LO = 6.12323399573676603587e-17;
    const double x = (double)xf;
    const int k = (int)(x * TWO_OVER_PI + 0.5);  // x >= 0
    const double kd = (double)k;
    double r = x - kd * PIO2_HI;
    r = r - kd * PIO2_LO;
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = ps * z + -2.50507602534068634195e-08;
    ps = ps * z + 2.75573137070700676789e-06;
    ps = ps * z + -1.98412698298579493134e-04;
    ps = ps * z + 8.33333333332248946124e-03;
    ps = ps * z + -1.66666666666666324348e-01;
    const double sr = r + (r * z) * ps;
    double pc = -1.13596475577881948265e-11;
    pc = pc * z + 2.08757232129817482790e-09;
    pc = pc * z + -2.75573143513906633035e-07;
    pc = pc * z + 2.48015872894767294178e-05;
    pc = pc * z + -1.38888888888741095749e-03;
    pc = pc * z + 4.16666666666666019037e-02;
    const double cr = 1.0 - (0.5 * z - (z * z) * pc);
    const int quad = k & 3;
    s_out = (quad == 0) ? sr : (quad == 1) ? cr : (quad == 2) ? -sr : -cr;
    c_out = (quad == 0) ? cr : (quad == 1) ? -sr : (quad == 2) ? -cr : sr;
}
RT_DEV float rt_sinf(float x) { double s, c; sincos_f64(x, s, c); return (float)s; }
RT_DEV float rt_cosf(float x) { double s, c; sincos_f64(x, s, c); return (float)c; }
RT_DEV float rt_tanf(float x) { double s, c; sincos_f64(x, s, c); return (float)(s / c); }

// pow(x,y) = 2^(y*log2 x), x >= 0.
RT_DEV float rt_powf(float xf, float yf) {
    if (yf == 0.f) return 1.f;
    if (!(xf > 0.f)) return 0.f;
    if (xf == 1.f) return 1.f;
    const double x = (double)xf;
    if (yf == 5.f) {  // Schlick's (1 - n.v)^5 (material.cpp:27,79; light.cpp:37): three exact-order f64 products
        const double x2 = x * x;
        const double x4 = x2 * x2;
        return (float)(x4 * x);
    }
    const uint64_t bits = __builtin_bit_cast(uint64_t, x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    double m = __builtin_bit_cast(double, (uint64_t)((bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL));
    if (m > 1.41421356237309514547) {
        m = m * 0.5;
        e = e + 1;
    }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 0.047619047619047616404;
    p = p * z + 0.052631578947368418131;
    p = p * z + 0.058823529411764705066;
    p = p * z + 0.066666666666666665741;
    p = p * z + 0.076923076923076927347;
    p = p * z + 0.090909090909090911614;
    p = p * z + 0.11111111111111110494;
    p = p * z + 0.14285714285714284921;
    p = p * z + 0.2000000000000000111;
    p = p * z + 0.33333333333333331483;
    p = p * z + 1.0;
    const double lnm = (2.0 * s) * p;
    const double log2x = (double)e + lnm * 1.4426950408889633870;
    const double t = (double)yf * log2x;
    if (t < -160.0) return 0.f;
    if (t > 160.0) return __builtin_inff();
    const int k = (int)(t + (t >= 0.0 ? 0.5 : -0.5));
    const double f = t - (double)k;
    const double g = f * 0.69314718055994528623;
    double q = 1.6059043836821613341e-10;
    q = q * g + 2.0876756987868100187e-09;
    q = q * g + 2.5052108385441720224e-08;
    q = q * g + 2.7557319223985892511e-07;
    q = q * g + 2.7557319223985888276e-06;
    q = q * g + 2.4801587301587301566e-05;
    q = q * g + 1.9841269841269841253e-04;
    q = q * g + 1.3888888888888889419e-03;
    q = q * g + 8.3333333333333332177e-03;
    q = q * g + 4.1666666666666664354e-02;
    q = q * g + 1.6666666666666665741e-01;
    q = q * g + 0.5;
    q = q * g + 1.0;
    q = q * g + 1.0;
    const double scale = __builtin_bit_cast(double, (uint64_t)(1023 + k) << 52);
    return (float)(q * scale);
}

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

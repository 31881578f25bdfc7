// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or executed by
// the product path (cpuraytracer_amd/); only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use it.
//
// dxmath_restate.h — CPU restatement of the DirectXMath 3.11 (SSE2 code path) functions the
// reference's hot path calls.  DirectXMath is NOT vendored in /root/reference (it ships in Windows
// SDK 10.0.16299.0, common-lib.vcxproj:17); its published algorithms are restated here from the
// library's public source, per the table in SURVEY.md §8(c).  PARITY UNPINNED with respect to the
// original binary: the reference has no tests or golden vectors at this level and is
// itself non-deterministic.  What pins this file: closed-form checks in
// tests/test_oracle_units.py (XMFresnelTerm(1,1.5)=0.04, Snell, XMCOLOR quantisation, ...).
//
// Every operation is spelled in the library's evaluation order with separate multiply and add
// (SSE2 has no FMA; the oracle is compiled with -ffp-contract=off).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <cfloat>

namespace orc {

// ------------------------------------------------------------------------------------ types
// XMVECTOR: 16-byte float4 (stdafx.h:26 -> DirectXMath.h).
struct alignas(16) XMVECTOR {
    float x, y, z, w;
};
struct XMFLOAT2 {
    float x, y;
    XMFLOAT2() = default;
    XMFLOAT2(float x_, float y_) : x(x_), y(y_) {}
};
struct XMFLOAT3 {
    float x, y, z;
    XMFLOAT3() = default;
    XMFLOAT3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
using XMVECTORF32 = XMVECTOR;

constexpr float XM_PI = 3.141592654f;

inline XMVECTOR XMVectorSet(float x, float y, float z, float w) { return XMVECTOR{x, y, z, w}; }
inline XMVECTOR XMVectorReplicate(float v) { return XMVECTOR{v, v, v, v}; }
inline XMVECTOR XMVectorZero() { return XMVECTOR{0.f, 0.f, 0.f, 0.f}; }
inline float XMVectorGetX(XMVECTOR v) { return v.x; }
inline void XMStoreFloat3(XMFLOAT3* d, XMVECTOR v) { d->x = v.x; d->y = v.y; d->z = v.z; }
inline XMVECTOR XMLoadFloat3(const XMFLOAT3* s) { return XMVECTOR{s->x, s->y, s->z, 0.f}; }

// stdafx.h:34-35
#define ORC_XM_One ::orc::XMVectorReplicate(1.f)
#define ORC_XM_Zero ::orc::XMVectorZero()

// Per-lane IEEE ops (XMVectorAdd/Subtract/Multiply/Divide, XMVectorScale, operator overloads).
inline XMVECTOR operator+(XMVECTOR a, XMVECTOR b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline XMVECTOR operator-(XMVECTOR a, XMVECTOR b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline XMVECTOR operator*(XMVECTOR a, XMVECTOR b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline XMVECTOR operator/(XMVECTOR a, XMVECTOR b) { return {a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w}; }
inline XMVECTOR operator*(XMVECTOR a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline XMVECTOR operator*(float s, XMVECTOR a) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline XMVECTOR operator/(XMVECTOR a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }  // true divide by replicated scalar
inline XMVECTOR operator-(XMVECTOR a) { return {-a.x, -a.y, -a.z, -a.w}; }
inline XMVECTOR& operator+=(XMVECTOR& a, XMVECTOR b) { a = a + b; return a; }

// XMVectorMultiplyAdd(a,b,c) = a*b + c, unfused under SSE2.
inline XMVECTOR XMVectorMultiplyAdd(XMVECTOR a, XMVECTOR b, XMVECTOR c) { return a * b + c; }

inline float sat1(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }  // max(0,.) then min(1,.)
inline XMVECTOR XMVectorSaturate(XMVECTOR v) { return {sat1(v.x), sat1(v.y), sat1(v.z), sat1(v.w)}; }
inline XMVECTOR XMVectorSqrt(XMVECTOR v) { return {std::sqrt(v.x), std::sqrt(v.y), std::sqrt(v.z), std::sqrt(v.w)}; }

// XMVector3Dot: ((x1*x2 + y1*y2) + z1*z2) replicated to all four lanes.
inline float Dot3(XMVECTOR a, XMVECTOR b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline XMVECTOR XMVector3Dot(XMVECTOR a, XMVECTOR b) { return XMVectorReplicate(Dot3(a, b)); }

// XMVector3Length: sqrt of the dot, replicated.
inline XMVECTOR XMVector3Length(XMVECTOR v) { return XMVectorReplicate(std::sqrt(Dot3(v, v))); }

// XMVector3Normalize (SSE2): lenSq; sqrt; true divide; zero length -> 0; infinite length -> QNaN.
inline XMVECTOR XMVector3Normalize(XMVECTOR v) {
    const float lenSq = Dot3(v, v);
    const float len = std::sqrt(lenSq);
    if (lenSq == INFINITY) {
        const float q = std::nanf("");
        return {q, q, q, q};
    }
    if (len == 0.f) return XMVectorZero();
    return {v.x / len, v.y / len, v.z / len, v.w / len};
}

// XMVector3Cross: (a.y b.z - a.z b.y, a.z b.x - a.x b.z, a.x b.y - a.y b.x, 0)
inline XMVECTOR XMVector3Cross(XMVECTOR a, XMVECTOR b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, 0.f};
}

// XMVector3Reflect(I,N): s = dot(I,N); s = s + s; I - s*N
inline XMVECTOR XMVector3Reflect(XMVECTOR I, XMVECTOR N) {
    float s = Dot3(I, N);
    s = s + s;
    return {I.x - s * N.x, I.y - s * N.y, I.z - s * N.z, I.w - s * N.w};
}

// XMVector3RefractV(I,N,eta): d = dot(I,N); k = 1 - ((1 - d*d)*eta)*eta; k <= 0 -> 0 vector;
// else eta*I - (eta*d + sqrt(k))*N.
inline XMVECTOR XMVector3RefractV(XMVECTOR I, XMVECTOR N, XMVECTOR eta) {
    const float d = Dot3(I, N);
    const float e = eta.x;
    float k = d * d;
    k = 1.f - k;
    k = k * e;
    k = k * e;
    k = 1.f - k;
    if (k <= 0.f) return XMVectorZero();
    float r = std::sqrt(k);
    r = r + e * d;
    return {e * I.x - r * N.x, e * I.y - r * N.y, e * I.z - r * N.z, e * I.w - r * N.w};
}

// XMFresnelTerm(c, n): g = sqrt(|c^2 + (n^2 - 1)|);
// 0.5*(g-c)^2/(g+c)^2 * (((c(g+c)-1)^2 / (c(g-c)+1)^2) + 1), clamped to [0,1].
inline float FresnelTerm1(float c, float n) {
    float g = n * n;
    float t = c * c;
    g = g - 1.f;
    t = t + g;
    g = std::fabs(t);  // max(0 - t, t)
    g = std::sqrt(g);
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
    res = res < 0.f ? 0.f : res;  // max(res, 0)
    res = res > 1.f ? 1.f : res;  // min(res, 1)
    return res;
}
inline XMVECTOR XMFresnelTerm(XMVECTOR c, XMVECTOR n) { return XMVectorReplicate(FresnelTerm1(c.x, n.x)); }

// XMVectorReciprocalEst: SSE rcpps (~12 bit, CPU-vendor dependent).  Restated as the exact
// reciprocal; deviation <= 3.7e-4 relative on eta, recorded in SURVEY.md §8(c) and DESIGN.md.
inline XMVECTOR XMVectorReciprocalEst(XMVECTOR v) { return {1.0f / v.x, 1.0f / v.y, 1.0f / v.z, 1.0f / v.w}; }

// Comparisons: XMVector3Greater/Less true iff ALL of x,y,z satisfy; XMVector3NotEqual iff ANY differs.
inline bool XMVector3Greater(XMVECTOR a, XMVECTOR b) { return a.x > b.x && a.y > b.y && a.z > b.z; }
inline bool XMVector3Less(XMVECTOR a, XMVECTOR b) { return a.x < b.x && a.y < b.y && a.z < b.z; }
inline bool XMVector3NotEqual(XMVECTOR a, XMVECTOR b) { return a.x != b.x || a.y != b.y || a.z != b.z; }
// XMVectorGreaterR compares all FOUR lanes; XMComparisonAnyTrue = not all false.
inline void XMVectorGreaterR(uint32_t* cr, XMVECTOR a, XMVECTOR b) {
    const int m = (a.x > b.x) | ((a.y > b.y) << 1) | ((a.z > b.z) << 2) | ((a.w > b.w) << 3);
    *cr = (m == 0xf) ? 0x80u /*XM_CRMASK_CR6TRUE*/ : (m == 0 ? 0x20u /*XM_CRMASK_CR6FALSE*/ : 0u);
}
inline bool XMComparisonAnyTrue(uint32_t cr) { return (cr & 0x20u) != 0x20u; }

// ------------------------------------------------------------------------ packed colour
// XMCOLOR: 32-bit ARGB word, B in the low byte.  Ctor from floats = XMStoreColor:
// saturate, *255, round to nearest even (cvtps2dq), pack.
inline uint32_t rne_u8(float c) {
    c = sat1(c);
    c = c * 255.0f;
    return (uint32_t)std::nearbyintf(c);  // default rounding mode: nearest even
}
struct XMCOLOR {
    uint32_t c;
    XMCOLOR() = default;
    XMCOLOR(float r, float g, float b, float a) { c = (rne_u8(a) << 24) | (rne_u8(r) << 16) | (rne_u8(g) << 8) | rne_u8(b); }
};
inline void XMStoreColor(XMCOLOR* d, XMVECTOR v) { *d = XMCOLOR(v.x, v.y, v.z, v.w); }
// XMLoadColor: (R,G,B,A) bytes scaled by 1/255 — byte * (1.0f/255.0f), see SURVEY.md §8(c).
inline XMVECTOR XMLoadColor(const XMCOLOR* s) {
    const float k = 1.0f / 255.0f;
    return {(float)((s->c >> 16) & 0xff) * k, (float)((s->c >> 8) & 0xff) * k, (float)(s->c & 0xff) * k,
            (float)((s->c >> 24) & 0xff) * k};
}

// ------------------------------------------------------------------- DirectXCollision
// BoundingBox with Intersects(origin, direction, dist) and CreateMerged.
struct BoundingBox {
    XMFLOAT3 Center{0, 0, 0};
    XMFLOAT3 Extents{1, 1, 1};
    BoundingBox() = default;
    BoundingBox(const XMFLOAT3& c, const XMFLOAT3& e) : Center(c), Extents(e) {}

    // T = centre - o; parallel if |d| <= 1e-20; inv = 1/d; t1 = (T-ext)*inv, t2 = (T+ext)*inv;
    // tmin = max3(min(t1,t2)) (parallel axes -> -FLT_MAX); tmax = min3(max(t1,t2)) (-> +FLT_MAX);
    // reject if tmin > tmax, tmax < 0, or (parallel and |T| > ext on that axis).
    bool Intersects(XMVECTOR origin, XMVECTOR dir, float& dist) const {
        const float c[3] = {Center.x, Center.y, Center.z};
        const float e[3] = {Extents.x, Extents.y, Extents.z};
        const float o[3] = {origin.x, origin.y, origin.z};
        const float d[3] = {dir.x, dir.y, dir.z};
        float tmin = -FLT_MAX, tmax = FLT_MAX;
        bool reject = false;
        float tmn[3], tmx[3];
        for (int k = 0; k < 3; ++k) {
            const float T = c[k] - o[k];
            const bool par = std::fabs(d[k]) <= 1e-20f;
            const float inv = 1.0f / d[k];
            const float t1 = (T - e[k]) * inv;
            const float t2 = (T + e[k]) * inv;
            // SSE min/max semantics: min(a,b) = a < b ? a : b
            tmn[k] = par ? -FLT_MAX : (t1 < t2 ? t1 : t2);
            tmx[k] = par ? FLT_MAX : (t1 > t2 ? t1 : t2);
            // XMVectorInBounds(T, ext): -ext <= T <= ext ; reject when parallel and NOT in bounds
            const bool inb = (T <= e[k]) && (-e[k] <= T);
            if (par && !inb) reject = true;
        }
        tmin = tmn[0] > tmn[1] ? tmn[0] : tmn[1];
        tmin = tmin > tmn[2] ? tmin : tmn[2];
        tmax = tmx[0] < tmx[1] ? tmx[0] : tmx[1];
        tmax = tmax < tmx[2] ? tmax : tmx[2];
        if (tmin > tmax) reject = true;
        if (tmax < 0.f) reject = true;
        if (!reject) {
            dist = tmin;
            return true;
        }
        dist = 0.f;
        return false;
    }

    static void CreateMerged(BoundingBox& out, const BoundingBox& b1, const BoundingBox& b2) {
        const float c1[3] = {b1.Center.x, b1.Center.y, b1.Center.z}, e1[3] = {b1.Extents.x, b1.Extents.y, b1.Extents.z};
        const float c2[3] = {b2.Center.x, b2.Center.y, b2.Center.z}, e2[3] = {b2.Extents.x, b2.Extents.y, b2.Extents.z};
        float cc[3], ee[3];
        for (int k = 0; k < 3; ++k) {
            const float mn1 = c1[k] - e1[k], mn2 = c2[k] - e2[k];
            const float mx1 = c1[k] + e1[k], mx2 = c2[k] + e2[k];
            const float mn = mn1 < mn2 ? mn1 : mn2;
            const float mx = mx1 > mx2 ? mx1 : mx2;
            cc[k] = (mn + mx) * 0.5f;
            ee[k] = (mx - mn) * 0.5f;
        }
        out.Center = XMFLOAT3(cc[0], cc[1], cc[2]);
        out.Extents = XMFLOAT3(ee[0], ee[1], ee[2]);
    }
};

// ------------------------------------------------------- elementary-function contract
// The reference calls MSVC's sinf/cosf/powf/tanf (quasi-random.cpp:28-29,45-47,58-59;
// XMVectorPow -> powf; camera.cpp:15).  Their bits are not reproducible here or on the GPU, so
// the path defines its own: IEEE binary64 kernels -- explicit fused multiply-adds (std::fma: one
// rounding; the first step of each Horner chain is a separate multiply and add), lookups into math_tables.inc (generated by tools/gen_math_tables.py; the table values
// are part of the contract) -- rounded once to binary32.  Accuracy ~1e-16 relative before the final
// rounding (i.e. correctly rounded except on ~1e-8 of inputs).  The HIP kernels implement the SAME
// operation sequence independently (cpuraytracer_amd/csrc/rt_device_math.h);
// tests/test_gpu_parity.py compares the two bit for bit.  Domain: x >= 0 for sin/cos/tan,
// base >= 0 and finite exponent for pow -- all call sites satisfy it.
// (Round 1 used fdlibm-style series without fma and a long division: 2.5x the f64 operations.)

#include "math_tables.inc"

inline double f64_from_bits(uint64_t b) { double d; std::memcpy(&d, &b, 8); return d; }
inline uint64_t f64_bits(double d) { uint64_t b; std::memcpy(&b, &d, 8); return b; }

inline void sincos_f64(float xf, double& s_out, double& c_out) {
    const double THIRTYTWO_OVER_PI = 10.185916357881301;
    const double STEP_HI = f64_from_bits(0x3FB921FB54442D18ULL);  // pi/32, leading 53 bits
    const double STEP_LO = f64_from_bits(0x3C51A62633145C07ULL);  // pi/32 - STEP_HI
    const double x = (double)xf;
    const double jd = std::nearbyint(x * THIRTYTWO_OVER_PI);  // round to nearest even (default rounding mode)
    const int j = (int)jd & 63;
    double r = std::fma(-jd, STEP_HI, x);
    r = std::fma(-jd, STEP_LO, r);
    const double z = r * r;
    // sin r = r + r z (-1/6 + z (1/120 - z/5040)),  cos r = 1 + z (-1/2 + z (1/24 + z (-1/720 + z/40320))),  |r| <= pi/64
    double ps = z * -1.984126984126984e-04 + 8.333333333333333e-03;
    ps = std::fma(z, ps, -1.6666666666666666e-01);
    const double sr = std::fma(r * z, ps, r);
    double pc = z * 2.48015873015873e-05 + -1.388888888888889e-03;
    pc = std::fma(z, pc, 4.1666666666666664e-02);
    pc = std::fma(z, pc, -0.5);
    const double cr = std::fma(z, pc, 1.0);
    const double S = f64_from_bits(kSinCosTabBits[j][0]);  // sin(j pi/32)
    const double C = f64_from_bits(kSinCosTabBits[j][1]);  // cos(j pi/32)
    s_out = std::fma(C, sr, S * cr);
    c_out = std::fma(-S, sr, C * cr);
}
inline float rt_sinf(float x) { double s, c; sincos_f64(x, s, c); return (float)s; }
inline float rt_cosf(float x) { double s, c; sincos_f64(x, s, c); return (float)c; }
inline float rt_tanf(float x) { double s, c; sincos_f64(x, s, c); return (float)(s / c); }

// pow(x,y) = 2^(y*log2 x), x >= 0.
inline float rt_powf(float xf, float yf) {
    if (yf == 0.f) return 1.f;
    if (!(xf > 0.f)) return 0.f;  // zero (and, off-domain, negative/NaN) base
    if (xf == 1.f) return 1.f;
    const double x = (double)xf;
    if (yf == 5.f) {  // Schlick's (1 - n.v)^5 (material.cpp:27,79; light.cpp:37): three exact-order f64 products
        const double x2 = x * x;
        const double x4 = x2 * x2;
        return (float)(x4 * x);
    }
    const uint64_t bits = f64_bits(x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    double m = f64_from_bits((bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);  // [1,2)
    if (m > 1.41421356237309514547) {
        m = m * 0.5;
        e = e + 1;
    }
    // table cell of m in [sqrt(1/2), sqrt(2)): centre c = 1 + idx/64; r ~ 1/c and l = -log2 r from the table
    const int idx = (int)std::nearbyint((m - 1.0) * 64.0);
    const double r = f64_from_bits(kLogTabBits[idx + 19][0]);
    const double l = f64_from_bits(kLogTabBits[idx + 19][1]);
    const double u = std::fma(m, r, -1.0);
    // log1p(u) = u + u^2 (-1/2 + u/3 - u^2/4 + u^3/5 - u^4/6 + u^5/7 - u^6/8 + u^7/9)
    double p = u * 1.1111111111111111e-01 + -0.125;
    p = std::fma(u, p, 1.4285714285714285e-01);
    p = std::fma(u, p, -1.6666666666666666e-01);
    p = std::fma(u, p, 0.2);
    p = std::fma(u, p, -0.25);
    p = std::fma(u, p, 3.3333333333333331e-01);
    p = std::fma(u, p, -0.5);
    const double lp = std::fma(u * u, p, u);
    const double log2x = std::fma(lp, 1.4426950408889634, (double)e + l);  // 1/ln 2
    const double t = (double)yf * log2x;
    if (t < -160.0) return 0.f;
    if (t > 160.0) return INFINITY;
    const double kd = std::nearbyint(t);
    const double f = t - kd;
    const double jd = std::nearbyint(f * 32.0);
    const double h = std::fma(jd, -0.03125, f);
    const double g = h * 0.69314718055994531;  // ln 2
    // e^g, |g| <= 0.0109: Taylor to g^6/720
    double q = g * 1.3888888888888889e-03 + 8.3333333333333332e-03;
    q = std::fma(g, q, 4.1666666666666664e-02);
    q = std::fma(g, q, 1.6666666666666666e-01);
    q = std::fma(g, q, 0.5);
    q = std::fma(g, q, 1.0);
    q = std::fma(g, q, 1.0);
    const double w = f64_from_bits(kExp2TabBits[(int)jd + 16]) * q;
    return (float)std::ldexp(w, (int)kd);
}

// XMVectorPow: scalar powf per lane.
inline XMVECTOR XMVectorPow(XMVECTOR a, XMVECTOR b) {
    return {rt_powf(a.x, b.x), rt_powf(a.y, b.y), rt_powf(a.z, b.z), rt_powf(a.w, b.w)};
}

}  // namespace orc

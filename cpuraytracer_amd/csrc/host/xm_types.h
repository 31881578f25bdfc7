// xm_types.h — the DirectXMath value types the reference's class API is written in, as plain PODs of
// the same names and layout: XMVECTOR (16-byte float4), XMFLOAT2/3, XMCOLOR (32-bit
// ARGB, B in the low byte).  Only the handful of operations the once-per-scene host code needs are
// provided (InitCamera/InitScene/Camera::Camera); their arithmetic is rt_device_math.h's, i.e. the
// same source the kernels compile.  Nothing here is on the render path.
#pragma once

#include <cstdint>

#include "../rt_device_math.h"

struct alignas(16) XMVECTOR {
    float x, y, z, w;
};
using XMVECTORF32 = XMVECTOR;
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

constexpr float XM_PI = 3.141592654f;

inline XMVECTOR XMVectorSet(float x, float y, float z, float w) { return XMVECTOR{x, y, z, w}; }
inline XMVECTOR XMVectorReplicate(float v) { return XMVECTOR{v, v, v, v}; }
inline XMVECTOR XMVectorZero() { return XMVECTOR{0.f, 0.f, 0.f, 0.f}; }
inline float XMVectorGetX(XMVECTOR v) { return v.x; }
#define XM_One XMVectorReplicate(1.f)  // stdafx.h:34
#define XM_Zero XMVectorZero()         // stdafx.h:35

inline XMVECTOR operator+(XMVECTOR a, XMVECTOR b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline XMVECTOR operator-(XMVECTOR a, XMVECTOR b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline XMVECTOR operator*(float s, XMVECTOR a) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline XMVECTOR operator*(XMVECTOR a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

inline rtd::V3 ToV3(XMVECTOR v) { return rtd::V3{v.x, v.y, v.z}; }
inline XMVECTOR FromV3(rtd::V3 v, float w = 0.f) { return XMVECTOR{v.x, v.y, v.z, w}; }
inline XMVECTOR XMVector3Length(XMVECTOR v) { return XMVectorReplicate(__builtin_sqrtf(rtd::dot3(ToV3(v), ToV3(v)))); }
inline XMVECTOR XMVector3Normalize(XMVECTOR v) { return FromV3(rtd::normalize3(ToV3(v))); }
inline XMVECTOR XMVector3Cross(XMVECTOR a, XMVECTOR b) { return FromV3(rtd::cross3(ToV3(a), ToV3(b))); }

// XMCOLOR(r,g,b,a) = XMStoreColor: saturate, *255, round to nearest even, pack ARGB.
struct XMCOLOR {
    uint32_t c;
    XMCOLOR() = default;
    XMCOLOR(float r, float g, float b, float a) { c = (rtd::rne_u8(a) << 24) | (rtd::rne_u8(r) << 16) | (rtd::rne_u8(g) << 8) | rtd::rne_u8(b); }
};
// XMLoadColor: byte * (1/255) per channel, (R,G,B,A).
inline XMVECTOR XMLoadColor(const XMCOLOR* s) {
    const float k = 1.0f / 255.0f;
    return {(float)((s->c >> 16) & 0xff) * k, (float)((s->c >> 8) & 0xff) * k, (float)(s->c & 0xff) * k, (float)((s->c >> 24) & 0xff) * k};
}

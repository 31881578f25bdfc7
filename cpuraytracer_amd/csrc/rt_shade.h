// rt_shade.h — hit processing: textures, Material::Scatter (material.cpp:20-164), DirectionalLight::Shade (light.cpp:11-42)
// and the exact shadow index that answers its any-hit query.
#pragma once

#include "rt_scan.h"  // root_possible(), wave helpers

namespace rtd {

// --------------------------------------------------------- textures (A14), getters (A13)
// Material record held in registers (loaded as three 16-byte reads; a by-value struct copy would
// be demoted to scratch/LDS by the compiler).
struct Mat {
    uint32_t type, tex_type;
    float smoothness, ior, tiling;
    float rgb0[3], rgb1[3];
    float luminance;
};
RT_DEV Mat load_material(const rt_material* tab, int idx) {
    const float4* q = reinterpret_cast<const float4*>(tab) + (size_t)idx * 3;
    const float4 a = q[0], b = q[1], c = q[2];
    Mat m;
    m.type = __float_as_uint(a.x); m.tex_type = __float_as_uint(a.y); m.smoothness = a.z; m.ior = a.w;
    m.tiling = b.x; m.rgb0[0] = b.y; m.rgb0[1] = b.z; m.rgb0[2] = b.w;
    m.rgb1[0] = c.x; m.rgb1[1] = c.y; m.rgb1[2] = c.z; m.luminance = c.w;
    return m;
}
// The packed record (rt_capi.hip PackMaterials): w0 = type | tex_type << 2 | rgb0 bytes << 8, w1 = rgb1 bytes, w2 = smoothness (the
// luminance of an Emissive, whose smoothness nobody reads), w3 = the ior of a DielectricTransparent, else the tiling (a glass
// sphere has no texture).  A colour is byte * (1 / 255) in binary32: XMLoadColor as the path's contract restates it, the same
// product the host formed -- the upload packs a scene only when every colour it holds IS such a product.
RT_DEV Mat load_material16(const uint4* tab, int idx) {
    RT_SITE(H_MAT16);
    const uint4 q = tab[idx];
    const float k = 1.0f / 255.0f;
    Mat m;
    m.type = q.x & 3u;
    m.tex_type = (q.x >> 2) & 1u;
    m.smoothness = __uint_as_float(q.z);
    m.luminance = __uint_as_float(q.z);
    m.ior = __uint_as_float(q.w);
    m.tiling = __uint_as_float(q.w);
    m.rgb0[0] = (float)((q.x >> 8) & 255u) * k; m.rgb0[1] = (float)((q.x >> 16) & 255u) * k; m.rgb0[2] = (float)(q.x >> 24) * k;
    m.rgb1[0] = (float)(q.y & 255u) * k; m.rgb1[1] = (float)((q.y >> 8) & 255u) * k; m.rgb1[2] = (float)((q.y >> 16) & 255u) * k;
    return m;
}
RT_DEV V3 eval_texture(const Mat& m, float u, float v) {
    if (m.tex_type == RT_TEX_CHECKER) {  // texture.cpp:20-33
        const int iu = (int)(m.tiling * u);
        const int iv = (int)(m.tiling * v);
        if (iu % 2 == iv % 2) return v3(m.rgb0[0], m.rgb0[1], m.rgb0[2]);
        return v3(m.rgb1[0], m.rgb1[1], m.rgb1[2]);
    }
    return v3(m.rgb0[0], m.rgb0[1], m.rgb0[2]);  // texture.cpp:8-11
}

// Source of the material draws: the path's xoshiro stream, or (unit tests) scripted uniforms.
struct StreamDraws {
    Rng rng;
    RT_DEV float next() { return rng_uniform(rng); }
};
struct ScriptedDraws {
    float d[3];
    uint32_t used;
    RT_DEV float next() {
        const float v = used < 3 ? d[used] : 0.f;
        ++used;
        return v;
    }
};

// ------------------------------------------------- hit processing (A8, A10-A13, A15)
// Runs Material::Scatter (draws first, material.cpp) then DirectionalLight::Shade's unoccluded
// value (light.cpp:21-40).  Outputs: scattered flag, attenuation, scattered direction, local =
// Emit + Shade with the sun visible and localOccluded = Emit + 0 (the caller adds one of the two once the
// shadow scan has decided; Emit is non-zero only for Emissive spheres, which never scatter).
template <class Draws>
RT_DEV bool scatter_only(const Mat& m, V3 rd, V3 nrm, Draws& draws, V3& atten, V3& outDir, V3& tex, const MathTabs& mt = default_math_tabs(),
                         uint32_t sampler = 0u) {
    RT_SITE(H_SCATTER);
    const float uvx = 0.5f * nrm.x + 0.5f;  // Sphere::ComputeUV, ray-tracing.cpp:26-40
    const float uvy = 0.5f * nrm.z + 0.5f;
    tex = eval_texture(m, uvx, uvy);
    bool scattered = false;
    atten = v3(1.f, 1.f, 1.f);
    // Every material's scattered direction is XMVector3Normalize of something, and three of the five cases normalise
    // XMVector3Reflect(ray.direction, hit.normal): the branches below only choose the un-normalised vector, and one
    // reflect / one normalise run for all lanes of the wave afterwards (same function of the same inputs: same bits).
    V3 raw = v3(0.f, 0.f, 0.f);
    const V3 mirror = reflect3(rd, nrm);
    const float ndv = dot3(-rd, nrm);  // material.cpp:22,74

    if (m.type == RT_MAT_DIELECTRIC_TRANSPARENT) {  // material.cpp:111-164
        RT_SITE(H_TRANSPARENT);
        const float dn = dot3(rd, nrm);
        V3 outwardNormal;
        float niOverNt, cosI;
        if (dn > 0.f) {
            outwardNormal = -nrm;
            niOverNt = m.ior;
            cosI = dot3(rd, nrm);
        } else {
            outwardNormal = nrm;
            niOverNt = 1.0f / m.ior;  // XMVectorReciprocalEst restated exact
            cosI = dot3(rd, -nrm);
        }
        const V3 refr = refract3(rd, outwardNormal, niOverNt);
        const bool canRefract = (refr.x != 0.f) || (refr.y != 0.f) || (refr.z != 0.f);
        const float prob = canRefract ? fresnel_term(cosI, m.ior) : 1.f;
        const float u = draws.next();
        raw = prob > u ? mirror : refr;
        scattered = true;
    } else if (m.type == RT_MAT_METAL) {  // material.cpp:72-103
        RT_SITE(H_METAL);
        if (ndv > 0.f) {
            // The 4-lane coin (XMVectorGreaterR + AnyTrue) is always true: lane w of f0 is the
            // colour's alpha = 1, so R.w = 1 > u.  The draw is still consumed (material.cpp:82).
            (void)draws.next();
            atten = tex;
            raw = mirror;
            scattered = true;
        }
    } else if (m.type == RT_MAT_DIELECTRIC_OPAQUE) {  // material.cpp:20-65
        if (ndv > 0.f) {
            RT_SITE(H_OPAQUE);
            const float nDotV = sat1(ndv);
            const float refl = 0.04f + (1.f - 0.04f) * rt_powf(1.f - nDotV, 5.f);
            const float u = draws.next();
            if (refl > u) {
                atten = v3(1.f, 1.f, 1.f);
                raw = mirror;
            } else {
                RT_SITE(H_OPAQUE_DIFFUSE);
                atten = tex;
                const float u1 = draws.next();  // HaltonSampleHemisphere's two dimensions
                const float u2 = draws.next();
                // quasi-random.cpp:41: uniform in solid angle (z = u1); the flagged variant is cosine weighted
                const bool cosine = (sampler & RT_SAMPLER_COSINE_HEMISPHERE) != 0u;
                const float r = sqrt_rn(cosine ? u1 : 1.f - u1 * u1);
                const float phi = (2.f * 3.141592654f) * u2;
                double sn, cs;
                sincos_f64(phi, sn, cs, mt);
                const float hx = r * (float)cs, hy = r * (float)sn, hz = cosine ? sqrt_rn(1.f - u1) : u1;
                const V3 b3 = nrm;
                const V3 up = __builtin_fabsf(nrm.x) < 0.5f ? v3(1.f, 0.f, 0.f) : v3(0.f, 1.f, 0.f);
                const V3 b1 = cross3(up, b3);
                const V3 b2 = cross3(b3, b1);
                raw = (hx * b1 + hy * b2) + hz * b3;
            }
            scattered = true;
        }
    }
    // (a lane that does not scatter normalises a unit vector instead of zero: its result is discarded, and a zero length in ANY
    // lane would send the whole wave through the slow square root -- a quarter of the waves of the cover scene hold such a lane)
    outDir = normalize3(scattered ? raw : v3(1.f, 0.f, 0.f));
    if (!scattered) outDir = v3(0.f, 0.f, 0.f);
    return scattered;
}

// DirectionalLight::Shade's unoccluded value for the light described by p (light.cpp:21-40; viewOrigin is always the camera
// origin, spheres-app.cpp:250).
template <class P>
RT_DEV V3 shade_only(const P& p, const Mat& m, V3 tex, V3 pos, V3 nrm, const MathTabs& mt = default_math_tabs()) {
    RT_SITE(H_SHADE);
    // Material getters (material.h:26-29,42-45,59-62,76-79)
    V3 albedo = v3(0.f, 0.f, 0.f), f0 = v3(0.04f, 0.04f, 0.04f);
    if (m.type == RT_MAT_DIELECTRIC_OPAQUE) albedo = tex;
    else if (m.type == RT_MAT_METAL) f0 = tex;
    else if (m.type == RT_MAT_EMISSIVE) f0 = v3(0.f, 0.f, 0.f);
    const float smooth = (m.type == RT_MAT_EMISSIVE) ? 0.f : m.smoothness;

    const V3 L = v3(p.sun_dir[0], p.sun_dir[1], p.sun_dir[2]);
    const float nDotL = sat1(dot3(nrm, L));
    const V3 radianceIn = v3(p.sun_rad[0] * nDotL, p.sun_rad[1] * nDotL, p.sun_rad[2] * nDotL);
    const V3 viewDir = normalize3(v3(p.cam_o[0], p.cam_o[1], p.cam_o[2]) - pos);
    const V3 halfVector = normalize3(L + viewDir);
    const float nDotH = sat1(dot3(nrm, halfVector));
    const float nDotV2 = sat1(dot3(viewDir, nrm));
    const float p5 = rt_powf(1.f - nDotV2, 5.f);
    const float ps = rt_powf(nDotH, smooth, mt);
    const V3 one = v3(1.f, 1.f, 1.f);
    const V3 reflectance = f0 + (one - f0) * p5;
    const V3 spec = ((reflectance * 0.125f) * (smooth + 8.f)) * ps;
    return radianceIn * (albedo + spec);
}

// Emit + DirectionalLight::Shade's unoccluded value (light.cpp:21-40) and Emit + 0 (its value when occluded).
template <class P>
RT_DEV void shade_value(const P& p, const Mat& m, V3 tex, V3 pos, V3 nrm, bool wantShade, V3& local, V3& localOccluded,
                        const MathTabs& mt = default_math_tabs()) {
    RT_SITE(H_SHADEV);
    V3 emit = v3(0.f, 0.f, 0.f);
    if (m.type == RT_MAT_EMISSIVE) emit = m.luminance * tex;  // material.cpp:172-175; 0 for every other material
    localOccluded = emit + v3(0.f, 0.f, 0.f);  // Shade returns XM_Zero when the sun is occluded (light.cpp:15-18)
    local = localOccluded;
    if (wantShade) local = emit + shade_only(p, m, tex, pos, nrm, mt);
}

// Scatter, then Emit + Shade with the sun assumed visible (the scan-based shadow path decides later).
template <class P, class Draws>
RT_DEV bool scatter_and_shade(const P& p, const Mat& m, V3 rd, V3 pos, V3 nrm, Draws& draws, V3& atten, V3& outDir,
                              V3& local, V3& localOccluded, uint32_t sampler = 0u) {
    V3 tex;
    const bool scattered = scatter_only(m, rd, nrm, draws, atten, outDir, tex, default_math_tabs(), sampler);
    shade_value(p, m, tex, pos, nrm, true, local, localOccluded);
    return scattered;
}

// ------------------------------------------------------------- shadow rays (A13) without a scan
// DirectionalLight::Shade asks whether ANY sphere yields an acceptable root for the ray (hit.pos, L)
// (light.cpp:13-15 -> BvhNode::Intersect used as any-hit).  L is the same for every shadow ray, so the
// host bins the spheres by their footprint (a disc) in the plane perpendicular to L; a query evaluates
// Sphere::Intersect's reference-order arithmetic only for the spheres whose inflated footprint covers
// the point's cell, plus a short list of spheres that cover much of the grid (the floor).  A sphere the
// reference test accepts has its centre within sqrt(r^2 + E/a) of the ray's line, E <= 16 eps a (2|p|^2 +
// 2|c|^2 + r^2); the footprints are inflated for that with |p| <= P0 (and for the rounding of the
// projection), so inside that radius the answer is exactly the reference's.  Points farther out fall
// back to the shadow scan.
// num / a > 0.001 (ray-tracing.cpp:56,69 with the bias test of :52) for the shadow rays, whose a = |L|^2 is the same for
// all of them: ya = recip_rn(a) once per kernel, then Markstein's correction (rt_device_math.h) instead of an 11-operation
// division.  Only the COMPARISON matters: for |num| < 2^100 and a in [2^-19, 2^100] the quotient is exact wherever it is
// normal -- in particular around 0.001 -- and where it underflows both forms are far below the bias.  aOk == false (or a
// huge numerator) takes the plain division.
RT_DEV bool root_exceeds_bias(float num, float a, float ya, bool aOk) {
#if RT_MARKSTEIN
    if (__builtin_expect(aOk && __builtin_fabsf(num) < 0x1p100f, 1)) {
        const float q0 = num * ya;
        const float r = __builtin_fmaf(-a, q0, num);
        return __builtin_fmaf(r, ya, q0) > 0.001f;
    }
#endif
    return num / a > 0.001f;
}
RT_DEV bool sphere_any_hit(const float4 S, V3 o, V3 d, float a, float ya, bool aOk) {
    const float ocx = o.x - S.x;
    const float ocy = o.y - S.y;
    const float ocz = o.z - S.z;
    const float b = (ocx * d.x + ocy * d.y) + ocz * d.z;
    const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - S.w;
    const float disc = b * b - a * cc;
    if (disc > 0.f) {  // ray-tracing.cpp:54-71
        const float sq = sqrt_rn(disc);
        if (root_exceeds_bias(-b - sq, a, ya, aOk)) return true;
        if (root_exceeds_bias(-b + sq, a, ya, aOk)) return true;
    }
    return false;
}

// Two phases, like the closest-hit resolve: first the discriminants of every listed sphere (cheap, uniform), keeping
// up to four spheres whose roots are possible in a register queue; then roots (sqrt + divides) only for those, until
// one occludes.  root_possible() is exact, so the answer is the reference's any-hit over the same spheres.
template <class P>
RT_DEV bool shadow_query(const P& p, const float4* __restrict__ tab, const uint16_t* __restrict__ cellStart,
                         const uint16_t* __restrict__ entries, const uint16_t* __restrict__ glob, bool globInLds,
                         const float4* globSph, const uint16_t* globIds, const float4* __restrict__ entrySph, V3 pos, V3 L, float aL) {
    RT_SITE(H_SHADOWQ);
    bool occluded = false;
    unsigned long long queue = 0ull;
    uint32_t nq = 0;
#if RT_MARKSTEIN
    const bool aOk = (unsigned)(float_exponent(aL) + 19) <= 119u;  // wave-uniform, loop-invariant: a of the sun direction
    const float yaL = recip_rn(aL);
#else
    const bool aOk = false;
    const float yaL = 0.f;
#endif
    // (lambdas, not macros: every statement keeps its own source line, which tools/phase_budget.py attributes instructions by)
    auto consider = [&](uint32_t id, const float4 S) __attribute__((always_inline)) {
        RT_SITE(H_SQ_CONSIDER);
        const float ocx = pos.x - S.x;
        const float ocy = pos.y - S.y;
        const float ocz = pos.z - S.z;
        const float b = (ocx * L.x + ocy * L.y) + ocz * L.z;
        const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - S.w;
        const float disc = b * b - aL * cc;
        if (root_possible(disc, b)) {
            if (nq < 4u) {
                queue = (queue << 16) | (unsigned long long)id;
                ++nq;
            } else {
                RT_SITE(H_SQ_FULL);
                occluded = occluded || sphere_any_hit(S, pos, L, aL, yaL, aOk);  // queue full (rare): evaluate now
            }
        }
    };
    // two entries per round: both ids, then both spheres, are requested together -- one entry at a time costs two LDS round
    // trips back to back (id, then sphere) per entry with nothing to do in between
    // sph: the spheres of the list's elements side by side with the ids (large scenes: rt_params.h sg_sph), or null: tab[id]
    auto walk = [&](const uint16_t* __restrict__ list, const float4* __restrict__ sph, uint32_t first, uint32_t end) __attribute__((always_inline)) {
        RT_SITE(H_SQ_WALK);
        uint32_t e = first;
        for (; e + 2u <= end; e += 2u) {
            RT_SITE(H_SQ_ROUND);
            const uint32_t idA = list[e], idB = list[e + 1u];
            const float4 SA = sph ? sph[e] : tab[idA], SB = sph ? sph[e + 1u] : tab[idB];
            consider(idA, SA);
            consider(idB, SB);
        }
        if (e < end) {
            RT_SITE(H_SQ_TAIL1);
            const uint32_t idA = list[e];
            consider(idA, sph ? sph[e] : tab[idA]);
        }
    };
    if (globInLds) {
        // the global list from its LDS copy (rt_params.h sg_glob_slots): spheres and ids read directly, two per round
        uint32_t k = 0;
        for (; k + 2u <= p.sg_nglobal; k += 2u) {
            RT_SITE(H_SQ_GROUND);
            const float4 SA = globSph[k], SB = globSph[k + 1u];
            const uint32_t idA = globIds[k], idB = globIds[k + 1u];
            consider(idA, SA);
            consider(idB, SB);
        }
        if (k < p.sg_nglobal) {
            RT_SITE(H_SQ_GTAIL);
            consider(globIds[k], globSph[k]);
        }
    } else {
        walk(glob, nullptr, 0u, p.sg_nglobal);
    }
    const float u = dot3(pos, v3(p.sg_e1[0], p.sg_e1[1], p.sg_e1[2]));
    const float v = dot3(pos, v3(p.sg_e2[0], p.sg_e2[1], p.sg_e2[2]));
    const float fx = (u - p.sg_u0) * p.sg_inv_cell, fy = (v - p.sg_v0) * p.sg_inv_cell;
    if (fx >= 0.f && fy >= 0.f && fx < (float)p.sg_nx && fy < (float)p.sg_ny) {
        RT_SITE(H_SQ_CELL);
        const uint32_t c = (uint32_t)fy * p.sg_nx + (uint32_t)fx;
        walk(entries, entrySph, (uint32_t)cellStart[c], (uint32_t)cellStart[c + 1]);
    }
    while (nq > 0u && !occluded) {
        RT_SITE(H_SQ_ROOTS);
        const uint32_t id = (uint32_t)(queue & 0xffffull);
        queue >>= 16;
        --nq;
        occluded = sphere_any_hit(tab[id], pos, L, aL, yaL, aOk);
    }
    return occluded;
}

// The any-hit of one shadow ray over EVERY scan entry (the reference's IsOccluded, light.cpp:13-15): the multi-light path's answer
// for hit points outside a light's index (|p| > P0; rare).  Padding entries (r*r = -1e30) never hit.
RT_DEV bool any_hit_all(const float4* __restrict__ tab, uint32_t nEntries, V3 pos, V3 L, float aL) {
#if RT_MARKSTEIN
    const bool aOk = (unsigned)(float_exponent(aL) + 19) <= 119u;
    const float yaL = recip_rn(aL);
#else
    const bool aOk = false;
    const float yaL = 0.f;
#endif
    bool occluded = false;
    for (uint32_t e = 0; e < nEntries && !occluded; ++e) occluded = sphere_any_hit(tab[e], pos, L, aL, yaL, aOk);
    return occluded;
}

}  // namespace rtd

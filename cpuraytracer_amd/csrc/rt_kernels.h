// rt_kernels.h — HIP kernels of the render-loop hot path for MI355X (gfx950 / CDNA4).
//
// What runs where (reference file:line -> kernel):
//   GenerateRays            spheres-app.cpp:132-161  -> gen_primary_ray()           (64 paths at a time into a per-wave LDS cache)
//   for_each(par) trace     spheres-app.cpp:177-184  -> rt_trace_kernel             (persistent-threads megakernel)
//   GetHitColor recursion   spheres-app.cpp:238-257  -> per-lane state machine in rt_trace_kernel
//   BvhNode/Sphere::Intersect ray-tracing.cpp:42-84,174-214 -> scan_list_mfma() (filter + pooled resolve) / scan_list_deferred()
//   Material::Scatter x3    material.cpp:20-164      -> scatter_only()
//   DirectionalLight::Shade light.cpp:11-42          -> shadow_query() (exact footprint index) + shade_value()
//   hdr[id] += L*exposure   spheres-app.cpp:182-183  -> rt_accumulate_kernel        (ordered in s)
//   tonemap transform(par)  spheres-app.cpp:196-214  -> rt_resolve_kernel
//
// Design (DESIGN.md §5 has the long form and the conservativeness argument):
//   * one work-item per (pixel, sample) PATH; a wave keeps 64 paths in flight and refills finished
//     lanes (ballot + prefix count) from a per-wave LDS cache of 64 prepared paths that all 64 lanes
//     generate together from the wave's block of a global queue, so lanes stay full despite path
//     lengths of 1..51 closest-hit scans;
//   * shadow rays are answered at the hit by an exact footprint index (spheres binned in the plane
//     perpendicular to the sun); far hit points fall back to a second scan ("needs shadow scan"
//     lane state) — the same booleans as the reference's any-hit either way;
//   * the scan is a conservative FILTER followed by an EXACT resolve.  Spheres are stored in k-d
//     groups of four with a bounding sphere each; "which ray may hit which group" is a K = 4 dense
//     contraction evaluated on the matrix cores with split-bf16 operands
//     (v_mfma_f32_32x32x16_bf16); rejected bits are shifted into register bitmaps, the wave's
//     (ray, group) pairs are pooled in an LDS work list, filtered per sphere with the same
//     conservative formula, and the surviving (ray, sphere) pairs evaluate Sphere::Intersect in the
//     reference's operation order, merged per ray with an LDS 64-bit minimum.  The image equals an
//     exhaustive scan bit for bit; scene tables, operand image, work lists and scene constants live
//     in LDS (measured: VALU ops with SGPR sources issue at half rate on gfx950, so scene data must
//     arrive in VGPRs);
//   * xoshiro128** state in 4 VGPRs per lane;
//   * per-path results go to an HBM sample buffer [tile of 64 pixels][sample][pixel] (12 B each) and are summed in
//     increasing s by rt_accumulate_kernel: the reference's summation order, bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_api.h"
#include "rt_device_math.h"

namespace rtd {

// Diagnostic build only (-DRT_STAMPS): per-section shader-clock sums go to g_dbg[], which no kernel reads.
// The shipped library is built without it.
#ifdef RT_STAMPS
static __device__ unsigned long long g_dbg[20];  // refill, scan, transitions, iterations, filter, resolve, resolve items, max items,
                                                 // phase A cycles, phase B cycles, A iterations, B iterations, scatter, shadow query, shade, -
RT_DEV unsigned long long rt_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define RT_STAMP(var) const unsigned long long var = rt_stamp()
#define RT_ACC(sum, a, b) sum += (b) - (a)
#else
#define RT_STAMP(var)
#define RT_ACC(sum, a, b)
#endif

constexpr int kWaveSize = 64;
constexpr uint32_t kMaxLevels = 6;  // levels of group bounds (4-ary): 128 * 4^5 groups at most
constexpr uint32_t kQueueBlock = 256;  // paths a wave takes from the global queue per atomic

struct TraceParams {
    // scene
    // Spheres are stored CLUSTERED: groups of four spatially close spheres (Morton order; large spheres alone),
    // each group with a conservative bounding sphere.  Results do not depend on the order: the closest-hit
    // update breaks ties by the ORIGINAL list index (orig[]).
    const float4* scan;        // [n_padded] cx, cy, cz, r*r in clustered order (padding: never-hit entries, r*r = -1e30)
    const uint32_t* orig;      // [n_padded] original list index of each entry (0xffffffff for padding)
    const float4* leaf;        // [n_padded] conservative one-sphere bounds (cx, cy, cz, |c|^2 - rf^2) for the sphere-level filter
    // Bounds hierarchy (4-ary): level 0 = the groups, level k+1 node j = level-k nodes 4j..4j+3; the top level
    // (<= 128 nodes) is filtered on the matrix cores, lower levels are descended per lane.
    const float4* tree;        // all levels, level 0 first: Cx, Cy, Cz, |C|^2 - Rf^2
    uint32_t level_off[kMaxLevels], level_cnt[kMaxLevels];
    uint32_t n_levels;
    uint32_t tree_in_lds;      // tree mode: stage every level of bounds into LDS (else the descent reads them through L2)
    const float* radius;       // [n_padded] per scan entry (clustered order)
    const rt_material* mats;   // [n_padded] per scan entry: material i of the reference belongs to sphere i = orig[entry]
    uint32_t n;                // real spheres
    uint32_t n_groups;         // groups of four entries (even)
    uint32_t mats_in_lds;      // stage the material table into LDS (else it is read through L2)
    float bound_norm;          // max over groups of |C| + R (scale of the filter's behind-the-origin threshold)
    uint32_t n_padded;         // 4 * n_groups + 4
    // Exact shadow index for the (single, directional) sun: spheres binned by their footprint in the plane
    // perpendicular to the light.  Valid for hit points with |p|^2 <= sg_p0sq.
    const uint16_t* sg_cell_start;  // [sg_nx * sg_ny + 1]
    const uint16_t* sg_entries;     // clustered entry indices per cell
    const uint16_t* sg_global;      // entries tested for every query (footprints covering much of the grid)
    uint32_t sg_nx, sg_ny, sg_nglobal, sg_nentries, sg_enabled, sg_in_lds;
    float sg_e1[3], sg_e2[3], sg_u0, sg_v0, sg_inv_cell, sg_p0sq;
    float cam_o[3], cam_x[3], cam_y[3], cam_oip[3];
    float aperture, focal;
    float sun_dir[3], sun_rad[3];  // sun_rad = luminance * colour (light.cpp:27, left factor)
    float sky_emit[3];             // luminance * colour (material.cpp:172-175)
    float exposure;
    // work
    uint32_t W, H;
    rt_rowset rs;
    uint32_t s0;          // first sample index of this pass (1-based)
    uint32_t spp_pass;    // samples per pixel in this pass
    uint32_t total_paths; // W * local_rows * spp_pass, or the path-list length
    uint32_t npix_local;  // W * local_rows (the sample buffer is [sample of the pass][local pixel])
    uint32_t max_depth;
    uint64_t seed;
    const uint32_t* path_list;  // optional explicit (i, j, s) triples (unit tests)
    const float2* jitter_tab;   // [spp_pass] Halton2D(s;2,3) for s = s0.. (rt_raygen_tables_kernel), or null
    const float2* lens_tab;     // [.] HaltonSampleDisk(k;4,5) for k = lens_k0.., or null
    uint32_t lens_k0;
    float* samples;             // [total_paths][3] radiance * exposure
    uint32_t* trav_out;         // optional per-path traversal counts
    uint32_t ray_cache_off16;  // float4 offset of the per-wave prepared-path caches in dynamic LDS, 0 = no cache
    uint32_t* queue_head;       // global work counter, zeroed before launch
    unsigned long long* counters;  // [0] traversals, [1] segments
};

// The scene and pass constants the hit processing and the ray generation read, copied once per workgroup into LDS
// (rt_trace_kernel): as kernel arguments they live in SGPRs for the whole persistent loop (~60 of them, spilled to VGPR
// lanes and read back with v_readlane at every use, and an SGPR source halves the VOP2 issue rate); from LDS they are
// short-lived VGPR temporaries.  Member names equal TraceParams' so that the device functions below take either.
struct SceneConsts {
    float cam_o[3], cam_x[3], cam_y[3], cam_oip[3];
    float aperture, focal;
    float sun_dir[3], sun_rad[3];
    float sky_emit[3];
    float exposure;
    float sg_e1[3], sg_e2[3], sg_u0, sg_v0, sg_inv_cell, sg_p0sq;
    uint32_t sg_nx, sg_ny, sg_nglobal, sg_enabled;
    uint32_t W, H, s0, lens_k0;
    const float2* jitter_tab;
    const float2* lens_tab;
};
constexpr uint32_t kConstBytes = 256;  // LDS reserved for SceneConsts at the start of the dynamic image
static_assert(sizeof(SceneConsts) <= kConstBytes, "SceneConsts must fit its LDS slot");
RT_DEV void fill_consts(const TraceParams& p, SceneConsts& k) {
    for (int i = 0; i < 3; ++i) {
        k.cam_o[i] = p.cam_o[i]; k.cam_x[i] = p.cam_x[i]; k.cam_y[i] = p.cam_y[i]; k.cam_oip[i] = p.cam_oip[i];
        k.sun_dir[i] = p.sun_dir[i]; k.sun_rad[i] = p.sun_rad[i]; k.sky_emit[i] = p.sky_emit[i];
        k.sg_e1[i] = p.sg_e1[i]; k.sg_e2[i] = p.sg_e2[i];
    }
    k.aperture = p.aperture; k.focal = p.focal; k.exposure = p.exposure;
    k.sg_u0 = p.sg_u0; k.sg_v0 = p.sg_v0; k.sg_inv_cell = p.sg_inv_cell; k.sg_p0sq = p.sg_p0sq;
    k.sg_nx = p.sg_nx; k.sg_ny = p.sg_ny; k.sg_nglobal = p.sg_nglobal; k.sg_enabled = p.sg_enabled;
    k.W = p.W; k.H = p.H; k.s0 = p.s0; k.lens_k0 = p.lens_k0;
    k.jitter_tab = p.jitter_tab; k.lens_tab = p.lens_tab;
}

// --------------------------------------------------------------------------- row sets
RT_DEV uint32_t rowset_global_row(const rt_rowset& rs, uint32_t lr) {
    const uint32_t lb = lr / rs.block_rows;
    const uint32_t k = lr - lb * rs.block_rows;
    return rs.first_row + (lb * rs.nshards + rs.shard) * rs.block_rows + k;
}

// Camera::GetRay (camera.cpp:30-48)
template <class P>
RT_DEV void camera_get_ray(const P& p, float uvx, float uvy, float lensx, float lensy, V3& origin, V3& dir) {
    const V3 camO = v3(p.cam_o[0], p.cam_o[1], p.cam_o[2]);
    const V3 mx = v3(p.cam_x[0], p.cam_x[1], p.cam_x[2]);
    const V3 my = v3(p.cam_y[0], p.cam_y[1], p.cam_y[2]);
    const V3 oip = v3(p.cam_oip[0], p.cam_oip[1], p.cam_oip[2]);
    const float ndcx = 2.f * uvx - 1.f;
    const float ndcy = -2.f * uvy + 1.f;
    const V3 pp = (oip + ndcx * mx) + ndcy * my;
    const V3 focalPoint = camO + p.focal * normalize3(pp - camO);
    const float rdx = (0.5f * p.aperture) * lensx;
    const float rdy = (0.5f * p.aperture) * lensy;
    origin = (camO + rdx * mx) + rdy * my;
    dir = normalize3(focalPoint - origin);
}

// ------------------------------------------------------------------ primary rays (A1, A2)
// SpheresApp::GenerateRays (spheres-app.cpp:132-161) + Camera::GetRay (camera.cpp:30-48) for one
// (i, j, s).  jitter = Halton2D(s;2,3); lens = HaltonSampleDisk(s+i+j;4,5).
RT_DEV void halton_disk_4_5(uint32_t k, float& lensx, float& lensy) {  // quasi-random.cpp:52-61
    const float theta = (2.f * 3.141592654f) * halton(k, 4);
    const float r = halton(k, 5);
    double sn, cs;
    sincos_f64(theta, sn, cs);
    lensx = r * (float)cs;
    lensy = r * (float)sn;
}
template <class P>
RT_DEV void gen_primary_ray(const P& p, uint32_t i, uint32_t j, uint32_t s, V3& origin, V3& dir) {
    const float xsize = (float)p.W;
    const float ysize = (float)p.H;
    float jx, jy, lensx, lensy;
    const uint32_t li = s + i + j;
    if (p.jitter_tab) {
        // the radical inverses depend on s (jitter) and s+i+j (lens) only: a per-pass device kernel
        // tabulates them so that a refilled lane does two 8-byte loads instead of four divide loops
        const float2 jt = p.jitter_tab[s - p.s0];
        const float2 lt = p.lens_tab[li - p.lens_k0];
        jx = jt.x; jy = jt.y; lensx = lt.x; lensy = lt.y;
    } else {
        jx = halton(s, 2);
        jy = halton(s, 3);
        halton_disk_4_5(li, lensx, lensy);
    }
    const float uvx = ((float)(int)i + jx) / xsize;
    const float uvy = ((float)(int)j + jy) / ysize;

    camera_get_ray(p, uvx, uvy, lensx, lensy, origin, dir);
}

// ---------------------------------------------------------------------- list scan (A4, A6)
// Closest acceptable root over the whole list; equal t keeps the lower index.  Per sphere:
// Sphere::Intersect's arithmetic (ray-tracing.cpp:44-50) = 17 f32 VALU ops in VOP2 form with all
// operands in VGPRs.  Spheres are taken four at a time: the next group's four ds_read_b128 are
// issued before the current group's arithmetic (software prefetch — the per-sphere branch would
// otherwise pin every read right in front of its use), and ONE compare + branch per group guards
// the rarely needed sqrt/div root evaluation: disc > 0 for some sphere of the group implies the
// AND of the four discriminants' bit patterns has a clear sign bit (a conservative pre-filter; the
// exact `disc > 0` is re-tested per sphere inside).  The table is padded by the host to a multiple
// of eight plus one group with never-hit entries (r*r = -1e30 => disc < 0).
constexpr uint32_t kScanGroup = 4;

// Reference-order root evaluation for entry i (ray-tracing.cpp:54-71); ties keep the lower ORIGINAL index.
RT_DEV void root_test(float disc, float b, float a, uint32_t i, const uint32_t* __restrict__ orig, float& tmin, int& idx) {
    if (disc > 0.f) {  // ray-tracing.cpp:54
        const float sq = __builtin_sqrtf(disc);
        float t = (-b - sq) / a;                // :56
        if (!(t > 0.001f)) t = (-b + sq) / a;   // :58, :69-71 (bias 0.001, :52)
        if (t > 0.001f && (t < tmin || (t == tmin && idx >= 0 && orig[i] < orig[idx]))) {
            tmin = t;
            idx = (int)i;
        }
    }
}

// Plain sequential scan over every entry (unit-test kernel; the megakernel uses the filtered scans below).
RT_DEV void scan_list(const float4* __restrict__ tab, const uint32_t* __restrict__ orig, uint32_t nEntries, V3 o, V3 d, float& tmin,
                      int& idx) {
    const float a = dot3(d, d);
    tmin = __builtin_inff();
    idx = -1;
    for (uint32_t i = 0; i < nEntries; ++i) {
        const float4 S = tab[i];
        const float ocx = o.x - S.x;
        const float ocy = o.y - S.y;
        const float ocz = o.z - S.z;
        const float b = (ocx * d.x + ocy * d.y) + ocz * d.z;
        const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - S.w;
        const float disc = b * b - a * cc;
        root_test(disc, b, a, i, orig, tmin, idx);
    }
}

// ------------------------------------------------------- list scan with deferred roots
// The megakernel's scan.  Evaluating roots inside the scan loop serialises the wave over every
// sphere that ANY of its 64 (incoherent) rays might hit — ~45 mostly-idle VALU instructions per such
// sphere, about as expensive as a whole group of the branch-free arithmetic.  Instead each lane only
// RECORDS the groups whose sign test it passes (one predicated 2-byte LDS store) and, after the scan
// (or when a list is nearly full), every lane walks its OWN list: reload the group's four spheres,
// recompute the same b and disc (same operations on the same inputs => same bits), and evaluate
// roots — all lanes busy on different spheres at once.  Lists are processed in increasing sphere
// index, and the update is a strict `t < tmin`, so the result equals the sequential scan's
// (lower index wins ties).  A root is impossible — exactly, in IEEE arithmetic — when disc <= 0, or
// when b > 0 and disc < fl(b*b): then fl(sqrt(disc)) <= fl(sqrt(fl(b*b))) = b, so both numerators
// -b -/+ sqrt(disc) are <= 0 and neither root exceeds the 0.001 bias (ray-tracing.cpp:52-71).
constexpr uint32_t kCandSlots = 16;  // per-lane list capacity (uint16 group offsets), flushed when nearly full

RT_DEV bool group_sign_test(const float4 S0, const float4 S1, const float4 S2, const float4 S3, V3 o, V3 d, float a) {
    float e0, e1, e2, e3;
#define RT_DISC_ONLY(S, E)                                             \
    {                                                                  \
        const float ocx = o.x - S.x;                                   \
        const float ocy = o.y - S.y;                                   \
        const float ocz = o.z - S.z;                                   \
        const float b = (ocx * d.x + ocy * d.y) + ocz * d.z;           \
        const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - S.w;  \
        E = b * b - a * cc;                                            \
    }
    RT_DISC_ONLY(S0, e0)
    RT_DISC_ONLY(S1, e1)
    RT_DISC_ONLY(S2, e2)
    RT_DISC_ONLY(S3, e3)
#undef RT_DISC_ONLY
    const int signs = __float_as_int(e0) & __float_as_int(e1) & __float_as_int(e2) & __float_as_int(e3);
    return signs >= 0;
}

RT_DEV bool root_possible(float e, float b) { return e > 0.f && !(b > 0.f && e < b * b); }

// Exact evaluation of one recorded group (four consecutive spheres starting at g) for this lane's ray:
// Sphere::Intersect's arithmetic (ray-tracing.cpp:44-71) and the closest-hit update.  Smaller t wins;
// equal t keeps the lower ORIGINAL sphere index whatever order groups are stored or resolved in.
RT_DEV void resolve_group(const float4* __restrict__ tab, const uint32_t* __restrict__ orig, uint32_t g, V3 o, V3 d, float a, float& tmin,
                          int& idx) {
    const float4 S0 = tab[g], S1 = tab[g + 1], S2 = tab[g + 2], S3 = tab[g + 3];
    float b0, b1, b2, b3, e0, e1, e2, e3;
#define RT_DISC(S, B, E)                                               \
    {                                                                  \
        const float ocx = o.x - S.x;                                   \
        const float ocy = o.y - S.y;                                   \
        const float ocz = o.z - S.z;                                   \
        B = (ocx * d.x + ocy * d.y) + ocz * d.z;                       \
        const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - S.w;  \
        E = B * B - a * cc;                                            \
    }
    RT_DISC(S0, b0, e0)
    RT_DISC(S1, b1, e1)
    RT_DISC(S2, b2, e2)
    RT_DISC(S3, b3, e3)
#undef RT_DISC
    uint32_t m = (root_possible(e0, b0) ? 1u : 0u) | (root_possible(e1, b1) ? 2u : 0u) | (root_possible(e2, b2) ? 4u : 0u) |
                 (root_possible(e3, b3) ? 8u : 0u);
    while (m != 0u) {  // per-lane loop: all lanes evaluate one of their own candidates per iteration
        const uint32_t k = (uint32_t)__builtin_ctz(m);
        m &= m - 1u;
        const float e = k == 0u ? e0 : (k == 1u ? e1 : (k == 2u ? e2 : e3));
        const float b = k == 0u ? b0 : (k == 1u ? b1 : (k == 2u ? b2 : b3));
        const float sq = __builtin_sqrtf(e);
        float t = (-b - sq) / a;               // ray-tracing.cpp:56
        if (!(t > 0.001f)) t = (-b + sq) / a;  // :69
        const int cand = (int)(g + k);
        if (t > 0.001f && (t < tmin || (t == tmin && idx >= 0 && orig[cand] < orig[idx]))) {
            tmin = t;
            idx = cand;
        }
    }
}

// cand: this lane's column of the wave's candidate list in LDS; slot stride is 64 entries.
RT_DEV void scan_list_deferred(const float4* __restrict__ tab, const uint32_t* __restrict__ orig, uint32_t nPadded, V3 o, V3 d, float& tmin,
                               int& idx, uint16_t* cand) {
    const float a = dot3(d, d);
    tmin = __builtin_inff();
    idx = -1;
    uint32_t cnt = 0;
    uint32_t i = 0;
    for (;;) {
        // ---- record phase: branch-free discriminants, two groups per iteration on ping-pong registers
        float4 A0 = tab[i], A1 = tab[i + 1], A2 = tab[i + 2], A3 = tab[i + 3];
        bool nearlyFull = false;
        for (; i + kScanGroup < nPadded && !nearlyFull; i += 2 * kScanGroup) {
            const float4 B0 = tab[i + 4], B1 = tab[i + 5], B2 = tab[i + 6], B3 = tab[i + 7];
            if (__builtin_expect(group_sign_test(A0, A1, A2, A3, o, d, a), 0)) {
                cand[cnt * kWaveSize] = (uint16_t)i;
                ++cnt;
            }
            A0 = tab[i + 8]; A1 = tab[i + 9]; A2 = tab[i + 10]; A3 = tab[i + 11];
            if (__builtin_expect(group_sign_test(B0, B1, B2, B3, o, d, a), 0)) {
                cand[cnt * kWaveSize] = (uint16_t)(i + 4);
                ++cnt;
            }
            nearlyFull = __ballot(cnt + 2 > kCandSlots) != 0ull;
        }
        // ---- resolve phase: every lane evaluates its own candidates, in increasing sphere index
        for (uint32_t it = 0; __ballot(it < cnt) != 0ull; ++it) {
            if (it < cnt) {
                const uint32_t g = cand[it * kWaveSize];
                resolve_group(tab, orig, g, o, d, a, tmin, idx);
            }
        }
        cnt = 0;
        if (!(i + kScanGroup < nPadded)) break;
    }
}

// ------------------------------------------------ list scan with a matrix-core pre-filter
// "Which of these 64 rays can hit which of these groups" is a dense contraction: for a bounding sphere
// (C, R) and a ray (o, d), b = d.(o - C) = [d, d.o].[-C, 1] and a*cc = a|o|^2 + [-2a o, a].[C, |C|^2 - R^2]
// are K = 4 inner products of a per-ray with a per-group vector.  v_mfma_f32_32x32x2_f32 evaluates
// 32 groups x 32 rays per instruction as exact f32 FMA chains.  That is NOT the reference's rounding, so
// it is used only as a conservative FILTER over the group bounds:
//     F = b~^2 - a*cc~ + M >= 0   =>   the group is recorded for that ray
// and every recorded group is then resolved EXACTLY by resolve_group (reference-order VALU arithmetic on
// the four member spheres).  The image is bit-identical to an exhaustive scan: the filter only decides
// how much exact work is skipped.  Conservativeness (DESIGN.md §5.1 has the derivation): if the
// reference-order discriminant of a member sphere i is positive, the ray's line passes within
// sqrt(r_i^2 + E_i/a) of c_i, hence within s_i + that of C; with R >= s_i + r_i the true bound
// discriminant exceeds -(E_i + 2 s_i sqrt(a E_i)) >= -(101 E_i + 0.01 a s_i^2).  The host folds 0.01 s_max^2
// and K eps (2(|C|+R)^2 + R^2) into Rf^2 and the kernel adds 2 K eps a|o|^2 per ray, K = 2048 > 101*16 + 30
// (E_i <= 16 eps a G and the filter's own rounding <= 30 eps a G, G = 2|o|^2 + 2(|C|+R)^2 + R^2).
//
// Tile mapping (groups = rows/A, rays = columns/B): lane l supplies A[l&31][l>>5] and B[l>>5][l&31] and
// receives, for ray column l&31, the 16 group rows (r&3) + 8(r>>2) + 4(l>>5).  Rays 0-31 and 32-63 are two
// column tiles; lanes l and l^32 split each ray's groups, so every ray has two producer lanes, each
// with its own sub-list and register counter (no atomics).  A sub-list that overflows makes its ray
// fall back to resolving every group (rare; still exact).
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr uint32_t kTreeWork = 768, kTreeExact = 576, kTreeReserve = 3 * kMaxLevels;  // hierarchy scan: (ray, node) and (ray, sphere) lists
constexpr uint32_t kPoolA = 640;                                        // pooled resolve: (ray, group) items per pass
constexpr uint32_t kPoolB = 512;                                        // pooled resolve: (ray, sphere) items before a drain
constexpr uint32_t kWaveListBytes = kPoolA * 2 + kPoolB * 2 + 64 * 8;      // item pools + per-ray best keys = 2816 B per wave
constexpr uint32_t kWaveCandBytes = kTreeWork * 4 + kTreeExact * 4 + 64 * 8;  // hierarchy scan: 5888 B per wave
// K of the filter margins (units of eps * a * G; the host folds the same K into each bound): the matrix-core level needs
// 101*16 (exact-path rounding, amplified by the member offsets) + ~600 (split-bf16 operands); levels tested on the VALU
// in f32 need 101*16 + 30; a one-sphere bound (offset 0) needs 16 + 30.
constexpr float kMarginK = 4096.f, kMarginKValu = 2048.f, kMarginKLeaf = 64.f;
constexpr float kMarginRel = kMarginK * 5.9604645e-8f;                 // K * eps
constexpr uint32_t kRayCacheBytes = 64 * 48;                            // per-wave cache of prepared paths
constexpr uint32_t kOpsPerTile = 8 * 64;                                // dwords of the group operand image per 32-group tile

// Split-bf16 operands.  An f32 value v is carried as h + l with h = bf16(v) and l = bf16(v - h) (both round to
// nearest even; v - h is exact), so |v - (h + l)| <= 2^-18 |v|, and a product x*y becomes the four exact bf16 products
// xh*yh + xh*yl + xl*yh + xl*yl accumulated in f32 by the matrix core.  One v_mfma_f32_32x32x16_bf16 (K = 16) therefore
// evaluates a K = 4 inner product of f32-like operands: lane l supplies the two values k = 2(l>>5), 2(l>>5)+1 of its
// row / column, each as four K-slots.  Group side (A): (yh, yh, yl, yl); ray side (B): (xh, xl, xh, xl).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

RT_DEV uint32_t bf16_pair_bits(float lo16, float hi16) {  // bf16(lo16) | bf16(hi16) << 16, round to nearest even
    const f32x2 v = {lo16, hi16};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
// (h | h << 16) and (l | l << 16) of v: the group-side slot pairs
RT_DEV void split_group_value(float v, uint32_t& hh, uint32_t& ll) {
    hh = bf16_pair_bits(v, v);
    const float rem = v - __uint_as_float(hh & 0xffff0000u);
    ll = bf16_pair_bits(rem, rem);
}
// (h | l << 16) of v: the ray-side slot pair (used twice)
RT_DEV uint32_t split_ray_value(float v) {
    const uint32_t hh = bf16_pair_bits(v, v);
    const float rem = v - __uint_as_float(hh & 0xffff0000u);
    return bf16_pair_bits(v, rem);
}

// Group operand image for the filter, built once per workgroup: ops[tile][8][64] dwords; lane l of tile t holds
// matrix row R = l&31: dwords 0-3 = the b chain's two values (-Cx,-Cy | -Cz,1 for l>>5 = 0 | 1) as (hh, ll) pairs,
// dwords 4-7 = the a*cc chain's (Cx,Cy | Cz,W) in its three-term form plus the constant slots (below).  Rows are PERMUTED so that the candidate bitmaps decode with two
// operations: output element e (0..15) of the lane in half h is matrix row (e&3) + 8(e>>2) + 4h, and that row holds
// group 32 t + 16 h + e.  The image always has an even number of tiles (bitmap words cover two tiles).
RT_DEV uint32_t mfma_tiles_for(uint32_t nTop) { return (((nTop + 31u) / 32u) + 1u) & ~1u; }
RT_DEV void build_mfma_operands(const float4* __restrict__ bounds, uint32_t nGroups, uint32_t nTiles, float* __restrict__ ops, uint32_t tid,
                                uint32_t nthreads) {
    uint32_t* img = reinterpret_cast<uint32_t*>(ops);
    for (uint32_t e = tid; e < nTiles * 64; e += nthreads) {
        const uint32_t t = e >> 6, l = e & 63, h = l >> 5, R = l & 31u;
        const uint32_t gi = t * 32 + 16u * ((R >> 2) & 1u) + (R & 3u) + 4u * (R >> 3);
        float4 B = make_float4(0.f, 0.f, 0.f, 1e30f);  // padding rows: a*cc~ = +huge => F < 0, never a candidate
        if (gi < nGroups) B = bounds[gi];  // bounds = the TOP level of the tree
        uint32_t* o = img + (size_t)t * kOpsPerTile + l;
        split_group_value(h == 0 ? -B.x : -B.z, o[0], o[64]);
        split_group_value(h == 0 ? -B.y : 1.f, o[128], o[192]);
        // a*cc chain: three cross terms per value (the lo*lo term, <= 2^-18 of the product, is left to the margin) and
        // two slots that add the per-ray constant a|o|^2 (1 - 2 K eps), carried by the ray side as hi + lo and
        // multiplied by 1 in the rows' first half only.  Slots: (y0h, y0h, y0l, y1h, y1h, y1l, one, one).
        uint32_t y0hh, y0ll, y1hh, y1ll;
        split_group_value(h == 0 ? B.x : B.z, y0hh, y0ll);
        split_group_value(h == 0 ? B.y : B.w, y1hh, y1ll);
        o[256] = y0hh;
        o[320] = (y0ll & 0xffffu) | (y1hh << 16);
        o[384] = (y1hh & 0xffffu) | (y1ll << 16);
        o[448] = h == 0 ? 0x3f803f80u : 0u;  // bf16(1.0) twice
    }
}

// Filter decision for this lane's 16 rows of one tile: the ray may hit the group unless F = b~^2 - t < 0 (t = a*cc~ - M)
// or the group is surely behind the origin.  "Behind" = the origin is outside the inflated bound (t > 0, which already
// includes the margin) and the centre is behind it by more than the rounding of b~ (b~ > bthr, bthr =
// 1e-4 sqrt(a) (|o| + max(|C|+R)) >= 185x the error bound of b~): then every point of the bound, hence of its member
// spheres, has t < 0 and the reference accepts no root (bias 0.001, ray-tracing.cpp:52).
// All three conditions are sign bits: one 3-input bit operation forms "rejected", one v_alignbit appends its sign bit
// to the lane's bitmap word (four VALU operations per (ray, group) pair, no branches, no LDS).
RT_DEV void mfma_post(const f32x16& Tb, const f32x16& Tg, float bthr, uint32_t& rejectedBits) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float t = Tg[e];  // a*cc~ - M: the per-ray constant is part of the contraction
        const float f = __builtin_fmaf(Tb[e], Tb[e], -t);
        const float u = bthr - Tb[e];  // negative <=> centre behind the origin
        const uint32_t rej = __float_as_uint(f) | (__float_as_uint(u) & ~__float_as_uint(t));  // sign bit = rejected
        rejectedBits = __builtin_amdgcn_alignbit(rejectedBits, rej, 31);  // (bits << 1) | (rej >> 31)
    }
}

// The filter formula on the VALU (any rounding; the bounds' margins cover it): sign bit set = the bound (C, W) cannot
// contain an acceptable root of the ray.  dO = d.o, m2a = -2a, cr = a|o|^2 (1 - 2 K eps), bt = the "behind" threshold.
RT_DEV int bound_rejected(const float4 B, V3 o, V3 d, float a, float dO, float m2a, float cr, float bt) {
    const float dC = __builtin_fmaf(d.z, B.z, __builtin_fmaf(d.y, B.y, d.x * B.x));
    const float oC = __builtin_fmaf(o.z, B.z, __builtin_fmaf(o.y, B.y, o.x * B.x));
    const float b = dO - dC;
    const float t = cr + __builtin_fmaf(m2a, oC, a * B.w);
    const float f = __builtin_fmaf(b, b, -t);
    const float u = bt - b;
    return __float_as_int(f) | (__float_as_int(u) & ~__float_as_int(t));
}

// lanes below mine that are set in mask
RT_DEV uint32_t prefix_count(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// Wave-wide inclusive prefix sum (DPP row shifts + row broadcasts, six VALU instructions, no LDS).
RT_DEV uint32_t wave_inclusive_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}
// value of `v` in lane `src` (per-lane source index; every lane of the wave must execute this)
RT_DEV float lane_fetch(uint32_t src, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src << 2), __float_as_int(v))); }

// One candidate of a ray's 128-bit bitmap (two 64-bit halves: the rows filtered by lanes l&31 and (l&31)+32).  Returns
// false when none is left.  Leading-zero order; bit N (from the top) of half h is group 16 h + N + (N & 48).
RT_DEV bool next_candidate(unsigned long long& cur, unsigned long long& nxt, uint32_t& hOff, uint32_t& gid) {
    if (cur == 0ull) {
        cur = nxt;
        nxt = 0ull;
        hOff = 16u;
    }
    if (cur == 0ull) return false;
    const uint32_t N = (uint32_t)__builtin_clzll(cur);
    cur &= ~(0x8000000000000000ull >> N);
    gid = hOff + N + (N & 48u);
    return true;
}

template <bool kTree>
RT_DEV void scan_list_mfma(const float4* __restrict__ tab, const float4* __restrict__ leaf, const uint32_t* __restrict__ orig,
                           const float* __restrict__ ops, uint32_t nTiles,
                           uint32_t nTop, const float4* __restrict__ tree, const uint32_t* levelOff, uint32_t nLevels, float boundNorm, V3 o,
                           V3 d, bool live, float& tmin, int& idx, uint16_t* waveCand, uint32_t lane, unsigned long long* dbg) {
    const float a = dot3(d, d);
    tmin = __builtin_inff();
    idx = -1;
    // per-ray operand values (filter arithmetic: any rounding, the margin covers it)
    const float dO = dot3(d, o);
    const float m2a = -2.f * a;
    const float gx = m2a * o.x, gy = m2a * o.y, gz = m2a * o.z;
    // a dead ray's a*cc~ is made huge so that nothing is ever recorded for it
    const float oo = dot3(o, o);
    const float cr = live ? (a * oo) * (1.f - 2.f * kMarginRel) : 1e30f;
    const float crLeaf = (a * oo) * (1.f - 2.f * kMarginKLeaf * 5.9604645e-8f);  // ... and of the one-sphere bounds
    const float bt = 1e-4f * __builtin_sqrtf(a) * (__builtin_sqrtf(oo) + boundNorm);
    // ray-side operands: values (k = 0,1 | 2,3) of the b chain [dx, dy | dz, d.o] and of the a*cc chain [gx, gy | gz, a].
    // Tile 0 (rays of lanes 0-31) takes k = 0,1 from the owner and k = 2,3 from lane+32; tile 1 the other way round:
    // v_permlane32_swap exchanges exactly those halves (upper half of the first register <-> lower half of the second).
    uint32_t b01x = split_ray_value(d.x), b01y = split_ray_value(d.y), b23x = split_ray_value(d.z), b23y = split_ray_value(dO);
    // a*cc chain, ray side: slots (x0h, x0l, x0h, x1h, x1l, x1h, ch, cl) = dwords (x0h|x0l, x0h|x1h, x1l|x1h, ch|cl)
    const uint32_t sgx = split_ray_value(gx), sgy = split_ray_value(gy), sgz = split_ray_value(gz), sga = split_ray_value(a);
    uint32_t g01a = sgx, g01b = __builtin_amdgcn_perm(sgy, sgx, 0x05040100u), g01c = __builtin_amdgcn_alignbit(sgy, sgy, 16);
    uint32_t g23a = sgz, g23b = __builtin_amdgcn_perm(sga, sgz, 0x05040100u), g23c = __builtin_amdgcn_alignbit(sga, sga, 16);
    uint32_t cr0 = __float_as_uint(cr), cr1 = cr0, bt0 = __float_as_uint(bt), bt1 = bt0;
#define RT_SWAP32(A, B)                                                        \
    {                                                                          \
        const auto r_ = __builtin_amdgcn_permlane32_swap((A), (B), false, false); \
        (A) = r_[0];                                                           \
        (B) = r_[1];                                                           \
    }
    RT_SWAP32(b01x, b23x)  // now b01* = tile 0's operand, b23* = tile 1's
    RT_SWAP32(b01y, b23y)
    RT_SWAP32(g01a, g23a)
    RT_SWAP32(g01b, g23b)
    RT_SWAP32(g01c, g23c)
    RT_SWAP32(cr0, cr1)  // per-ray scalars: cr0/bt0 belong to the ray of column lane&31 in tile 0, cr1/bt1 in tile 1
    RT_SWAP32(bt0, bt1)
#undef RT_SWAP32
    const bf16x8 Bb0 = __builtin_bit_cast(bf16x8, (u32x4){b01x, b01x, b01y, b01y});
    const bf16x8 Bb1 = __builtin_bit_cast(bf16x8, (u32x4){b23x, b23x, b23y, b23y});
    const bf16x8 Bg0 = __builtin_bit_cast(bf16x8, (u32x4){g01a, g01b, g01c, split_ray_value(__uint_as_float(cr0))});
    const bf16x8 Bg1 = __builtin_bit_cast(bf16x8, (u32x4){g23a, g23b, g23c, split_ray_value(__uint_as_float(cr1))});
    const float btT0 = __uint_as_float(bt0), btT1 = __uint_as_float(bt1);
    RT_STAMP(tf0);
    const uint32_t* opsImg = reinterpret_cast<const uint32_t*>(ops);
    // rejected-bits words: w0* = ray tile 0 (the ray of lane l&31), w1* = ray tile 1 (the ray of lane (l&31)+32);
    // *a = tiles 0,1, *b = tiles 2,3 (all ones = nothing to resolve when the image has only two tiles)
    uint32_t w0a = 0xffffffffu, w1a = 0xffffffffu, w0b = 0xffffffffu, w1b = 0xffffffffu;
    for (uint32_t sp = 0; sp < nTiles; sp += 2) {
        uint32_t r0 = 0u, r1 = 0u;
#pragma unroll
        for (uint32_t k = 0; k < 2; ++k) {
            const uint32_t* op = opsImg + (size_t)(sp + k) * kOpsPerTile + lane;
            const bf16x8 Ab = __builtin_bit_cast(bf16x8, (u32x4){op[0], op[64], op[128], op[192]});
            const bf16x8 Ag = __builtin_bit_cast(bf16x8, (u32x4){op[256], op[320], op[384], op[448]});
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            {   // ray tile 0 (rays 0..31)
                const f32x16 Tb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ab, Bb0, zero, 0, 0, 0);
                const f32x16 Tg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ag, Bg0, zero, 0, 0, 0);
                mfma_post(Tb, Tg, btT0, r0);
            }
            {   // ray tile 1 (rays 32..63)
                const f32x16 Tb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ab, Bb1, zero, 0, 0, 0);
                const f32x16 Tg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ag, Bg1, zero, 0, 0, 0);
                mfma_post(Tb, Tg, btT1, r1);
            }
        }
        if (sp == 0u) {
            w0a = r0;
            w1a = r1;
        } else {
            w0b = r0;
            w1b = r1;
        }
    }
    RT_STAMP(tf1);
    // every ray has two producer lanes (l&31 filtered rows of half 0, (l&31)+32 those of half 1); after the swaps
    // w0* holds the half-0 words and w1* the half-1 words of THIS lane's own ray
    {
        const auto ra = __builtin_amdgcn_permlane32_swap(w0a, w1a, false, false);
        const auto rb = __builtin_amdgcn_permlane32_swap(w0b, w1b, false, false);
        w0a = ra[0]; w1a = ra[1]; w0b = rb[0]; w1b = rb[1];
    }
    unsigned long long cur = ~(((unsigned long long)w0a << 32) | (unsigned long long)w0b);  // candidates of half 0
    unsigned long long nxt = ~(((unsigned long long)w1a << 32) | (unsigned long long)w1b);  // candidates of half 1
    uint32_t hOff = 0u;
#ifdef RT_STAMPS
    const uint32_t tot = (uint32_t)(__popcll(cur) + __popcll(nxt));
#endif
    if (!kTree) {
        // flat: the top level IS the groups.  The resolve is POOLED over the wave: a ray has 0..20 candidate groups (the
        // slowest lane of a wave ~9, the average ~3), so instead of every lane walking its own ray's candidates the
        // wave's (ray, group) pairs go to one work list in LDS and every lane takes the next pair, fetching that ray
        // from its owner lane (ds_bpermute).  Two pooled phases:
        //  A. sphere-level filter: the conservative formula on the four one-sphere bounds of the group (13 operations
        //     per sphere); surviving (ray, sphere) pairs are appended to a second list (ballot + mbcnt offsets);
        //  B. exact: Sphere::Intersect in the reference's operation order for one (ray, sphere) pair per lane, and the
        //     closest-hit merge as an LDS 64-bit minimum per ray over the key (t bits, original index, entry):
        //     smaller t wins, equal t keeps the lower ORIGINAL index, whatever the order of evaluation.
        uint16_t* poolA = waveCand;
        uint16_t* poolB = waveCand + kPoolA;
        unsigned long long* best = reinterpret_cast<unsigned long long*>(waveCand + kPoolA + kPoolB);
        best[lane] = ~0ull;
        uint32_t cntB = 0;
        const uint32_t nMine = (uint32_t)(__popcll(cur) + __popcll(nxt));
        bool pending = nMine != 0u;
        RT_STAMP(ta0);
        // phase B over the current contents of poolB (wave-uniform count)
        auto drainB = [&]() {
            for (uint32_t base = 0; base < cntB; base += kWaveSize) {
                const uint32_t k = base + lane;
                const bool has = k < cntB;
                const uint32_t ent = has ? (uint32_t)poolB[k] : 0u;
                const uint32_t r = ent >> 10, cand = ent & 1023u;
                const float rox = lane_fetch(r, o.x), roy = lane_fetch(r, o.y), roz = lane_fetch(r, o.z);
                const float rdx = lane_fetch(r, d.x), rdy = lane_fetch(r, d.y), rdz = lane_fetch(r, d.z);
                const float ra = lane_fetch(r, a);
#ifdef RT_STAMPS
                dbg[7] += 1;
#endif
                const float4 S = tab[cand];
                const float ocx = rox - S.x;
                const float ocy = roy - S.y;
                const float ocz = roz - S.z;
                const float b = (ocx * rdx + ocy * rdy) + ocz * rdz;
                const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - S.w;
                const float e = b * b - ra * cc;
                const float sq = __builtin_sqrtf(e);
                float t = (-b - sq) / ra;               // ray-tracing.cpp:56
                if (!(t > 0.001f)) t = (-b + sq) / ra;  // :69
                // `e > 0` is the reference's own test (ray-tracing.cpp:54); `t < inf` is the scan's initial tmin
                if (has && e > 0.f && t > 0.001f && t < __builtin_inff()) {
                    const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)((orig[cand] << 16) | cand);
                    __hip_atomic_fetch_min(best + r, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            cntB = 0;
        };
        while (__ballot(pending) != 0ull) {
            // lanes whose items fit into the list this pass: a prefix of the pending lanes
            const uint32_t incl = wave_inclusive_sum(pending ? nMine : 0u);
            const bool take = pending && incl <= kPoolA;
            const uint64_t takeMask = __ballot(take);
            const uint32_t lastLane = 63u - (uint32_t)__builtin_clzll(takeMask);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)lastLane);
            if (take) {
                // one loop per bitmap half (leading-zero order; bit N of half h is group 16 h + N + (N & 48))
                uint16_t* wp = poolA + (incl - nMine);
                const uint32_t tag = lane << 7;
                while (cur != 0ull) {
                    const uint32_t N = (uint32_t)__builtin_clzll(cur);
                    cur &= ~(0x8000000000000000ull >> N);
                    *wp++ = (uint16_t)(tag | (N + (N & 48u)));
                }
                while (nxt != 0ull) {
                    const uint32_t N = (uint32_t)__builtin_clzll(nxt);
                    nxt &= ~(0x8000000000000000ull >> N);
                    *wp++ = (uint16_t)(tag | (16u + N + (N & 48u)));
                }
                pending = false;
            }
            for (uint32_t base = 0; base < total; base += kWaveSize) {
#ifdef RT_STAMPS
                dbg[6] += 1;
#endif
                const uint32_t k = base + lane;
                const bool has = k < total;
                const uint32_t item = has ? (uint32_t)poolA[k] : 0u;
                const uint32_t r = item >> 7, gid = item & 127u;
                const V3 fo = v3(lane_fetch(r, o.x), lane_fetch(r, o.y), lane_fetch(r, o.z));
                const V3 fd = v3(lane_fetch(r, d.x), lane_fetch(r, d.y), lane_fetch(r, d.z));
                const float fa = lane_fetch(r, a), fdO = lane_fetch(r, dO), fcr = lane_fetch(r, crLeaf), fbt = lane_fetch(r, bt);
                const float4* lb = leaf + 4u * gid;
                uint32_t rb = 0u;
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q)
                    rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected(lb[q], fo, fd, fa, fdO, -2.f * fa, fcr, fbt), 31);
                const uint32_t m = has ? (~rb & 15u) : 0u;  // bit 3-q = sphere q of the group
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) {
                    const bool hit = ((m >> (3u - q)) & 1u) != 0u;
                    const uint64_t hm = __ballot(hit);
                    if (hit) poolB[cntB + prefix_count(hm)] = (uint16_t)(r << 10 | (4u * gid + q));
                    cntB += (uint32_t)__popcll(hm);
                }
                if (cntB > kPoolB - 4u * kWaveSize) drainB();
            }
        }
        RT_STAMP(ta1);
        RT_ACC(dbg[4], ta0, ta1);
        drainB();
        const unsigned long long mineKey = best[lane];
        const uint32_t tb = (uint32_t)(mineKey >> 32);
        if (tb < 0x7f800000u) {
            tmin = __uint_as_float(tb);
            idx = (int)(mineKey & 0xffffull);
        }
        RT_STAMP(ta2);
        RT_ACC(dbg[5], ta1, ta2);
    } else {
        // descent, POOLED over the wave: a LIFO work list of (ray, node) pairs in LDS.  Every round the 64 lanes pop up
        // to 64 pairs, fetch the pair's ray from its owner lane, test the node's four children with the conservative
        // formula -- bounds of the level below for an internal node, one-sphere bounds for a group -- and push the
        // surviving children (ballot + mbcnt offsets): internal survivors back on the work list, sphere survivors on the
        // exact list, which is drained by the same pooled exact phase as the flat scan (ds_min_u64 merge per ray).  The
        // walk order is irrelevant to the result.  The list cannot overflow: a round pops at most (room - reserve) / 3
        // pairs, and the reserve lets a single pair always be expanded down to the leaves (3 slots per level).
        uint32_t* work = reinterpret_cast<uint32_t*>(waveCand);               // kTreeWork entries: ray << 20 | level << 16 | index
        uint32_t* exact = work + kTreeWork;                                   // kTreeExact entries: ray << 16 | scan entry
        unsigned long long* best = reinterpret_cast<unsigned long long*>(exact + kTreeExact);
        best[lane] = ~0ull;
        const uint32_t topLevel = nLevels - 1u;
        const float aoo = a * oo;
        uint32_t nWork = 0, nExact = 0;
        auto drainExact = [&]() {
            for (uint32_t base = 0; base < nExact; base += kWaveSize) {
                const uint32_t k = base + lane;
                const bool has = k < nExact;
                const uint32_t ent = has ? exact[k] : 0u;
                const uint32_t r = ent >> 16, cand = ent & 0xffffu;
                const float rox = lane_fetch(r, o.x), roy = lane_fetch(r, o.y), roz = lane_fetch(r, o.z);
                const float rdx = lane_fetch(r, d.x), rdy = lane_fetch(r, d.y), rdz = lane_fetch(r, d.z);
                const float ra = lane_fetch(r, a);
                const float4 S = tab[cand];
                const float ocx = rox - S.x;
                const float ocy = roy - S.y;
                const float ocz = roz - S.z;
                const float b = (ocx * rdx + ocy * rdy) + ocz * rdz;
                const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - S.w;
                const float e = b * b - ra * cc;
                const float sq = __builtin_sqrtf(e);
                float t = (-b - sq) / ra;               // ray-tracing.cpp:56
                if (!(t > 0.001f)) t = (-b + sq) / ra;  // :69
                if (has && e > 0.f && t > 0.001f && t < __builtin_inff()) {
                    const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)((orig[cand] << 16) | cand);
                    __hip_atomic_fetch_min(best + r, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            nExact = 0;
        };
        for (;;) {
            // feed: when fewer than a round's worth of pairs is listed, every lane with top-level candidates left adds one
            if (nWork < (uint32_t)kWaveSize) {
                uint32_t t = 0;
                const bool add = next_candidate(cur, nxt, hOff, t);
                const uint64_t am = __ballot(add);
                if (add) work[nWork + prefix_count(am)] = lane << 20 | topLevel << 16 | t;
                nWork += (uint32_t)__popcll(am);
            }
            if (nWork == 0u) break;
            const uint32_t room = nWork < kTreeWork - kTreeReserve ? kTreeWork - kTreeReserve - nWork : 0u;
            uint32_t np = room / 3u;
            np = np < 1u ? 1u : np;
            np = np > (uint32_t)kWaveSize ? (uint32_t)kWaveSize : np;
            np = np > nWork ? nWork : np;
            const bool has = lane < np;
            const uint32_t ent = has ? work[nWork - 1u - lane] : 0u;
            nWork -= np;
            const uint32_t r = ent >> 20, lvl = (ent >> 16) & 7u, j = ent & 0xffffu;
            const V3 fo = v3(lane_fetch(r, o.x), lane_fetch(r, o.y), lane_fetch(r, o.z));
            const V3 fd = v3(lane_fetch(r, d.x), lane_fetch(r, d.y), lane_fetch(r, d.z));
            const float fa = lane_fetch(r, a), fdO = lane_fetch(r, dO), faoo = lane_fetch(r, aoo), fbt = lane_fetch(r, bt);
            const bool internal = lvl > 0u;
            float4 B0, B1, B2, B3;
            if (internal) {
                const uint32_t cl = lvl - 1u;
                uint32_t off = levelOff[0];
#pragma unroll
                for (uint32_t k = 1; k < kMaxLevels - 1; ++k) off = cl == k ? levelOff[k] : off;
                const float4* ch = tree + off + 4u * j;
                B0 = ch[0]; B1 = ch[1]; B2 = ch[2]; B3 = ch[3];
            } else {
                const float4* lb = leaf + 4u * j;
                B0 = lb[0]; B1 = lb[1]; B2 = lb[2]; B3 = lb[3];
            }
            const float fcr = faoo * (internal ? (1.f - 2.f * kMarginKValu * 5.9604645e-8f) : (1.f - 2.f * kMarginKLeaf * 5.9604645e-8f));
            const float fm2a = -2.f * fa;
            uint32_t rb = 0u;
            rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected(B0, fo, fd, fa, fdO, fm2a, fcr, fbt), 31);
            rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected(B1, fo, fd, fa, fdO, fm2a, fcr, fbt), 31);
            rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected(B2, fo, fd, fa, fdO, fm2a, fcr, fbt), 31);
            rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected(B3, fo, fd, fa, fdO, fm2a, fcr, fbt), 31);
            const uint32_t m = has ? (~rb & 15u) : 0u;  // bit 3-q = child 4j + q
            if (nExact > kTreeExact - 4u * kWaveSize) drainExact();
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const bool hit = ((m >> (3u - q)) & 1u) != 0u;
                const uint64_t wm = __ballot(hit && internal);
                const uint64_t em = __ballot(hit && !internal);
                if (hit && internal) work[nWork + prefix_count(wm)] = r << 20 | (lvl - 1u) << 16 | (4u * j + q);
                if (hit && !internal) exact[nExact + prefix_count(em)] = r << 16 | (4u * j + q);
                nWork += (uint32_t)__popcll(wm);
                nExact += (uint32_t)__popcll(em);
            }
        }
        drainExact();
        const unsigned long long mineKey = best[lane];
        const uint32_t tb = (uint32_t)(mineKey >> 32);
        if (tb < 0x7f800000u) {
            tmin = __uint_as_float(tb);
            idx = (int)(mineKey & 0xffffull);
        }
    }
#ifdef RT_STAMPS
    {
        RT_STAMP(tf2);
        dbg[0] += tf1 - tf0;
        dbg[1] += tf2 - tf1;
        dbg[2] += tot;
        uint32_t mx = tot;
        for (int off = 32; off > 0; off >>= 1) { const uint32_t o2 = __shfl_xor(mx, off); mx = o2 > mx ? o2 : mx; }
        dbg[3] += mx;
    }
#endif
}

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
RT_DEV bool scatter_only(const Mat& m, V3 rd, V3 nrm, Draws& draws, V3& atten, V3& outDir, V3& tex) {
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
            const float nDotV = sat1(ndv);
            const float refl = 0.04f + (1.f - 0.04f) * rt_powf(1.f - nDotV, 5.f);
            const float u = draws.next();
            if (refl > u) {
                atten = v3(1.f, 1.f, 1.f);
                raw = mirror;
            } else {
                atten = tex;
                const float u1 = draws.next();  // HaltonSampleHemisphere's two dimensions
                const float u2 = draws.next();
                const float r = __builtin_sqrtf(1.f - u1 * u1);  // quasi-random.cpp:41
                const float phi = (2.f * 3.141592654f) * u2;
                double sn, cs;
                sincos_f64(phi, sn, cs);
                const float hx = r * (float)cs, hy = r * (float)sn, hz = u1;
                const V3 b3 = nrm;
                const V3 up = __builtin_fabsf(nrm.x) < 0.5f ? v3(1.f, 0.f, 0.f) : v3(0.f, 1.f, 0.f);
                const V3 b1 = cross3(up, b3);
                const V3 b2 = cross3(b3, b1);
                raw = (hx * b1 + hy * b2) + hz * b3;
            }
            scattered = true;
        }
    }
    outDir = normalize3(raw);
    if (!scattered) outDir = v3(0.f, 0.f, 0.f);
    return scattered;
}

// Emit + DirectionalLight::Shade's unoccluded value (light.cpp:21-40) and Emit + 0 (its value when occluded).
template <class P>
RT_DEV void shade_value(const P& p, const Mat& m, V3 tex, V3 pos, V3 nrm, bool wantShade, V3& local, V3& localOccluded) {
    V3 emit = v3(0.f, 0.f, 0.f);
    if (m.type == RT_MAT_EMISSIVE) emit = m.luminance * tex;  // material.cpp:172-175; 0 for every other material
    localOccluded = emit + v3(0.f, 0.f, 0.f);  // Shade returns XM_Zero when the sun is occluded (light.cpp:15-18)
    local = localOccluded;
    if (!wantShade) return;
    // Material getters (material.h:26-29,42-45,59-62,76-79)
    V3 albedo = v3(0.f, 0.f, 0.f), f0 = v3(0.04f, 0.04f, 0.04f);
    if (m.type == RT_MAT_DIELECTRIC_OPAQUE) albedo = tex;
    else if (m.type == RT_MAT_METAL) f0 = tex;
    else if (m.type == RT_MAT_EMISSIVE) f0 = v3(0.f, 0.f, 0.f);
    const float smooth = (m.type == RT_MAT_EMISSIVE) ? 0.f : m.smoothness;

    // DirectionalLight::Shade, light.cpp:21-40 (viewOrigin is always the camera origin, spheres-app.cpp:250)
    const V3 L = v3(p.sun_dir[0], p.sun_dir[1], p.sun_dir[2]);
    const float nDotL = sat1(dot3(nrm, L));
    const V3 radianceIn = v3(p.sun_rad[0] * nDotL, p.sun_rad[1] * nDotL, p.sun_rad[2] * nDotL);
    const V3 viewDir = normalize3(v3(p.cam_o[0], p.cam_o[1], p.cam_o[2]) - pos);
    const V3 halfVector = normalize3(L + viewDir);
    const float nDotH = sat1(dot3(nrm, halfVector));
    const float nDotV2 = sat1(dot3(viewDir, nrm));
    const float p5 = rt_powf(1.f - nDotV2, 5.f);
    const float ps = rt_powf(nDotH, smooth);
    const V3 one = v3(1.f, 1.f, 1.f);
    const V3 reflectance = f0 + (one - f0) * p5;
    const V3 spec = ((reflectance * 0.125f) * (smooth + 8.f)) * ps;
    const V3 shade = radianceIn * (albedo + spec);
    local = emit + shade;
}

// Scatter, then Emit + Shade with the sun assumed visible (the scan-based shadow path decides later).
template <class P, class Draws>
RT_DEV bool scatter_and_shade(const P& p, const Mat& m, V3 rd, V3 pos, V3 nrm, Draws& draws, V3& atten, V3& outDir,
                              V3& local, V3& localOccluded) {
    V3 tex;
    const bool scattered = scatter_only(m, rd, nrm, draws, atten, outDir, tex);
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
RT_DEV bool sphere_any_hit(const float4 S, V3 o, V3 d, float a) {
    const float ocx = o.x - S.x;
    const float ocy = o.y - S.y;
    const float ocz = o.z - S.z;
    const float b = (ocx * d.x + ocy * d.y) + ocz * d.z;
    const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - S.w;
    const float disc = b * b - a * cc;
    if (disc > 0.f) {  // ray-tracing.cpp:54-71
        const float sq = __builtin_sqrtf(disc);
        if ((-b - sq) / a > 0.001f) return true;
        if ((-b + sq) / a > 0.001f) return true;
    }
    return false;
}

// Two phases, like the closest-hit resolve: first the discriminants of every listed sphere (cheap, uniform), keeping
// up to four spheres whose roots are possible in a register queue; then roots (sqrt + divides) only for those, until
// one occludes.  root_possible() is exact, so the answer is the reference's any-hit over the same spheres.
template <class P>
RT_DEV bool shadow_query(const P& p, const float4* __restrict__ tab, const uint16_t* __restrict__ cellStart,
                         const uint16_t* __restrict__ entries, const uint16_t* __restrict__ glob, V3 pos, V3 L, float aL) {
    bool occluded = false;
    unsigned long long queue = 0ull;
    uint32_t nq = 0;
#define RT_CONSIDER(ID)                                                              \
    {                                                                                \
        const uint32_t id_ = (ID);                                                   \
        const float4 S = tab[id_];                                                   \
        const float ocx = pos.x - S.x;                                               \
        const float ocy = pos.y - S.y;                                               \
        const float ocz = pos.z - S.z;                                               \
        const float b = (ocx * L.x + ocy * L.y) + ocz * L.z;                         \
        const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - S.w;                \
        const float disc = b * b - aL * cc;                                          \
        if (root_possible(disc, b)) {                                                \
            if (nq < 4u) {                                                           \
                queue = (queue << 16) | (unsigned long long)id_;                     \
                ++nq;                                                                \
            } else {                                                                 \
                occluded = occluded || sphere_any_hit(S, pos, L, aL); /* queue full (rare): evaluate now */ \
            }                                                                        \
        }                                                                            \
    }
    for (uint32_t k = 0; k < p.sg_nglobal; ++k) RT_CONSIDER(glob[k])
    const float u = dot3(pos, v3(p.sg_e1[0], p.sg_e1[1], p.sg_e1[2]));
    const float v = dot3(pos, v3(p.sg_e2[0], p.sg_e2[1], p.sg_e2[2]));
    const float fx = (u - p.sg_u0) * p.sg_inv_cell, fy = (v - p.sg_v0) * p.sg_inv_cell;
    if (fx >= 0.f && fy >= 0.f && fx < (float)p.sg_nx && fy < (float)p.sg_ny) {
        const uint32_t c = (uint32_t)fy * p.sg_nx + (uint32_t)fx;
        const uint32_t e1 = cellStart[c + 1];
        for (uint32_t e = cellStart[c]; e < e1; ++e) RT_CONSIDER(entries[e])
    }
#undef RT_CONSIDER
    while (nq > 0u && !occluded) {
        const uint32_t id = (uint32_t)(queue & 0xffffull);
        queue >>= 16;
        --nq;
        occluded = sphere_any_hit(tab[id], pos, L, aL);
    }
    return occluded;
}


enum : uint32_t { kIdle = 0u, kNeedClosest = 1u, kNeedShadow = 2u };

// Work-item index -> (column i, global row j, sample s) and the path's slot in the sample buffer: the explicit path
// list of the unit tests, or the tiled order below over the rows of this shard.  slot = index either way.
RT_DEV void path_coordinates(const TraceParams& p, uint32_t q, uint32_t& i, uint32_t& j, uint32_t& s, uint32_t& slot) {
    if (p.path_list) {
        i = p.path_list[3 * q];
        j = p.path_list[3 * q + 1];
        s = p.path_list[3 * q + 2];
        slot = q;
    } else {
        // work order = storage order = [tile of 64 local pixels][sample of the pass][pixel in tile] (the last tile is as
        // wide as the pixels left): a wave's block of 256 consecutive work-items is four sample planes of one tile, so
        // its 12-byte results fill whole cache lines, and rt_accumulate_kernel reads 768 contiguous bytes per plane
        const uint32_t tileSpan = 64u * p.spp_pass;
        const uint32_t nFull = p.npix_local >> 6;
        const uint32_t tile = q / tileSpan;
        uint32_t pl, k;
        if (tile < nFull) {
            const uint32_t rem = q - tile * tileSpan;
            k = rem >> 6;
            pl = (tile << 6) + (rem & 63u);
        } else {
            const uint32_t rem = q - nFull * tileSpan;
            const uint32_t wl = p.npix_local - (nFull << 6);
            k = rem / wl;
            pl = (nFull << 6) + (rem - k * wl);
        }
        s = p.s0 + k;
        const uint32_t lr = pl / p.W;
        i = pl - lr * p.W;
        j = rowset_global_row(p.rs, lr);
        slot = q;
    }
}

// ============================================================================ megakernel
// Persistent threads: every wave loops { refill idle lanes from the queue; one list scan for all
// lanes; per-lane state transition } until the queue is empty and all its lanes are idle.  Waves
// never synchronise with each other after the LDS staging barrier, and every wave's loop ends when
// the (bounded, monotonically consumed) queue is exhausted and its at most 64 paths of at most
// max_depth+1 segments have finished.
// kScan: 0 = VALU sign filter per group (any scene size), 1 = matrix-core filter over the groups (tables in LDS),
// 2 = matrix-core filter over the top level of the bounds hierarchy + per-lane descent (tables through L2).
template <bool kLds, int kThreads, int kScan, bool kCache>
__global__ void __launch_bounds__(kThreads, 1) rt_trace_kernel(const TraceParams p) {
    constexpr bool kMfma = kScan != 0;
    extern __shared__ float4 smem[];
    const float4* scanTab = p.scan;
    const uint32_t* origTab = p.orig;
    const float4* leafTab = p.leaf;
    const float* radTab = p.radius;
    const rt_material* matTab = p.mats;
    // per-wave candidate regions first (kWaveCandBytes each; the VALU scan uses the first 2 KiB of its region)
    // [0, kConstBytes): the scene constants; then the per-wave regions; then the tables
    SceneConsts* ldsK = reinterpret_cast<SceneConsts*>(smem);
    if (threadIdx.x == 0) fill_consts(p, *ldsK);
    uint16_t* candBase = reinterpret_cast<uint16_t*>(smem + kConstBytes / 16);
    constexpr uint32_t kWaveRegion = kScan == 2 ? kWaveCandBytes : kWaveListBytes;  // the descent stack only exists in tree mode
    float4* tabBase = smem + kConstBytes / 16 + (kThreads / kWaveSize) * (kWaveRegion / 16);
    const float* mfmaOps = nullptr;
    const float4* treeTab = p.tree;
    const uint16_t* sgCell = p.sg_cell_start;
    const uint16_t* sgEntries = p.sg_entries;
    const uint16_t* sgGlobal = p.sg_global;
    const uint32_t topLevel = p.n_levels - 1u;
    const uint32_t nTop = p.level_cnt[topLevel];
    const uint32_t nTiles = mfma_tiles_for(nTop);  // even; nTop <= 128 => at most four
    if (kLds) {
        // LDS image (16-byte aligned pieces): scan | one-sphere bounds (matrix-core scan only) | orig | materials (48 B =
        // 3 float4) | radii | filter operands | shadow index
        float4* ldsScan = tabBase;
        float4* ldsLeaf = ldsScan + p.n_padded;
        uint32_t* ldsOrig = reinterpret_cast<uint32_t*>(ldsLeaf + (kScan == 1 ? p.n_padded : 0u));
        float4* ldsMat = reinterpret_cast<float4*>(ldsOrig + p.n_padded);  // n_padded is a multiple of 4
        const uint32_t nMatLds = p.mats_in_lds ? p.n_padded : 0u;
        float* ldsRad = reinterpret_cast<float*>(ldsMat + (size_t)nMatLds * 3);
        const float4* gMat = reinterpret_cast<const float4*>(p.mats);
        for (uint32_t k = threadIdx.x; k < p.n_padded; k += blockDim.x) {
            ldsScan[k] = p.scan[k];
            ldsOrig[k] = p.orig[k];
            if (kScan == 1) ldsLeaf[k] = p.leaf[k];
        }
        for (uint32_t k = threadIdx.x; k < nMatLds * 3; k += blockDim.x) ldsMat[k] = gMat[k];
        for (uint32_t k = threadIdx.x; k < p.n_padded; k += blockDim.x) ldsRad[k] = p.radius[k];
        if (kMfma) {
            float* ldsOps = ldsRad + p.n_padded;  // a multiple of 4
            build_mfma_operands(p.tree + p.level_off[topLevel], nTop, nTiles, ldsOps, threadIdx.x, blockDim.x);
            mfmaOps = ldsOps;
            if (p.sg_enabled && p.sg_in_lds) {  // shadow index after the operand image
                uint16_t* g = reinterpret_cast<uint16_t*>(ldsOps + (size_t)nTiles * kOpsPerTile);
                const uint32_t nc = p.sg_nx * p.sg_ny + 1u;
                for (uint32_t k = threadIdx.x; k < nc; k += blockDim.x) g[k] = p.sg_cell_start[k];
                for (uint32_t k = threadIdx.x; k < p.sg_nentries; k += blockDim.x) g[nc + k] = p.sg_entries[k];
                for (uint32_t k = threadIdx.x; k < p.sg_nglobal; k += blockDim.x) g[nc + p.sg_nentries + k] = p.sg_global[k];
                sgCell = g;
                sgEntries = g + nc;
                sgGlobal = g + nc + p.sg_nentries;
            }
        }
        __syncthreads();
        scanTab = ldsScan;
        origTab = ldsOrig;
        if (kScan == 1) leafTab = ldsLeaf;
        radTab = ldsRad;
        if (p.mats_in_lds) matTab = reinterpret_cast<const rt_material*>(ldsMat);
    } else if (kMfma) {
        // exact tables stay in global memory (L2); the top level's operand image and, when they fit, all bounds live in LDS
        float* ldsOps = reinterpret_cast<float*>(tabBase);
        build_mfma_operands(p.tree + p.level_off[topLevel], nTop, nTiles, ldsOps, threadIdx.x, blockDim.x);
        mfmaOps = ldsOps;
        if (p.tree_in_lds) {
            float4* ldsTree = reinterpret_cast<float4*>(ldsOps + (size_t)nTiles * kOpsPerTile);
            const uint32_t nNodes = p.level_off[topLevel] + nTop;
            for (uint32_t k = threadIdx.x; k < nNodes; k += blockDim.x) ldsTree[k] = p.tree[k];
            treeTab = ldsTree;
        }
        __syncthreads();
    }

    const uint32_t lane = threadIdx.x & (kWaveSize - 1);
    uint16_t* waveCand = candBase + (threadIdx.x / kWaveSize) * (kWaveRegion / 2);
    uint16_t* cand = waveCand + lane;
    __syncthreads();  // the constants block (and, above, the staged tables) are visible to every wave from here on
    const SceneConsts& K = *ldsK;
    const V3 sunDir = v3(K.sun_dir[0], K.sun_dir[1], K.sun_dir[2]);
    const float aSun = dot3(sunDir, sunDir);  // the `a` of every shadow ray (ray-tracing.cpp:46)

    // per-lane path state
    V3 ro = v3(0.f, 0.f, 0.f), rd = v3(0.f, 0.f, 1.f);  // current ray
    V3 thr = v3(1.f, 1.f, 1.f), rad = v3(0.f, 0.f, 0.f);
    V3 pend = v3(0.f, 0.f, 0.f), nextDir = v3(0.f, 0.f, 0.f);
    StreamDraws draws{Rng{1u, 0u, 0u, 0u}};
    uint32_t q = 0, depth = 0, state = kIdle, pathTrav = 0;  // q = the path's slot in the sample buffer
    bool contAfterShadow = false, pathScattered = false;
    uint32_t nTrav = 0, nSeg = 0;

    // wave-uniform queue window and prepared-path cache (48-byte slots: origin, direction, stream state, path index)
    uint32_t blkNext = 0, blkEnd = 0, cachePos = 0, cacheCnt = 0;
    bool queueEmpty = false;
    float4* rayCache = kCache ? smem + p.ray_cache_off16 + (threadIdx.x / kWaveSize) * (kRayCacheBytes / 16) : nullptr;

    unsigned long long dbgScan[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)dbgScan;
#ifdef RT_STAMPS
    unsigned long long cyRefill = 0, cyScan = 0, cyTrans = 0, cyIters = 0, cyHit[6] = {0, 0, 0, 0, 0, 0};
#endif
    for (;;) {
        RT_STAMP(ts0);
        // ------------------------------------------------ refill idle lanes (ballot + prefix)
        // New paths come from a per-wave cache of 64 prepared paths in LDS: when it runs empty ALL 64 lanes generate
        // the next 64 paths of the wave's queue block at once (index arithmetic, stream seeding, Camera::GetRay with
        // its two normalisations: ~350 instructions at full lane utilisation instead of once per handful of idle
        // lanes), and an idle lane just pops a 48-byte slot.  In-flight paths are untouched: the generation only uses
        // temporaries.  Without room for the cache in LDS (kCache == false) idle lanes generate their own path.
        uint64_t idleMask = __ballot(state == kIdle);
        if (kCache) {
            while (idleMask != 0ull) {
                if (cachePos == cacheCnt) {
                    if (queueEmpty) break;
                    if (blkNext == blkEnd) {
                        uint32_t b = 0;
                        if (lane == 0) b = atomicAdd(p.queue_head, kQueueBlock);
                        b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                        blkNext = b < p.total_paths ? b : p.total_paths;
                        blkEnd = (b + kQueueBlock) < p.total_paths ? (b + kQueueBlock) : p.total_paths;
                        if (b >= p.total_paths) {
                            queueEmpty = true;
                            break;
                        }
                    }
                    const uint32_t nGen = (blkEnd - blkNext) < (uint32_t)kWaveSize ? (blkEnd - blkNext) : (uint32_t)kWaveSize;
                    if (lane < nGen) {
                        const uint32_t qn = blkNext + lane;
                        uint32_t i, j, s, slotn;
                        path_coordinates(p, qn, i, j, s, slotn);
                        const Rng g = rng_seed(p.seed, j * p.W + i, s);
                        V3 go, gd;
                        gen_primary_ray(K, i, j, s, go, gd);
                        float4* slot = rayCache + 3u * lane;
                        slot[0] = make_float4(go.x, go.y, go.z, gd.x);
                        slot[1] = make_float4(gd.y, gd.z, __uint_as_float(g.s0), __uint_as_float(g.s1));
                        slot[2] = make_float4(__uint_as_float(g.s2), __uint_as_float(g.s3), __uint_as_float(slotn), 0.f);
                    }
                    blkNext += nGen;
                    cacheCnt = nGen;
                    cachePos = 0;
                }
                const uint32_t avail = cacheCnt - cachePos;
                const uint32_t want = (uint32_t)__popcll(idleMask);
                const uint32_t rank = prefix_count(idleMask);
                if (state == kIdle && rank < avail) {
                    const float4* slot = rayCache + 3u * (cachePos + rank);
                    const float4 A = slot[0], B = slot[1], C = slot[2];
                    ro = v3(A.x, A.y, A.z);
                    rd = v3(A.w, B.x, B.y);
                    draws.rng = Rng{__float_as_uint(B.z), __float_as_uint(B.w), __float_as_uint(C.x), __float_as_uint(C.y)};
                    q = __float_as_uint(C.z);
                    thr = v3(1.f, 1.f, 1.f);
                    rad = v3(0.f, 0.f, 0.f);
                    depth = 0;
                    pathTrav = 0;
                    state = kNeedClosest;
                }
                cachePos += want < avail ? want : avail;
                idleMask = __ballot(state == kIdle);
            }
        } else {
            while (idleMask != 0ull && !queueEmpty) {
                if (blkNext == blkEnd) {
                    uint32_t b = 0;
                    if (lane == 0) b = atomicAdd(p.queue_head, kQueueBlock);
                    b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                    blkNext = b < p.total_paths ? b : p.total_paths;
                    blkEnd = (b + kQueueBlock) < p.total_paths ? (b + kQueueBlock) : p.total_paths;
                    if (b >= p.total_paths) {
                        queueEmpty = true;
                        break;
                    }
                }
                const uint32_t avail = blkEnd - blkNext;
                const uint32_t want = (uint32_t)__popcll(idleMask);
                const uint32_t rank = prefix_count(idleMask);
                if (state == kIdle && rank < avail) {
                    uint32_t i, j, s;
                    path_coordinates(p, blkNext + rank, i, j, s, q);
                    draws.rng = rng_seed(p.seed, j * p.W + i, s);
                    gen_primary_ray(K, i, j, s, ro, rd);
                    thr = v3(1.f, 1.f, 1.f);
                    rad = v3(0.f, 0.f, 0.f);
                    depth = 0;
                    pathTrav = 0;
                    state = kNeedClosest;
                }
                blkNext += want < avail ? want : avail;
                idleMask = __ballot(state == kIdle);
            }
        }
        if (__ballot(state != kIdle) == 0ull) break;  // queue empty and every lane drained

        RT_STAMP(ts1);
        // ------------------------------------------------ one list scan for every live lane
        float tmin = 0.f;
        int idx = -1;
        if (kMfma) {
            // every lane takes part: lane l also supplies operands for, and filters half the spheres of,
            // the ray owned by lane l^32, whether or not its own ray is live
            scan_list_mfma<kScan == 2>(scanTab, leafTab, origTab, mfmaOps, nTiles, nTop, treeTab, p.level_off, p.n_levels, p.bound_norm, ro, rd,
                                       state != kIdle, tmin, idx, waveCand, lane, dbgScan);
        } else if (state != kIdle) {
            scan_list_deferred(scanTab, origTab, p.n_padded, ro, rd, tmin, idx, cand);
        }
        if (state != kIdle) {
            ++nTrav;
            ++pathTrav;
        }

        RT_STAMP(ts2);
        // ------------------------------------------------ state transitions
        bool finished = false;
#ifdef RT_STAMPS
        unsigned long long thA = 0, thB = 0;
#endif
        if (state == kNeedClosest) {
            ++nSeg;
            if (idx < 0) {
                // miss: sky Emissive::Emit (spheres-app.cpp:255)
                const V3 sky = v3(K.sky_emit[0], K.sky_emit[1], K.sky_emit[2]);
                rad = rad + thr * sky;
                finished = true;
            } else {
                const float4 S = scanTab[idx];
                const float radius = radTab[idx];  // radius and material tables are in scan-entry (clustered) order
                const Mat m = load_material(matTab, idx);
                const V3 center = v3(S.x, S.y, S.z);
                const V3 pos = tmin * rd + ro;          // XMVectorMultiplyAdd(t, dir, origin), ray-tracing.cpp:57
                const V3 nrm = (pos - center) / radius;  // ray-tracing.cpp:58
                V3 atten, local, localOcc, tex;
                RT_STAMP(th0);
                const bool scattered = scatter_only(m, rd, nrm, draws, atten, nextDir, tex);  // Scatter first: it draws (spheres-app.cpp:246)
                RT_STAMP(th1);
                const bool cont = (depth < p.max_depth) && scattered;  // spheres-app.cpp:247
                const bool useIndex = K.sg_enabled && dot3(pos, pos) <= K.sg_p0sq;
                bool occluded = false;
                if (useIndex) occluded = shadow_query(K, scanTab, sgCell, sgEntries, sgGlobal, pos, sunDir, aSun);
                RT_STAMP(th2);
                // the Blinn-Phong value (two normalisations, two pows) is only needed when the sun is visible or unknown
                shade_value(K, m, tex, pos, nrm, !occluded, local, localOcc);
                RT_STAMP(th3);
#ifdef RT_STAMPS
                thA = th0;
                thB = th3;
#endif
                RT_ACC(cyHit[0], th0, th1);
                RT_ACC(cyHit[1], th1, th2);
                RT_ACC(cyHit[2], th2, th3);
                if (useIndex) {
                    // shadow ray answered by the exact footprint index: no second scan for this hit
                    ++nTrav;  // the shadow ray still counts as a traversal of the scene (matches the oracle's counter)
                    ++pathTrav;
                    if (!occluded) rad = rad + thr * local;           // radiance += throughput * (Emit + Shade)
                    else if (!scattered) rad = rad + thr * localOcc;  // occluded: throughput * (Emit + 0)
                    if (cont) {
                        thr = thr * atten;
                        ro = pos;
                        rd = nextDir;
                        ++depth;
                    } else {
                        finished = true;
                    }
                } else {
                    pend = thr * local;
                    // a path that does not scatter ends here, so its nextDir registers carry throughput * (Emit + 0),
                    // the value the reference adds when the sun is occluded (0 for every non-emissive material)
                    pathScattered = scattered;
                    if (!scattered) nextDir = thr * localOcc;
                    contAfterShadow = cont;
                    thr = thr * atten;
                    ro = pos;  // shadow ray and scattered ray both start at hit.pos
                    rd = sunDir;
                    state = kNeedShadow;
                }
            }
        } else if (state == kNeedShadow) {
            if (idx < 0) rad = rad + pend;              // sun visible: radiance += throughput * (Emit + Shade)
            else if (!pathScattered) rad = rad + nextDir;  // occluded: radiance += throughput * (Emit + 0)
            if (contAfterShadow) {
                rd = nextDir;
                ++depth;
                state = kNeedClosest;
            } else {
                finished = true;
            }
        }
        if (finished) {
            // GetHitColor * exposureAdjustment, spheres-app.cpp:183; one 12-byte store
            const float expo = K.exposure;
            *reinterpret_cast<float3*>(p.samples + (size_t)q * 3) = make_float3(rad.x * expo, rad.y * expo, rad.z * expo);
            if (p.trav_out) p.trav_out[q] = pathTrav;
            state = kIdle;
        }
        RT_STAMP(ts3);
#ifdef RT_STAMPS
        if (thA != 0) {  // lanes that processed a hit this iteration (uniform enough: lane 0 reports)
            cyHit[3] += thA - ts2;
            cyHit[4] += ts3 - thB;
            cyHit[5] += 1;
        }
#endif
        RT_ACC(cyRefill, ts0, ts1);
        RT_ACC(cyScan, ts1, ts2);
        RT_ACC(cyTrans, ts2, ts3);
#ifdef RT_STAMPS
        cyIters += 1;
#endif
    }

    // counters: wave reduce, one atomic pair per wave
    unsigned long long t = nTrav, s = nSeg;
    for (int off = 32; off > 0; off >>= 1) {
        t += __shfl_down(t, off);
        s += __shfl_down(s, off);
    }
#ifdef RT_STAMPS
    {
        unsigned long long it = dbgScan[2];
        for (int off = 32; off > 0; off >>= 1) it += __shfl_down(it, off);
        if (lane == 0) atomicAdd(&g_dbg[6], it);
    }
#endif
    if (lane == 0) {
        atomicAdd(&p.counters[0], t);
        atomicAdd(&p.counters[1], s);
#ifdef RT_STAMPS
        atomicAdd(&g_dbg[0], cyRefill);
        atomicAdd(&g_dbg[1], cyScan);
        atomicAdd(&g_dbg[2], cyTrans);
        atomicAdd(&g_dbg[3], cyIters);
        atomicAdd(&g_dbg[4], dbgScan[0]);
        atomicAdd(&g_dbg[5], dbgScan[1]);
        atomicAdd(&g_dbg[7], dbgScan[3]);
        atomicAdd(&g_dbg[8], dbgScan[4]);
        atomicAdd(&g_dbg[9], dbgScan[5]);
        atomicAdd(&g_dbg[10], dbgScan[6]);
        atomicAdd(&g_dbg[11], dbgScan[7]);
        atomicAdd(&g_dbg[12], cyHit[0]);
        atomicAdd(&g_dbg[13], cyHit[1]);
        atomicAdd(&g_dbg[14], cyHit[2]);
        atomicAdd(&g_dbg[15], cyHit[3]);
        atomicAdd(&g_dbg[16], cyHit[4]);
        atomicAdd(&g_dbg[17], cyHit[5]);
#endif
    }
}

// ============================================================ ray-generation tables (A1, A9)
// jitter[k] = Halton2D(s0+k; 2,3) (spheres-app.cpp:140), lens[k] = HaltonSampleDisk(k0+k; 4,5) (:152)
__global__ void __launch_bounds__(256) rt_raygen_tables_kernel(float2* jitter, uint32_t s0, uint32_t nJitter, float2* lens, uint32_t k0,
                                                               uint32_t nLens) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nJitter) jitter[k] = make_float2(halton(s0 + k, 2), halton(s0 + k, 3));
    if (k < nLens) {
        float lx, ly;
        halton_disk_4_5(k0 + k, lx, ly);
        lens[k] = make_float2(lx, ly);
    }
}

// ============================================================== ordered accumulation (A16)
// hdr[pixel] += sample(pixel, s) for s = s0 .. s0+spp-1 in that order (spheres-app.cpp:182-183).
__global__ void __launch_bounds__(256) rt_accumulate_kernel(const float* __restrict__ samples, float* __restrict__ hdr, uint32_t npix,
                                                            uint32_t spp) {
    const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    float r = hdr[3 * (size_t)pix], g = hdr[3 * (size_t)pix + 1], b = hdr[3 * (size_t)pix + 2];
    // tiled buffer [tile of 64 pixels][sample][pixel in tile] (path_coordinates): the 64 lanes of a wave read 768
    // contiguous bytes per sample plane; eight planes in flight, the adds stay sequential in s
    const uint32_t nFull = npix >> 6, tile = pix >> 6;
    const uint32_t stride = tile < nFull ? 64u : npix - (nFull << 6);
    const float3* sp = reinterpret_cast<const float3*>(samples) + (size_t)tile * 64u * spp + (pix - (tile << 6));
    uint32_t s = 0;
    for (; s + 8 <= spp; s += 8) {
        float3 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = sp[(size_t)(s + k) * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            r += v[k].x;
            g += v[k].y;
            b += v[k].z;
        }
    }
    for (; s < spp; ++s) {
        const float3 v = sp[(size_t)s * stride];
        r += v.x;
        g += v.y;
        b += v.z;
    }
    hdr[3 * (size_t)pix] = r;
    hdr[3 * (size_t)pix + 1] = g;
    hdr[3 * (size_t)pix + 2] = b;
}

// ====================================================================== resolve (A17)
// hdr / n, ACES fit, gamma 1/2.2, XMStoreColor (spheres-app.cpp:186-214); output R,G,B bytes.
RT_DEV float tonemap_channel(float h, float n) {
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    float color = h / n;
    color = sat1((color * (a * color + b)) / (color * (c * color + d) + e));
    color = rt_powf(color, 1 / 2.2f);
    return color;
}
__global__ void __launch_bounds__(256) rt_resolve_kernel(const float* __restrict__ hdr, uint8_t* __restrict__ ldr, uint32_t npix,
                                                         uint32_t nSamples) {
    const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    const float n = (float)nSamples;
    for (int ch = 0; ch < 3; ++ch) ldr[3 * (size_t)pix + ch] = (uint8_t)rne_u8(tonemap_channel(hdr[3 * (size_t)pix + ch], n));
}

// ================================================================== unit-test kernels
__global__ void k_unit_halton(const uint32_t* index, uint32_t base, uint32_t n, float* out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = halton(index[k], base);
}
__global__ void k_unit_math(uint32_t op, const float* x, const float* y, uint32_t n, float* out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    float r = 0.f;
    if (op == 0) r = rt_sinf(x[k]);
    else if (op == 1) r = rt_cosf(x[k]);
    else if (op == 2) r = rt_powf(x[k], y[k]);
    else if (op == 3) r = rt_tanf(x[k]);
    out[k] = r;
}
__global__ void k_unit_primary(const TraceParams p, const uint32_t* ijs, uint32_t n, float* out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    V3 o, d;
    gen_primary_ray(p, ijs[3 * k], ijs[3 * k + 1], ijs[3 * k + 2], o, d);
    float* w = out + 6 * (size_t)k;
    w[0] = o.x; w[1] = o.y; w[2] = o.z; w[3] = d.x; w[4] = d.y; w[5] = d.z;
}
__global__ void k_unit_closest(const TraceParams p, const float* rays, uint32_t n, float* out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float* r = rays + 6 * (size_t)k;
    const V3 o = v3(r[0], r[1], r[2]), d = v3(r[3], r[4], r[5]);
    float tmin;
    int idx;
    scan_list(p.scan, p.orig, p.n_padded - 4u, o, d, tmin, idx);
    float* w = out + 10 * (size_t)k;
    for (int c = 0; c < 10; ++c) w[c] = 0.f;
    const int oidx = idx >= 0 ? (int)p.orig[idx] : -1;
    w[1] = __int_as_float(oidx);
    if (idx >= 0) {
        const float4 S = p.scan[idx];
        const V3 pos = tmin * d + o;
        const V3 nrm = (pos - v3(S.x, S.y, S.z)) / p.radius[idx];
        w[0] = tmin;
        w[2] = pos.x; w[3] = pos.y; w[4] = pos.z;
        w[5] = nrm.x; w[6] = nrm.y; w[7] = nrm.z;
        w[8] = 0.5f * nrm.x + 0.5f;
        w[9] = 0.5f * nrm.z + 0.5f;
    }
}

// Camera::GetRay for given (uv, lens offset) pairs: in 4 floats, out origin xyz + direction xyz.
__global__ void k_unit_camera(const TraceParams p, const float* uvoff, uint32_t n, float* out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float uvx = uvoff[4 * k], uvy = uvoff[4 * k + 1], lensx = uvoff[4 * k + 2], lensy = uvoff[4 * k + 3];
    V3 o, d;
    camera_get_ray(p, uvx, uvy, lensx, lensy, o, d);
    float* w = out + 6 * (size_t)k;
    w[0] = o.x; w[1] = o.y; w[2] = o.z; w[3] = d.x; w[4] = d.y; w[5] = d.z;
}
// Material::Scatter + Emit + unoccluded DirectionalLight::Shade for one material record.
// in: ray dir 3, pos 3, normal 3, draws 3 (12 floats); out: scattered, atten 3, dir 3, draws used, local 3 (11 floats)
__global__ void k_unit_scatter(const TraceParams p, const rt_material* mat, const float* in, uint32_t n, float* out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float* q = in + 12 * (size_t)k;
    const Mat m = load_material(mat, 0);
    ScriptedDraws draws{{q[9], q[10], q[11]}, 0u};
    V3 atten, dir, local, localOcc;
    const bool sc = scatter_and_shade(p, m, v3(q[0], q[1], q[2]), v3(q[3], q[4], q[5]), v3(q[6], q[7], q[8]), draws, atten, dir, local, localOcc);
    float* w = out + 11 * (size_t)k;
    w[0] = sc ? 1.f : 0.f;
    w[1] = atten.x; w[2] = atten.y; w[3] = atten.z;
    w[4] = dir.x; w[5] = dir.y; w[6] = dir.z;
    w[7] = (float)draws.used;
    w[8] = local.x; w[9] = local.y; w[10] = local.z;
}
// Resolve for given HDR triples (tonemap unit test): in 3 floats, out 3 bytes
__global__ void k_unit_tonemap(const float* hdr, uint32_t n, uint32_t nSamples, uint8_t* out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    for (int ch = 0; ch < 3; ++ch) out[3 * (size_t)k + ch] = (uint8_t)rne_u8(tonemap_channel(hdr[3 * (size_t)k + ch], (float)nSamples));
}

}  // namespace rtd

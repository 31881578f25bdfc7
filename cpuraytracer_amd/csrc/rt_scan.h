// rt_scan.h — the closest-hit list scan: plain and VALU-filtered scans, the split-bf16 matrix-core filter, the wave-pooled
// exact resolve and the pooled hierarchy descent (BvhNode/Sphere::Intersect, ray-tracing.cpp:42-84,174-214).
#pragma once

#include "rt_params.h"

namespace rtd {

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
        const float sq = sqrt_rn(disc);
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
        const float sq = sqrt_rn(e);
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
// (round 1: 768 + 576 entries.  Since the descent culls by span the lists run short: 384 + 384 measures the same, and 320 + 384
// leaves grid10k's LDS image room for the prepared-path cache, +3 %; 256 + 384 loses 6 %.)
#ifndef RT_TREE_WORK
#define RT_TREE_WORK 320
#endif
#ifndef RT_TREE_EXACT
#define RT_TREE_EXACT 384
#endif
constexpr uint32_t kTreeWork = RT_TREE_WORK, kTreeExact = RT_TREE_EXACT, kTreeReserve = 3 * kMaxLevels;  // hierarchy scan: (ray, node) and (ray, sphere) lists
constexpr uint32_t kFarDrain = 128;                                     // hierarchy scan: exact entries that trigger a drain (far limits, below)
#ifndef RT_POOL_A
#define RT_POOL_A 640
#endif
#ifndef RT_POOL_B
#define RT_POOL_B 512
#endif
constexpr uint32_t kPoolA = RT_POOL_A;                                  // pooled resolve: (ray, group) items per pass
constexpr uint32_t kPoolB = RT_POOL_B;                                  // pooled resolve: (ray, sphere) items before a drain
static_assert(kPoolB > 4 * 64 && (kPoolA * 2 + kPoolB * 2) % 16 == 0, "a round may add 256 entries; the keys behind the pools are 8-byte aligned");
constexpr uint32_t kWaveListBytes = kPoolA * 2 + kPoolB * 2 + 64 * 8;      // item pools + per-ray best keys = 2816 B per wave
static_assert(kTreeExact >= kFarDrain + 4 * 64, "a round may add 256 entries to an exact list that holds up to kFarDrain - 1");
constexpr uint32_t kWaveCandBytes = kTreeWork * 4 + kTreeExact * 4 + 64 * 8;  // hierarchy scan: 5888 B per wave
// Cell-grid scan: (ray, slab) items of up to RT_GRID_FEED slabs per ray and feed pass.  Measured on grid10k (4096^2, spp 64; the
// list's bytes come out of the hit stash): 4 slabs / 320 items 8,205 Msamples/s, 6 / 384 8,175, 8 / 512 8,383, 8 / 640 8,474,
// 10 / 640 8,448, 8 / 768 8,326, 12 / 768 8,413, 16 / 1024 8,210; 2 / 320 7,770.  Without the consumers' re-check of the closest hit
// (RT_GRID_ROUNDSKIP): 8 / 640 8,622, 10 / 704 8,667.
#ifndef RT_GRID_FEED
#define RT_GRID_FEED 10
#endif
#ifndef RT_GRID_WORK
#define RT_GRID_WORK 704
#endif
constexpr uint32_t kGridWork = RT_GRID_WORK;
#ifndef RT_GRID_DRAIN
#define RT_GRID_DRAIN 32
#endif
// exact entries that trigger a drain in the grid scan: the walk is front to back, so early hits end it early -- measured on
// grid10k: 128 (the hierarchy's value) -> 64 +1.5 %, 48 +2.3 %, 32 +3.5 %; round 4 (no limits in the step loop): 16 / 24 / 32 / 64 /
// 128: 8,193 / 8,208 / 8,228 / 8,068 / 7,956 Msamples/s
constexpr uint32_t kGridDrain = RT_GRID_DRAIN;
constexpr uint32_t kGridExact = kGridDrain + 4 * 64;  // a step may add 256 entries to an exact list that holds up to kGridDrain - 1
constexpr uint32_t kWaveGridBytes = kGridWork * 4 + kGridExact * 4 + 64 * 8;  // grid scan: 4480 B per wave
static_assert(kWaveGridBytes % 16 == 0 && ((kGridWork + kGridExact) * 4) % 8 == 0, "per-wave regions are float4 aligned, the keys behind the lists 8-byte aligned");
template <int kScan>
constexpr uint32_t wave_region_bytes() { return kScan == 3 ? kWaveGridBytes : (kScan == 2 ? kWaveCandBytes : kWaveListBytes); }
// K of the filter margins (units of eps * a * G; the host folds the same K into each bound): the matrix-core level needs
// 101*16 (exact-path rounding, amplified by the member offsets) + ~600 (split-bf16 operands); levels tested on the VALU
// in f32 need 101*16 + 30; a one-sphere bound (offset 0) needs 16 + 30.
constexpr float kMarginK = 4096.f, kMarginKValu = 2048.f, kMarginKLeaf = 64.f;
constexpr float kMarginRel = kMarginK * 5.9604645e-8f;                 // K * eps
constexpr uint32_t kRayCacheBytes = 64 * 48;                            // per-wave cache of prepared paths
// Flat scan: the LDS copy of the one-sphere bounds keeps each group's four float4 at a stride of FIVE (80 bytes).  Phase A
// reads bound q of (ray, group) items with one ds_read_b128 per q, which the LDS serves in sets of 16 lanes over 64 banks: at a
// stride of 64 bytes the bank quad of bound q is (4 gid + q) mod 16 -- four values whatever the groups are, so 16 lanes with
// different groups collided at least four to a quad (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.21); at 80 bytes it is
// (5 gid + q) mod 16, which takes all sixteen values.  2 KB of LDS for the cover scene.
#ifndef RT_LEAF_STRIDE
#define RT_LEAF_STRIDE 5
#endif
constexpr uint32_t kFlatLeafStride = RT_LEAF_STRIDE;
#ifndef RT_A2
#define RT_A2 1  // flat scan, pooled phase A: two (ray, group) items per lane and step
#endif
#ifndef RT_B2
#define RT_B2 1  // flat scan, exact phase: two (ray, sphere) pairs per lane and step
#endif
#ifndef RT_LIST_WORDS
#define RT_LIST_WORDS 1  // flat scan: candidate lists are pushed word by word (0: the round-3 loops over 64-bit halves)
#endif                    // float4 slots per group in the flat scan's LDS copy (4 = dense)
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

// Square roots of the FILTERS' thresholds (the "behind" threshold bt, the reach of a root's rounding): any value >= the true root
// keeps a filter conservative, so the hardware's 1-ulp v_sqrt_f32, nudged up, stands in for the correctly rounded one (16
// instructions).  Inputs below 1e-30 (which the instruction would flush) are raised to it: the result only grows.
#ifndef RT_FILTER_SQRT_RAW
#define RT_FILTER_SQRT_RAW 1
#endif
RT_DEV float filter_sqrt_up(float x) {
#if RT_FILTER_SQRT_RAW && defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(__builtin_fmaxf(x, 1e-30f)) * 1.000001f;
#else
    return __builtin_sqrtf(x);
#endif
}

// t of Sphere::Intersect for a pooled (ray, sphere) pair (ray-tracing.cpp:56, :69): the first root, or the second where the first
// does not exceed the bias.  RT_EXACT_MARKSTEIN: the two IEEE quotients through Markstein's correction (rt_device_math.h) when
// every lane's a is in [2^-19, 2^100] and its numerators are below 2^100: a quotient that is normal is then the IEEE quotient,
// and one that is not (|t| < 2^-80) fails `t > 0.001` in either form and is never kept (rt_shade.h root_exceeds_bias argues the
// same for the shadow rays).
#ifndef RT_EXACT_MARKSTEIN
#define RT_EXACT_MARKSTEIN 1
#endif
RT_DEV float pooled_root(float b, float sq, float ra) {
#if RT_MARKSTEIN && RT_EXACT_MARKSTEIN
    const bool ok = (__float_as_uint(ra) - 0x36000000u) <= 0x3b800000u && (__builtin_fabsf(b) + sq) < 0x1p100f;
    if (__builtin_expect(__ballot(!ok) == 0ull, 1)) {
        const float ya = recip_rn(ra);
        float t = div_rn(-b - sq, ra, ya);
        if (!(t > 0.001f)) t = div_rn(-b + sq, ra, ya);
        return t;
    }
#endif
    {
        RT_SITE(M_ROOT_SLOW);
        float t = (-b - sq) / ra;
        if (!(t > 0.001f)) t = (-b + sq) / ra;
        return t;
    }
}

// min(x, p) for p > 0 as ONE integer instruction: a negative float's pattern is a negative integer (below every
// positive one) and positive floats order like their patterns.  (fminf would add a canonicalising v_max per operand.)
RT_DEV float min_with_positive(float x, float p) {
    const int xi = __float_as_int(x), pi = __float_as_int(p);
    return __int_as_float(xi < pi ? xi : pi);
}

// Filter decision for this lane's 16 rows of one tile: the ray may hit the group unless F = b~^2 - t < 0 (t = a*cc~ - M)
// or the group is surely behind the origin.  "Behind" = the origin is outside the inflated bound (t > 0, which already
// includes the margin) and the centre is behind it by more than the rounding of b~ (b~ > bthr, bthr =
// 1e-4 sqrt(a) (|o| + max(|C|+R)) >= 185x the error bound of b~): then every point of the bound, hence of its member
// spheres, has t < 0 and the reference accepts no root (bias 0.001, ray-tracing.cpp:52).
// All three conditions are sign bits: one 3-input bit operation forms "rejected", one v_alignbit appends its sign bit
// to the lane's bitmap word (four VALU operations per (ray, group) pair, no branches, no LDS).
#ifndef RT_PK_POST
#define RT_PK_POST 0  // 1: the fma of two (ray, group) pairs as one v_pk_fma_f32 (half the issue slots of that third of the filter's arithmetic)
#endif
RT_DEV void mfma_post(const f32x16& Tb, const f32x16& Tg, float bthr, uint32_t& rejectedBits) {
#if RT_PK_POST
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
        const f32x2 m = {min_with_positive(Tb[e], bthr), min_with_positive(Tb[e + 1], bthr)};
        const f32x2 b = {Tb[e], Tb[e + 1]};
        const f32x2 t = {Tg[e], Tg[e + 1]};
        const f32x2 g = __builtin_elementwise_fma(m, b, -t);
        rejectedBits = __builtin_amdgcn_alignbit(rejectedBits, __float_as_uint(g[0]), 31);
        rejectedBits = __builtin_amdgcn_alignbit(rejectedBits, __float_as_uint(g[1]), 31);
    }
    return;
#endif
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float t = Tg[e];  // a*cc~ - M: the per-ray constant is part of the contraction
        // One sign test for both conditions: rejected <=> t > b~ min(b~, bthr).  For b~ <= bthr that is F = b~^2 - t < 0;
        // for b~ > bthr (centre behind the origin) it is t > b~ bthr, a hair weaker than "origin outside the inflated bound"
        // (t >= 0; bthr ~ 1e-3, so only origins grazing the bound from outside are kept) and implies it: the candidate set
        // only grows, the filter stays conservative.  Three VALU operations per (ray, group) pair: min, fma, alignbit.
        const float g = __builtin_fmaf(min_with_positive(Tb[e], bthr), Tb[e], -t);
        rejectedBits = __builtin_amdgcn_alignbit(rejectedBits, __float_as_uint(g), 31);  // (bits << 1) | sign(g)
    }
}

// The filter formula on the VALU (any rounding; the bounds' margins cover it): sign bit set = the bound (C, W) cannot
// contain an acceptable root of the ray.  g = -2a o (per ray, rounded once per component: <= 2 eps a |o||C| more rounding
// than the un-scaled form, inside every K), dO = d.o, cr = a|o|^2 (1 - 2 K eps), bt = the "behind" threshold.
// b = d.o - d.C and t = a (|C|^2 - Rf^2) - 2a o.C + cr are two fused chains of three and four operations; with the
// min / fma decision of mfma_post that is nine VALU operations per bound.
RT_DEV int bound_rejected(const float4 B, V3 g, V3 d, float a, float dO, float cr, float bt) {
    const float b = __builtin_fmaf(-d.z, B.z, __builtin_fmaf(-d.y, B.y, __builtin_fmaf(-d.x, B.x, dO)));
    const float t = __builtin_fmaf(g.z, B.z, __builtin_fmaf(g.y, B.y, __builtin_fmaf(g.x, B.x, __builtin_fmaf(a, B.w, cr))));
    return __float_as_int(__builtin_fmaf(min_with_positive(b, bt), b, -t));  // as mfma_post: rejected <=> t > b min(b, bt)
}

// ... and the same with a FAR LIMIT (hierarchy descent): u = a * tmax * (1 + 2^-10), tmax = the ray's closest hit so far (+inf:
// none).  The bound is also rejected when it lies wholly beyond tmax: the ray is still approaching its centre at tmax
// (b + u < -bt: by more than the rounding of b and u) and the point at tmax is outside the inflated bound (f(tmax) > 0, with
// f(t) a = t~ + a t (2b + a t): the filter's own margin-carrying t~, so "outside" holds for the bound inflated by M).  f falls
// on [0, tmax] and is positive at its end, so no point of the segment is in the inflated bound -- and the hit point of every
// root the reference accepts for a member sphere is (that is what makes the ordinary test conservative: |f_i(t^)| <= E''/a,
// DESIGN.md 5.1).  Such roots are > tmax (1 + 2^-10) > tmax: they can neither win nor tie.  Rounding: u and b + u are
// accurate to 3 eps (|b| + u) <= 6 eps sqrt(a) (|o| + |o'|), o' the hit point at tmax, far below bt unless o' is thousands of
// scene sizes away -- and then b + u < 0 with the point outside means the LINE misses the bound, which the ordinary test
// rejects anyway; the three extra roundings of f(tmax) a are <= 8 eps a G (a tmax^2 <= 2|o|^2 + 2|o'|^2 and a bound that
// o' is close to has |C| ~ |o'|), inside every K.  tests/test_filter_margin_cpu.py emulates it against f64 roots.
RT_DEV int bound_rejected_far(const float4 B, V3 g, V3 d, float a, float dO, float cr, float bt, float u) {
    const float b = __builtin_fmaf(-d.z, B.z, __builtin_fmaf(-d.y, B.y, __builtin_fmaf(-d.x, B.x, dO)));
    const float t = __builtin_fmaf(g.z, B.z, __builtin_fmaf(g.y, B.y, __builtin_fmaf(g.x, B.x, __builtin_fmaf(a, B.w, cr))));
    const float dec = __builtin_fmaf(min_with_positive(b, bt), b, -t);
    const float bu = b + u;
    const float fo = __builtin_fmaf(u, b + bu, t);
    return (bu < -bt && fo > 0.f) ? -1 : __float_as_int(dec);
}

// ... and with a NEAR limit as well (un = a * tnear * (1 - 2^-10), or -inf): the bound lies wholly before tnear when the ray has
// passed its centre there (b + un > bt) and the point at tnear is outside the inflated bound.
RT_DEV int bound_rejected_span(const float4 B, V3 g, V3 d, float a, float dO, float cr, float bt, float un, float uf) {
    const float b = __builtin_fmaf(-d.z, B.z, __builtin_fmaf(-d.y, B.y, __builtin_fmaf(-d.x, B.x, dO)));
    const float t = __builtin_fmaf(g.z, B.z, __builtin_fmaf(g.y, B.y, __builtin_fmaf(g.x, B.x, __builtin_fmaf(a, B.w, cr))));
    const float dec = __builtin_fmaf(min_with_positive(b, bt), b, -t);
    const float bf = b + uf;
    const float ff = __builtin_fmaf(uf, b + bf, t);
    const float bn = b + un;
    const float fn = __builtin_fmaf(un, b + bn, t);
    return ((bf < -bt && ff > 0.f) || (bn > bt && fn > 0.f)) ? -1 : __float_as_int(dec);
}

// Cross-lane hand-off through LDS inside ONE wave (work lists, closest-hit keys, the prepared-path cache): one set of
// lanes stores, other lanes of the same wave load right after.  The hardware executes a wave's LDS operations in order;
// this stops the COMPILER from moving may-alias accesses across the hand-off (it emits no instruction beyond, at most, a
// wait the loads needed anyway).
RT_DEV void wave_lds_handoff() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
                           uint32_t nTop, const float4* __restrict__ tree, const uint32_t* levelOff, uint32_t nLevels, float boundNorm,
                           const unsigned long long* singleMask, uint32_t nAlways, const float* treeBox, V3 o, V3 d, bool live, float& tmin, int& idx, uint16_t* waveCand, uint32_t lane, unsigned long long* dbg) {
    RT_SITE(S_SCAN);
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
    const float bt = 1e-4f * filter_sqrt_up(a) * (filter_sqrt_up(oo) + boundNorm);
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
        RT_SITE(S_TILEPAIR);
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
        wave_lds_handoff();
        uint32_t cntB = 0;
#if RT_LIST_WORDS
        // The candidates as four 32-bit words: bit N (from the top) of word k is group kBase[k] + N + (N & 16), kBase = 0, 64, 16, 80.
        // Every push loop below walks ONE word (v_ffbh, clear, two operations for the item, the store): a loop over a 64-bit half
        // ran 14 VALU instructions per candidate of the wave's BUSIEST lane; four word loops run ~6 per candidate of the busiest
        // lane of each word.
        uint32_t cw0 = (uint32_t)(cur >> 32), cw1 = (uint32_t)cur, cw2 = (uint32_t)(nxt >> 32), cw3 = (uint32_t)nxt;
        auto pushWord = [&](uint32_t w, uint32_t tagBase, uint32_t scale, uint16_t*& wp) __attribute__((always_inline)) {
            while (w != 0u) {
                RT_SITE(S_PUSH_WORD);
                const uint32_t N = (uint32_t)__builtin_clz(w);
                w ^= 0x80000000u >> N;
                *wp++ = (uint16_t)(tagBase + scale * (N + (N & 16u)));
            }
        };
        {   // Candidate groups of ONE sphere (the floor, the big spheres: a third of all candidates) skip the sphere-level filter:
            // its only finding would be the sliver between the two margins, and an exact slot costs no more than a filter slot.
            const uint32_t s0 = cw0 & (uint32_t)(singleMask[0] >> 32), s1 = cw1 & (uint32_t)singleMask[0];
            const uint32_t s2 = cw2 & (uint32_t)(singleMask[1] >> 32), s3 = cw3 & (uint32_t)singleMask[1];
            const uint32_t nS = (uint32_t)(__builtin_popcount(s0) + __builtin_popcount(s1) + __builtin_popcount(s2) + __builtin_popcount(s3));
            const uint32_t inclS = wave_inclusive_sum(nS);
            const uint32_t totalS = (uint32_t)__builtin_amdgcn_readlane((int)inclS, 63);
            if (totalS != 0u && totalS <= kPoolB - 4u * kWaveSize) {  // (more than that: they take the ordinary way)
                RT_SITE(S_SINGLE);
                cw0 ^= s0; cw1 ^= s1; cw2 ^= s2; cw3 ^= s3;
                uint16_t* wp = poolB + (inclS - nS);
                const uint32_t tag = lane << 10;  // entry = ray << 10 | scan entry, scan entry = 4 * group
                pushWord(s0, tag, 4u, wp);
                pushWord(s1, tag + 4u * 64u, 4u, wp);
                pushWord(s2, tag + 4u * 16u, 4u, wp);
                pushWord(s3, tag + 4u * 80u, 4u, wp);
                cntB = totalS;
            }
        }
        const uint32_t nMine = (uint32_t)(__builtin_popcount(cw0) + __builtin_popcount(cw1) + __builtin_popcount(cw2) + __builtin_popcount(cw3));
        bool pending = nMine != 0u;
#else
        {   // Candidate groups of ONE sphere (the floor, the big spheres: a third of all candidates) skip the sphere-level filter:
            // its only finding would be the sliver between the two margins, and an exact slot costs no more than a filter slot.
            const unsigned long long sCur = cur & singleMask[0], sNxt = nxt & singleMask[1];
            const uint32_t nS = (uint32_t)(__popcll(sCur) + __popcll(sNxt));
            const uint32_t inclS = wave_inclusive_sum(nS);
            const uint32_t totalS = (uint32_t)__builtin_amdgcn_readlane((int)inclS, 63);
            if (totalS != 0u && totalS <= kPoolB - 4u * kWaveSize) {  // (more than that: they take the ordinary way)
                RT_SITE(S_SINGLE);
                cur &= ~sCur;
                nxt &= ~sNxt;
                uint16_t* wp = poolB + (inclS - nS);
                const uint32_t tag = lane << 10;
                unsigned long long m0 = sCur, m1 = sNxt;
                while (m0 != 0ull) {
                    RT_SITE(S_SINGLE_PUSH0);
                    const uint32_t N = (uint32_t)__builtin_clzll(m0);
                    m0 &= ~(0x8000000000000000ull >> N);
                    *wp++ = (uint16_t)(tag | (4u * (N + (N & 48u))));
                }
                while (m1 != 0ull) {
                    RT_SITE(S_SINGLE_PUSH1);
                    const uint32_t N = (uint32_t)__builtin_clzll(m1);
                    m1 &= ~(0x8000000000000000ull >> N);
                    *wp++ = (uint16_t)(tag | (4u * (16u + N + (N & 48u))));
                }
                cntB = totalS;
            }
        }
        const uint32_t nMine = (uint32_t)(__popcll(cur) + __popcll(nxt));
        bool pending = nMine != 0u;
#endif
        RT_STAMP(ta0);
        // phase B over the current contents of poolB (wave-uniform count)
#if RT_B2
        auto drainB = [&]() {
            RT_SITE(S_DRAIN);
            wave_lds_handoff();  // poolB entries written by other lanes
            auto exactPair = [&](uint32_t ent, bool has) __attribute__((always_inline)) {
                const uint32_t r = ent >> 10, cand = ent & 1023u;
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
                const float sq = sqrt_rn(e > 0.f ? e : 1.f);  // (a root is read only where e > 0: the other lanes keep the wave on the fast path)
                const float t = pooled_root(b, sq, ra);  // ray-tracing.cpp:56, :69
                // `e > 0` is the reference's own test (ray-tracing.cpp:54); `t < inf` is the scan's initial tmin
                if (has && e > 0.f && t > 0.001f && t < __builtin_inff()) {
                    RT_SITE(S_BMIN);
                    const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)((orig[cand] << 16) | cand);
                    __hip_atomic_fetch_min(best + r, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            };
            for (uint32_t base = 0; base < cntB; base += 2u * kWaveSize) {
                RT_SITE(S_BSTEP);
                const uint32_t k0 = base + lane, k1 = k0 + (uint32_t)kWaveSize;
                const bool has0 = k0 < cntB, has1 = k1 < cntB;
                const uint32_t ent0 = has0 ? (uint32_t)poolB[k0] : 0u, ent1 = has1 ? (uint32_t)poolB[k1] : 0u;
#ifdef RT_STAMPS
                dbg[7] += 1;
#endif
                exactPair(ent0, has0);
                if (base + (uint32_t)kWaveSize < cntB) {  // wave-uniform
                    RT_SITE(S_BSTEP2);
                    exactPair(ent1, has1);
                }
            }
            cntB = 0;
            wave_lds_handoff();  // poolB may be refilled from here on
        };
#else
        auto drainB = [&]() {
            RT_SITE(S_DRAIN);
            wave_lds_handoff();  // poolB entries written by other lanes
            for (uint32_t base = 0; base < cntB; base += kWaveSize) {
                RT_SITE(S_BSTEP);
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
                const float sq = sqrt_rn(e > 0.f ? e : 1.f);  // (a root is read only where e > 0: the other lanes keep the wave on the fast path)
                const float t = pooled_root(b, sq, ra);  // ray-tracing.cpp:56, :69
                // `e > 0` is the reference's own test (ray-tracing.cpp:54); `t < inf` is the scan's initial tmin
                if (has && e > 0.f && t > 0.001f && t < __builtin_inff()) {
                    RT_SITE(S_BMIN);
                    const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)((orig[cand] << 16) | cand);
                    __hip_atomic_fetch_min(best + r, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            cntB = 0;
            wave_lds_handoff();  // poolB may be refilled from here on
        };
#endif
        while (__ballot(pending) != 0ull) {
            RT_SITE(S_PASS);
            // lanes whose items fit into the list this pass: a prefix of the pending lanes
            const uint32_t incl = wave_inclusive_sum(pending ? nMine : 0u);
            const bool take = pending && incl <= kPoolA;
            const uint64_t takeMask = __ballot(take);
            const uint32_t lastLane = 63u - (uint32_t)__builtin_clzll(takeMask);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)lastLane);
#if RT_LIST_WORDS
            if (take) {
                RT_SITE(S_TAKE);
                uint16_t* wp = poolA + (incl - nMine);
                const uint32_t tag = lane << 7;  // item = ray << 7 | group
                pushWord(cw0, tag, 1u, wp);
                pushWord(cw1, tag + 64u, 1u, wp);
                pushWord(cw2, tag + 16u, 1u, wp);
                pushWord(cw3, tag + 80u, 1u, wp);
                pending = false;
            }
#else
            if (take) {
                RT_SITE(S_TAKE);
                // one loop per bitmap half (leading-zero order; bit N of half h is group 16 h + N + (N & 48))
                uint16_t* wp = poolA + (incl - nMine);
                const uint32_t tag = lane << 7;
                while (cur != 0ull) {
                    RT_SITE(S_TAKE_PUSH0);
                    const uint32_t N = (uint32_t)__builtin_clzll(cur);
                    cur &= ~(0x8000000000000000ull >> N);
                    *wp++ = (uint16_t)(tag | (N + (N & 48u)));
                }
                while (nxt != 0ull) {
                    RT_SITE(S_TAKE_PUSH1);
                    const uint32_t N = (uint32_t)__builtin_clzll(nxt);
                    nxt &= ~(0x8000000000000000ull >> N);
                    *wp++ = (uint16_t)(tag | (16u + N + (N & 48u)));
                }
                pending = false;
            }
#endif
            wave_lds_handoff();  // poolA items written by their owner lanes, read by any lane
#if RT_A2
            // TWO items per lane and step: the two chains (item -> ray fetch + bounds -> tests) are independent, so their LDS round
            // trips overlap and a scan takes about half the steps; the survivors of both share one prefix sum.  The exact list
            // is drained when THIS step's survivors would not fit (no worst-case reserve of 4 x 128 entries).
            for (uint32_t base = 0; base < total; base += 2u * kWaveSize) {
                RT_SITE(S_ASTEP);
#ifdef RT_STAMPS
                dbg[6] += 1;
#endif
                auto testItem = [&](uint32_t item, bool has) __attribute__((always_inline)) -> uint32_t {
                    const uint32_t r = item >> 7, gid = item & 127u;
                    const V3 fg = v3(lane_fetch(r, gx), lane_fetch(r, gy), lane_fetch(r, gz));
                    const V3 fd = v3(lane_fetch(r, d.x), lane_fetch(r, d.y), lane_fetch(r, d.z));
                    const float fa = lane_fetch(r, a), fdO = lane_fetch(r, dO), fcr = lane_fetch(r, crLeaf), fbt = lane_fetch(r, bt);
                    const float4* lb = leaf + kFlatLeafStride * gid;
                    uint32_t rb = 0u;
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q)
                        rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected(lb[q], fg, fd, fa, fdO, fcr, fbt), 31);
                    return has ? (~rb & 15u) : 0u;  // bit 3-q = sphere q of the group
                };
                const uint32_t k0 = base + lane, k1 = k0 + (uint32_t)kWaveSize;
                const bool has0 = k0 < total, has1 = k1 < total;
                const uint32_t item0 = has0 ? (uint32_t)poolA[k0] : 0u;
                const bool second = base + (uint32_t)kWaveSize < total;  // wave-uniform: a last, short step tests one item per lane
                const uint32_t item1 = has1 ? (uint32_t)poolA[k1] : 0u;
                const uint32_t m0 = testItem(item0, has0);
                uint32_t m1 = 0u;
                if (second) {
                    RT_SITE(S_ASTEP2);
                    m1 = testItem(item1, has1);
                }
                const uint32_t nh = (uint32_t)(__builtin_popcount(m0) + __builtin_popcount(m1));
                const uint32_t incl = wave_inclusive_sum(nh);
                const uint32_t totalH = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (cntB + totalH > kPoolB) drainB();  // (totalH <= 512 = kPoolB: fits an empty list)
                uint16_t* wp = poolB + cntB + (incl - nh);
                auto pushSurvivors = [&](uint32_t mm, uint32_t item) __attribute__((always_inline)) {
                    const uint32_t tagEntry = (item >> 7) << 10 | (4u * (item & 127u) + 3u);
                    while (mm != 0u) {
                        RT_SITE(S_APUSH);
                        const uint32_t bit = 31u - (uint32_t)__builtin_clz(mm);
                        mm &= ~(1u << bit);
                        *wp++ = (uint16_t)(tagEntry - bit);
                    }
                };
                pushSurvivors(m0, item0);
                pushSurvivors(m1, item1);
                cntB += totalH;
            }
#else
            for (uint32_t base = 0; base < total; base += kWaveSize) {
                RT_SITE(S_ASTEP);
#ifdef RT_STAMPS
                dbg[6] += 1;
#endif
                const uint32_t k = base + lane;
                const bool has = k < total;
                const uint32_t item = has ? (uint32_t)poolA[k] : 0u;
                const uint32_t r = item >> 7, gid = item & 127u;
                const V3 fg = v3(lane_fetch(r, gx), lane_fetch(r, gy), lane_fetch(r, gz));
                const V3 fd = v3(lane_fetch(r, d.x), lane_fetch(r, d.y), lane_fetch(r, d.z));
                const float fa = lane_fetch(r, a), fdO = lane_fetch(r, dO), fcr = lane_fetch(r, crLeaf), fbt = lane_fetch(r, bt);
                const float4* lb = leaf + kFlatLeafStride * gid;  // (the flat scan's strided LDS copy: stage_scene)
                uint32_t rb = 0u;
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q)
                    rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected(lb[q], fg, fd, fa, fdO, fcr, fbt), 31);
                const uint32_t m = has ? (~rb & 15u) : 0u;  // bit 3-q = sphere q of the group
                {   // one prefix sum over the lanes' survivor counts, then every lane appends its own (0..4) entries
                    const uint32_t nh = (uint32_t)__builtin_popcount(m);
                    const uint32_t incl = wave_inclusive_sum(nh);
                    const uint32_t totalH = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    uint16_t* wp = poolB + cntB + (incl - nh);
                    uint32_t mm = m;
                    while (mm != 0u) {
                        RT_SITE(S_APUSH);
                        const uint32_t bit = 31u - (uint32_t)__builtin_clz(mm);
                        mm &= ~(1u << bit);
                        *wp++ = (uint16_t)(r << 10 | (4u * gid + (3u - bit)));
                    }
                    cntB += totalH;
                }
                if (cntB > kPoolB - 4u * kWaveSize) drainB();
            }
#endif
            wave_lds_handoff();  // the next pass overwrites poolA
        }
        RT_STAMP(ta1);
        RT_ACC(dbg[4], ta0, ta1);
        drainB();
        const unsigned long long mineKey = best[lane];  // behind drainB's closing hand-off: every lane's minimum is in
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
        wave_lds_handoff();
        const uint32_t topLevel = nLevels - 1u;
        const float aoo = a * oo;
        uint32_t nWork = 0, nExact = 0;
        // The ray clipped to the box around every sphere in the hierarchy: roots can only lie in [tn, tf] (widened by 2^-10 and
        // by 16 eps of the coordinates' magnitude in position), so the descent gets a near limit and a second far limit -- a flat
        // layer of spheres (the cover scene's kind) is entered late and left early by most rays -- and a ray that misses the box
        // has no candidates at all.  minNum/maxNum ignore the NaN of 0 * inf (an axis the ray is parallel to and inside of).
        float boxUn = -__builtin_inff(), boxUf = __builtin_inff();
        if (treeBox) {
            float tn = 0.f, tf = __builtin_inff();
            const float oc[3] = {o.x, o.y, o.z}, dc[3] = {d.x, d.y, d.z};
            // how far outside its sphere the hit point of an ACCEPTED root can lie: the reference's discriminant is off by at most
            // E <= 16 eps a G, G = 2|o|^2 + 2(|c| + r)^2 + r^2 <= 2|o|^2 + 3 A^2, so the point is within delta of the surface with
            // (r + delta)^2 <= r^2 + E''/a: delta <= min(sqrt(X), X / (2 r_min)), X = 32 eps (2|o|^2 + 3 A^2) (twice E for the root's
            // own square root and division)
            const float X = 32.f * 5.9604645e-8f * __builtin_fmaf(2.f, oo, treeBox[7]);
            const float reach = __builtin_fminf(filter_sqrt_up(X), X * treeBox[8]);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const float pad = reach + 1e-6f * (__builtin_fabsf(oc[ax]) + treeBox[6]);
                const float inv = __builtin_amdgcn_rcpf(dc[ax]);
                const float t0 = ((treeBox[ax] - pad) - oc[ax]) * inv, t1 = ((treeBox[3 + ax] + pad) - oc[ax]) * inv;
                tn = __builtin_fmaxf(tn, __builtin_fminf(t0, t1));
                tf = __builtin_fminf(tf, __builtin_fmaxf(t0, t1));
            }
            tn *= 1.f - 0x1p-10f;
            tf *= 1.f + 0x1p-10f;
            if (tn > tf || !(tf > 0.f)) {  // the ray misses the box (or leaves it behind its origin)
                cur = 0ull;
                nxt = 0ull;
            }
            if (tn > 0.f) boxUn = a * tn;
            boxUf = a * tf;
        }
        auto drainExact = [&]() {
            wave_lds_handoff();  // exact-list entries written by other lanes
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
                const float sq = sqrt_rn(e > 0.f ? e : 1.f);  // (a root is read only where e > 0: the other lanes keep the wave on the fast path)
                const float t = pooled_root(b, sq, ra);  // ray-tracing.cpp:56, :69
                if (has && e > 0.f && t > 0.001f && t < __builtin_inff()) {
                    const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)((orig[cand] << 16) | cand);
                    __hip_atomic_fetch_min(best + r, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            nExact = 0;
            wave_lds_handoff();  // the exact list may be refilled from here on
        };
        // the big spheres are not in the hierarchy (rt_params.h n_always): one exact slot per live ray each
        for (uint32_t q = 0; q < nAlways; ++q) {
            if (nExact + (uint32_t)kWaveSize > kTreeExact) drainExact();
            const uint64_t lm = __ballot(live);
            if (live) exact[nExact + prefix_count(lm)] = lane << 16 | (4u * q);
            nExact += (uint32_t)__popcll(lm);
        }
#ifndef RT_NO_FAR_LIMIT
        if (nExact != 0u) drainExact();  // their hits are the far limits of the descent from its first round on
#else
        if (nExact > kTreeExact - 4u * kWaveSize) drainExact();  // (more than five of them: keep a round's room)
#endif
        for (;;) {
            // feed: when fewer than a round's worth of pairs is listed, every lane with top-level candidates left adds one
            // (until the round is full or nobody has a candidate left: a fed candidate may be dropped, below)
            while (nWork < (uint32_t)kWaveSize && __ballot((cur | nxt) != 0ull) != 0ull) {
                uint32_t t = 0;
                bool add = next_candidate(cur, nxt, hOff, t);
                if (add) {
                    // the lane owns this ray: the candidate node ITSELF against the ray's span (closest hit so far, box entry and
                    // exit), before anybody visits it -- one bound test instead of a round's slot and four child tests
                    const uint32_t tb = reinterpret_cast<const uint32_t*>(best + lane)[1];
                    float fu = tb < 0x7f800000u ? (a * __uint_as_float(tb)) * (1.f + 0x1p-10f) : __builtin_inff();
                    fu = __builtin_fminf(fu, boxUf);
                    const float4 Bn = tree[levelOff[topLevel] + t];
                    const float crTop = aoo * (1.f - 2.f * kMarginK * 5.9604645e-8f);  // the top level's bounds carry the matrix-core K
                    add = bound_rejected_span(Bn, v3(gx, gy, gz), d, a, dO, crTop, bt, boxUn, fu) >= 0;
                }
                const uint64_t am = __ballot(add);
                if (add) work[nWork + prefix_count(am)] = lane << 20 | topLevel << 16 | t;
                nWork += (uint32_t)__popcll(am);
            }
            if (nWork == 0u) break;
            wave_lds_handoff();  // work-list entries pushed by other lanes (feed above, survivors of the last round)
            const uint32_t room = nWork < kTreeWork - kTreeReserve ? kTreeWork - kTreeReserve - nWork : 0u;
            uint32_t np = room / 3u;
            np = np < 1u ? 1u : np;
            np = np > (uint32_t)kWaveSize ? (uint32_t)kWaveSize : np;
            np = np > nWork ? nWork : np;
            const bool has = lane < np;
            const uint32_t ent = has ? work[nWork - 1u - lane] : 0u;
            nWork -= np;
            wave_lds_handoff();  // popped slots are free for this round's pushes
            const uint32_t r = ent >> 20, lvl = (ent >> 16) & 7u, j = ent & 0xffffu;
            const V3 fg = v3(lane_fetch(r, gx), lane_fetch(r, gy), lane_fetch(r, gz));
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
            uint32_t rb = 0u;
#ifndef RT_NO_FAR_LIMIT
            // far limit: the ray's closest hit so far (high word of its key in LDS; a stale value is larger, hence still valid)
            const uint32_t tbits = reinterpret_cast<const uint32_t*>(best + r)[1];
            float fu = tbits < 0x7f800000u ? (fa * __uint_as_float(tbits)) * (1.f + 0x1p-10f) : __builtin_inff();
            if (treeBox) {
                const float fun = lane_fetch(r, boxUn);
                fu = __builtin_fminf(fu, lane_fetch(r, boxUf));
                rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected_span(B0, fg, fd, fa, fdO, fcr, fbt, fun, fu), 31);
                rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected_span(B1, fg, fd, fa, fdO, fcr, fbt, fun, fu), 31);
                rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected_span(B2, fg, fd, fa, fdO, fcr, fbt, fun, fu), 31);
                rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected_span(B3, fg, fd, fa, fdO, fcr, fbt, fun, fu), 31);
            } else {
            rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected_far(B0, fg, fd, fa, fdO, fcr, fbt, fu), 31);
            rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected_far(B1, fg, fd, fa, fdO, fcr, fbt, fu), 31);
            rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected_far(B2, fg, fd, fa, fdO, fcr, fbt, fu), 31);
            rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected_far(B3, fg, fd, fa, fdO, fcr, fbt, fu), 31);
            }
            const uint32_t m = has ? (~rb & 15u) : 0u;  // bit 3-q = child 4j + q
            if (nExact >= kFarDrain) drainExact();  // early and often: every exact round may pull the far limits in
#else
            rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected(B0, fg, fd, fa, fdO, fcr, fbt), 31);
            rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected(B1, fg, fd, fa, fdO, fcr, fbt), 31);
            rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected(B2, fg, fd, fa, fdO, fcr, fbt), 31);
            rb = __builtin_amdgcn_alignbit(rb, (uint32_t)bound_rejected(B3, fg, fd, fa, fdO, fcr, fbt), 31);
            const uint32_t m = has ? (~rb & 15u) : 0u;  // bit 3-q = child 4j + q
            if (nExact > kTreeExact - 4u * kWaveSize) drainExact();
#endif
            {   // one prefix sum per list over the lanes' survivor counts (a lane's survivors all go to the same list), then
                // every lane appends its own 0..4 entries -- instead of eight ballots and conditional stores per round
                const uint32_t nh = (uint32_t)__builtin_popcount(m);
                const uint32_t nW = internal ? nh : 0u, nE = internal ? 0u : nh;
                const uint32_t inclW = wave_inclusive_sum(nW), inclE = wave_inclusive_sum(nE);
                const uint32_t totW = (uint32_t)__builtin_amdgcn_readlane((int)inclW, 63);
                const uint32_t totE = (uint32_t)__builtin_amdgcn_readlane((int)inclE, 63);
                uint32_t* wp = internal ? work + nWork + (inclW - nW) : exact + nExact + (inclE - nE);
                const uint32_t tag = internal ? (r << 20 | (lvl - 1u) << 16) : (r << 16);
                uint32_t mm = m;
                while (mm != 0u) {
                    const uint32_t bit = 31u - (uint32_t)__builtin_clz(mm);
                    mm &= ~(1u << bit);
                    *wp++ = tag | (4u * j + (3u - bit));
                }
                nWork += totW;
                nExact += totE;
            }
        }
        drainExact();
        const unsigned long long mineKey = best[lane];  // behind drainExact's closing hand-off
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

// ------------------------------------------------ list scan through a uniform cell grid (large flat scenes)
// Scenes too large for the flat matrix-core filter used to descend a hierarchy of bounding spheres (above): ~100 bound tests
// per ray on the 10,004-sphere scene.  Such scenes are layers of small spheres: the host sorts the small spheres by their
// HOME CELL in a uniform 2-D grid over the layer's two long axes (cell index iu * nv + iv, so the cells of one u-slab are
// contiguous in v, and so are their spheres in the scan table: a run of cells IS a run of scan entries, no index lists), and
// a ray only meets the spheres whose home cell lies within D cells (Chebyshev) of the part of the ray inside the layer's box:
//   * the ray is clipped to the box around the small spheres, padded per ray by `reach`, the distance an accepted root's hit
//     point can lie outside its sphere (the argument of the hierarchy's box clip above): roots lie in [tn, tf];
//   * the hit point of an accepted root of sphere i is within r_i + reach of c_i, so its home cell is within
//     D = (r_max + reach) / h + slack cells of the segment's projection -- the walk below visits EVERY cell within D of the
//     segment (u-slab by u-slab, front to back; within a slab the v-range of the segment over the slab widened by D), a
//     superset computed with slack for its own float rounding;
//   * the work is POOLED over the wave like the other scans: a ray lists (ray, slab) items, up to four per feed pass, front to
//     back; any lane takes an item, fetches that ray from its owner (ds_bpermute), computes the slab's rows, and tests the
//     one-sphere bounds of the spheres of those cells, four per step (the same conservative test, near and far limits
//     included, as the hierarchy's last level); the survivors go to the exact list, drained by the same pooled
//     reference-order phase with the ds_min_u64 merge.
// The big spheres (at most eight) are tested exactly for every live ray first; their hits are the first far limits, and a ray
// stops walking once the next slab starts beyond its closest hit so far.  Which spheres are TESTED never changes a result:
// the exact phase is Sphere::Intersect in the reference's order and the merge is (t, original index).
struct GridParams {
    const uint16_t* cellStart;  // [nu * nv + 1] first scan entry of each cell (cell = iu * nv + iv)
    uint32_t nu, nv;
    uint32_t axU, axV;          // which coordinates span the grid (the third is the layer's thin axis)
    float g0u, g0v, invH;       // grid origin and 1 / cell size
    float rmaxOverH;            // largest radius among the grid's spheres, in cells
    float bigNorm;              // max |c| + r over the big spheres (the scale of their bounds' behind-the-origin threshold)
};
// QUANTISED one-sphere bounds of the grid's small spheres: 4 bytes per scan entry, in LDS (40 KB for 10,004 spheres).
//   byte 0      du   position in the slab's u-extent:   cu' = fma(du, su, fma(iu, h, uBase0)),  su = h / 256, uBase0 = g0u + su / 2
//   bytes 1-2   v16  = iv * 256 + dv, position in v:     cv' = fma(v16, su, vBase),              vBase = g0v + su / 2
//   bits 24-27  kw   position on the layer's thin axis:  cw' = fma(kw, wstep, wBase)
//   bits 28-31  kr   radius class, rounded UP:           R'  = fma(kr, rstep, rstep)
// (u, v, w) are the scene's coordinates permuted so that u, v span the grid; the test runs in that order (dot products do not
// care).  The host evaluates the same four fmas (fmaf: the same bits) and picks kr so that R' >= r_i + |C' - c_i|: the sphere
// (C', R') CONTAINS sphere i, its centre is off by s_i <= s_max.  Conservativeness, as for every bound: a positive
// reference discriminant puts the line within sqrt(r_i^2 + E/a) of c_i, hence within s_i + that of C'; against R' >= s_i + r_i the
// true bound discriminant exceeds -(E + 2 s_i sqrt(a E)) >= -(2E + a s_i^2) (AM-GM with weight 1: s_i is tiny here, where the
// group bounds' 101 E + 0.01 a s^2 paid for large offsets).  2E <= 32 eps a G plus the filter's own arithmetic (<= 30 eps a G) is
// inside K = 64, the one-sphere margin; a s_max^2 goes into Rf^2:
//   Rf^2 = R'^2 (1 + 1e-5) + s_max^2 + 64 eps (2 (|C'| + R')^2 + R'^2),  with 2 (|C'| + R')^2 + R'^2 <= 4 |C'|^2 + 5 R'^2
//   W'   = |C'|^2 - Rf^2 >= |C'|^2 (1 - 264 eps) - R'^2 (1 + 1e-5 + 320 eps) - s_max^2     (8 eps |C'|^2: the rounding of W' itself)
struct GridQuant {
    const uint32_t* rec;  // LDS
    float su, uBase0, h, vBase, wBase, wstep, rstep, s2;
};
RT_DEV float4 grid_quant_bound(uint32_t rec, float uBase, const GridQuant& Q) {
    const float fu = (float)(rec & 255u), fv = (float)((rec >> 8) & 0xffffu), fw = (float)((rec >> 24) & 15u), fr = (float)(rec >> 28);
    const float cu = __builtin_fmaf(fu, Q.su, uBase), cv = __builtin_fmaf(fv, Q.su, Q.vBase), cw = __builtin_fmaf(fw, Q.wstep, Q.wBase);
    const float R = __builtin_fmaf(fr, Q.rstep, Q.rstep);
    const float c2 = __builtin_fmaf(cw, cw, __builtin_fmaf(cv, cv, cu * cu));
    const float w = __builtin_fmaf(c2, 1.f - 264.f * 5.9604645e-8f, __builtin_fmaf(R * R, -(1.f + 1e-5f + 320.f * 5.9604645e-8f), -Q.s2));
    return make_float4(cu, cv, cw, w);
}
#ifndef RT_GRID_STEP
#define RT_GRID_STEP 4
#endif
constexpr uint32_t kGridStep = RT_GRID_STEP;  // one-sphere bounds a lane tests per step of the grid scan (4 or 8): eight halves the dependent L2 round trips of 5-8-sphere runs
static_assert(kGridStep == 4u || kGridStep == 8u, "grid step");
constexpr float kGridSlack = 1e-3f;  // cells: >= 40 x the rounding of a grid coordinate (|coordinate| <= 256 cells, 2^-24 relative)

// The v-range (rows) of slab iu for a segment S -> E in grid coordinates, widened by D: rows [r0, r1] clamped to the grid, or
// r0 > r1 when the slab is not within D of the segment.  Also the segment parameter s in [0, 1] at which the walk enters the slab.
// slope = dv/du and invAbsDu = 1/|du| of the segment, once per ray (grid_segment_slope); invAbsDu == 0: the segment is (nearly)
// parallel to v.
RT_DEV void grid_segment_slope(float su, float sv, float eu, float ev, float& slope, float& invAbsDu) {
    const float du = eu - su, dv = ev - sv;
    const bool steep = __builtin_fabsf(du) < 1e-4f * __builtin_fmaxf(__builtin_fabsf(dv), 1.f);
    slope = steep ? 0.f : dv / du;
    invAbsDu = steep ? 0.f : 1.f / __builtin_fabsf(du);
}
RT_DEV void grid_slab_rows(float su, float sv, float eu, float ev, float D, float slope, float invAbsDu, int iu, int nv, int& r0, int& r1,
                           float& sEnter) {
    const float ulo = __builtin_fminf(su, eu), uhi = __builtin_fmaxf(su, eu);
    // the part of the segment whose u lies within D of the slab [iu, iu + 1]
    const float a = __builtin_fmaxf(ulo, (float)iu - D - kGridSlack), b = __builtin_fminf(uhi, (float)(iu + 1) + D + kGridSlack);
    float vlo, vhi;
    if (invAbsDu == 0.f) {  // (nearly) parallel to v: the whole v-extent of the segment
        vlo = __builtin_fminf(sv, ev);
        vhi = __builtin_fmaxf(sv, ev);
        sEnter = 0.f;
    } else {
        const float va = __builtin_fmaf(a - su, slope, sv), vb = __builtin_fmaf(b - su, slope, sv);
        const float pad = 1e-4f * __builtin_fabsf(slope);  // (u - su) is good to 4e-5 cells; times the slope
        vlo = __builtin_fminf(va, vb) - pad;
        vhi = __builtin_fmaxf(va, vb) + pad;
        const float uNear = eu > su ? a : b;  // where the walk direction enters the slab's neighbourhood ...
        // ... moved a slack towards the start; as a fraction of the segment (rounded down by more than the product's rounding)
        sEnter = __builtin_fminf(__builtin_fmaxf((__builtin_fabsf(uNear - su) - kGridSlack) * invAbsDu * (1.f - 0x1p-20f), 0.f), 1.f);
    }
    if (a > b) {  // the slab is farther than D from the segment
        r0 = 1;
        r1 = 0;
        return;
    }
    const float f0 = __builtin_floorf(vlo - D - kGridSlack), f1 = __builtin_floorf(vhi + D + kGridSlack);
    r0 = f0 < 0.f ? 0 : (int)f0;
    r1 = f1 > (float)(nv - 1) ? nv - 1 : (int)f1;
    if (f1 < 0.f) r1 = -1;
}

// kQ: the step loop tests the QUANTISED bounds from LDS (GridQuant above) instead of the float4 bounds `leaf` from global memory.
template <bool kQ = false>
RT_DEV void scan_list_grid(const float4* __restrict__ tab, const float4* __restrict__ leaf, const uint32_t* __restrict__ orig,
                           const GridParams G, const uint16_t* __restrict__ cellStart, uint32_t nAlways, const float* treeBox, float boundNorm,
                           V3 o, V3 d, bool live, float& tmin, int& idx, uint16_t* waveCand, uint32_t lane, const GridQuant Q = GridQuant{}) {
    RT_SITE(G_SCAN);
    const float a = dot3(d, d);
    tmin = __builtin_inff();
    idx = -1;
    const float dO = dot3(d, o);
    const float m2a = -2.f * a;
    const float gx = m2a * o.x, gy = m2a * o.y, gz = m2a * o.z;
    // kQ: the quantised bounds live in (u, v, w) = the scene's axes permuted so that u, v span the grid; the owner lane hands out
    // its ray's g and d in that order (the products of a dot product commute; the filter's rounding is inside every margin)
    const uint32_t axW = 3u - G.axU - G.axV;
    const float guQ = G.axU == 0u ? gx : (G.axU == 1u ? gy : gz), gvQ = G.axV == 0u ? gx : (G.axV == 1u ? gy : gz), gwQ = axW == 0u ? gx : (axW == 1u ? gy : gz);
    const float duQ = G.axU == 0u ? d.x : (G.axU == 1u ? d.y : d.z), dvQ = G.axV == 0u ? d.x : (G.axV == 1u ? d.y : d.z), dwQ = axW == 0u ? d.x : (axW == 1u ? d.y : d.z);
    const float oo = dot3(o, o);
    const float aoo = a * oo;
    const float bt = 1e-4f * filter_sqrt_up(a) * (filter_sqrt_up(oo) + boundNorm);
    const float crLeaf = aoo * (1.f - 2.f * kMarginKLeaf * 5.9604645e-8f);
    uint32_t* work = reinterpret_cast<uint32_t*>(waveCand);               // kGridWork items: ray << 8 | slab
    uint32_t* exact = work + kGridWork;                                   // kGridExact entries: ray << 16 | scan entry
    unsigned long long* best = reinterpret_cast<unsigned long long*>(exact + kGridExact);
    best[lane] = ~0ull;
    wave_lds_handoff();
    uint32_t nWork = 0, nExact = 0;
    // the ray clipped to the padded box of the grid's spheres (the derivation is the hierarchy scan's, above)
    float su = 0.f, sv = 0.f, eu = 0.f, ev = 0.f, D = 0.f, tn = 0.f, tf = __builtin_inff(), gSlope = 0.f, gInvDu = 0.f;
    int slab = 0, slabLast = 0, slabStep = 1;  // the ray's next slab, its last one, the walk's direction (front to back)
    bool pending = false;
    {
        const float oc[3] = {o.x, o.y, o.z}, dc[3] = {d.x, d.y, d.z};
        const float X = 32.f * 5.9604645e-8f * __builtin_fmaf(2.f, oo, treeBox[7]);
        const float reach = __builtin_fminf(filter_sqrt_up(X), X * treeBox[8]);
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const float pad = reach + 1e-6f * (__builtin_fabsf(oc[ax]) + treeBox[6]);
            const float inv = __builtin_amdgcn_rcpf(dc[ax]);
            const float t0 = ((treeBox[ax] - pad) - oc[ax]) * inv, t1 = ((treeBox[3 + ax] + pad) - oc[ax]) * inv;
            tn = __builtin_fmaxf(tn, __builtin_fminf(t0, t1));
            tf = __builtin_fminf(tf, __builtin_fmaxf(t0, t1));
        }
        tn *= 1.f - 0x1p-10f;
        tf *= 1.f + 0x1p-10f;
        const bool hitsBox = live && !(tn > tf || !(tf > 0.f)) && tf < 3.0e38f;
        if (hitsBox) {
            const float ou = G.axU == 0u ? o.x : (G.axU == 1u ? o.y : o.z), ov = G.axV == 0u ? o.x : (G.axV == 1u ? o.y : o.z);
            const float du = G.axU == 0u ? d.x : (G.axU == 1u ? d.y : d.z), dv = G.axV == 0u ? d.x : (G.axV == 1u ? d.y : d.z);
            su = (__builtin_fmaf(tn, du, ou) - G.g0u) * G.invH;
            sv = (__builtin_fmaf(tn, dv, ov) - G.g0v) * G.invH;
            eu = (__builtin_fmaf(tf, du, ou) - G.g0u) * G.invH;
            ev = (__builtin_fmaf(tf, dv, ov) - G.g0v) * G.invH;
            // points are good to ~1e-5 scene units x 1/h here (|t d| <= the scene's size); the slack of the walk covers it
            D = G.rmaxOverH + reach * G.invH + kGridSlack;
            grid_segment_slope(su, sv, eu, ev, gSlope, gInvDu);
            const float fa = __builtin_floorf(__builtin_fminf(su, eu) - D - kGridSlack), fb = __builtin_floorf(__builtin_fmaxf(su, eu) + D + kGridSlack);
            const int iuA = fa < 0.f ? 0 : (int)fa, iuB = fb > (float)(G.nu - 1u) ? (int)G.nu - 1 : (int)fb;
            if (iuA <= iuB && fb >= 0.f) {
                const bool fwd = eu >= su;
                slab = fwd ? iuA : iuB;
                slabLast = fwd ? iuB : iuA;
                slabStep = fwd ? 1 : -1;
                pending = true;
            }
        }
    }
    auto drainExact = [&]() {
        RT_SITE(G_DRAIN);
        wave_lds_handoff();  // exact-list entries written by other lanes
        for (uint32_t base = 0; base < nExact; base += kWaveSize) {
            RT_SITE(G_BSTEP);
            const uint32_t k = base + lane;
            const bool has = k < nExact;
            const uint32_t ent = has ? exact[k] : 0u;
            const uint32_t r = ent >> 16, cand = ent & 0xffffu;
            const float rox = lane_fetch(r, o.x), roy = lane_fetch(r, o.y), roz = lane_fetch(r, o.z);
            const float rdx = lane_fetch(r, d.x), rdy = lane_fetch(r, d.y), rdz = lane_fetch(r, d.z);
            const float ra = lane_fetch(r, a);
            const float4 S = tab[cand];
            const uint32_t og = orig[cand];  // (requested WITH the sphere: inside the branch below it was a second dependent L2 read per accepted root)
            const float ocx = rox - S.x;
            const float ocy = roy - S.y;
            const float ocz = roz - S.z;
            const float b = (ocx * rdx + ocy * rdy) + ocz * rdz;
            const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - S.w;
            const float e = b * b - ra * cc;
            const float sq = sqrt_rn(e > 0.f ? e : 1.f);  // (a root is read only where e > 0: the other lanes keep the wave on the fast path)
            const float t = pooled_root(b, sq, ra);  // ray-tracing.cpp:56, :69
            if (has && e > 0.f && t > 0.001f && t < __builtin_inff()) {
                RT_SITE(G_BMIN);
                const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)((og << 16) | cand);
                __hip_atomic_fetch_min(best + r, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        nExact = 0;
        wave_lds_handoff();  // the exact list may be refilled from here on
    };
    // The big spheres (at most eight, entries 4q): the owner lane tests its own ray against the sphere's one-sphere bound --
    // the floor is hit by most rays, the others by few -- and the survivors take one exact slot each; their hits are the walk's
    // first far limits.
    const float btBig = 1e-4f * filter_sqrt_up(a) * (filter_sqrt_up(oo) + G.bigNorm);
#ifndef RT_BIG_BATCH
#define RT_BIG_BATCH 1  // 1: the big spheres' bounds four at a time, ONE prefix sum and one push loop per four (0: a ballot and a push per sphere)
#endif
#if RT_BIG_BATCH
    static_assert(kGridExact >= 4u * kWaveSize, "four big spheres per batch: 256 entries");
    for (uint32_t q0 = 0; q0 < nAlways; q0 += 4u) {
        uint32_t mb = 0u;  // bit k: the bound of big sphere q0 + k may hold a root of this lane's ray
        const uint32_t qe = nAlways - q0 < 4u ? nAlways - q0 : 4u;
        for (uint32_t k = 0; k < qe; ++k) {
            RT_SITE(G_BIG);
            const int rej = bound_rejected(leaf[4u * (q0 + k)], v3(gx, gy, gz), d, a, dO, crLeaf, btBig);
            mb |= (rej >= 0 ? 1u : 0u) << k;
        }
        if (!live) mb = 0u;
        const uint32_t nh = (uint32_t)__builtin_popcount(mb);
        const uint32_t incl = wave_inclusive_sum(nh);
        uint32_t* wp = exact + (incl - nh);  // (the list is empty here: every batch ends in a drain)
        while (mb != 0u) {
            RT_SITE(G_BIGPUSH);
            const uint32_t k = (uint32_t)__builtin_ctz(mb);
            mb &= mb - 1u;
            *wp++ = lane << 16 | (4u * (q0 + k));
        }
        nExact = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (nExact != 0u) drainExact();
    }
#else
    for (uint32_t q = 0; q < nAlways; ++q) {
        RT_SITE(G_BIG);
        if (nExact + (uint32_t)kWaveSize > kGridExact) drainExact();
        const bool cand = live && bound_rejected(leaf[4u * q], v3(gx, gy, gz), d, a, dO, crLeaf, btBig) >= 0;
        const uint64_t lm = __ballot(cand);
        if (cand) exact[nExact + prefix_count(lm)] = lane << 16 | (4u * q);
        nExact += (uint32_t)__popcll(lm);
    }
    if (nExact != 0u) drainExact();
#endif
    const int nv = (int)G.nv;
    constexpr uint32_t kFeed = RT_GRID_FEED;  // slabs a ray lists per feed pass (a pass adds at most 64 * kFeed items)
    static_assert(kGridWork >= 64u * kFeed, "one feed pass must fit the work list");
    for (;;) {
        // feed: every ray with slabs left lists its next (up to) kFeed of them, front to back -- (ray, slab) is all an item says;
        // a ray whose next slab begins beyond its closest hit so far is done
        while (nWork + 64u * kFeed <= kGridWork && __ballot(pending) != 0ull) {
            RT_SITE(G_FEED);
            uint32_t cnt = 0;
            if (pending) {
                RT_SITE(G_FEED_LANE);
                const uint32_t left = (uint32_t)((slabLast - slab) * slabStep) + 1u;
                cnt = left < kFeed ? left : kFeed;
#ifndef RT_GRID_FEEDSKIP
#define RT_GRID_FEEDSKIP 1
#endif
                const uint32_t tb = RT_GRID_FEEDSKIP ? reinterpret_cast<const uint32_t*>(best + lane)[1] : 0x7f800000u;
                if (tb < 0x7f800000u) {
                    // where the walk enters the neighbourhood of slab `slab` (grid_slab_rows' sEnter), against the far limit
                    const float uNear = eu > su ? __builtin_fmaxf(__builtin_fminf(su, eu), (float)slab - D - kGridSlack)
                                                : __builtin_fminf(__builtin_fmaxf(su, eu), (float)(slab + 1) + D + kGridSlack);
                    const float sEnter = __builtin_fminf(__builtin_fmaxf((__builtin_fabsf(uNear - su) - kGridSlack) * gInvDu * (1.f - 0x1p-20f), 0.f), 1.f);
                    if (__builtin_fmaf(sEnter, tf - tn, tn) > __uint_as_float(tb) * (1.f + 0x1p-10f)) cnt = 0;  // every later slab starts farther still
                }
                if (cnt == 0u) pending = false;
            }
            const uint32_t incl = wave_inclusive_sum(cnt);
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            uint32_t* wp = work + nWork + (incl - cnt);
            for (uint32_t k = 0; k < cnt; ++k) wp[k] = lane << 8 | (uint32_t)(slab + (int)k * slabStep);
            nWork += tot;
            if (cnt != 0u) {
                if ((uint32_t)((slabLast - slab) * slabStep) + 1u == cnt) pending = false;
                slab += (int)cnt * slabStep;
            }
        }
        if (nWork == 0u) break;
        wave_lds_handoff();  // items listed by other lanes
        // consume the whole list in the order it was fed: 64 (ray, slab) items per round
        for (uint32_t base = 0; base < nWork; base += kWaveSize) {
            RT_SITE(G_ROUND);
            const bool has = base + lane < nWork;
            const uint32_t ent = has ? work[base + lane] : 0u;
            const uint32_t r = ent >> 8;
            const int iu = (int)(ent & 255u);
            const V3 fg = kQ ? v3(lane_fetch(r, guQ), lane_fetch(r, gvQ), lane_fetch(r, gwQ)) : v3(lane_fetch(r, gx), lane_fetch(r, gy), lane_fetch(r, gz));
            const V3 fd = kQ ? v3(lane_fetch(r, duQ), lane_fetch(r, dvQ), lane_fetch(r, dwQ)) : v3(lane_fetch(r, d.x), lane_fetch(r, d.y), lane_fetch(r, d.z));
            const float fa = lane_fetch(r, a), fdO = lane_fetch(r, dO), fcr = lane_fetch(r, crLeaf), fbt = lane_fetch(r, bt);
            const float uBaseQ = kQ ? __builtin_fmaf((float)iu, Q.h, Q.uBase0) : 0.f;  // the slab's u origin (GridQuant)
            const float fsu = lane_fetch(r, su), fsv = lane_fetch(r, sv), feu = lane_fetch(r, eu), fev = lane_fetch(r, ev), fD = lane_fetch(r, D);
            const float fsl = lane_fetch(r, gSlope), fiv = lane_fetch(r, gInvDu);
            const float ftn = lane_fetch(r, tn), ftf = lane_fetch(r, tf);
            const float fun = ftn > 0.f ? fa * ftn : -__builtin_inff(), fuf = fa * ftf;  // the owner's boxUn, boxUf: the same products
            int r0, r1;
            float sEnter;
            grid_slab_rows(fsu, fsv, feu, fev, fD, fsl, fiv, iu, nv, r0, r1, sEnter);
            uint32_t eb = 0u, ee = 0u;
            if (has && r0 <= r1) {
                // the slab's cells r0..r1 are consecutive, and so are their spheres in the scan table
                eb = (uint32_t)cellStart[iu * nv + r0];
                ee = (uint32_t)cellStart[iu * nv + r1 + 1];
#ifndef RT_GRID_ROUNDSKIP
#define RT_GRID_ROUNDSKIP 0  // 1: a consumer re-checks its slab against the ray's closest hit so far before testing it (round 3; -1.5 % on grid10k: the feed's own check is enough)
#endif
#if RT_GRID_ROUNDSKIP
                // (a slab the ray enters beyond its closest hit so far has nothing to add)
                const uint32_t tb0 = reinterpret_cast<const uint32_t*>(best + r)[1];
                if (tb0 < 0x7f800000u && __builtin_fmaf(sEnter, ftf - ftn, ftn) > __uint_as_float(tb0) * (1.f + 0x1p-10f)) ee = eb;
#endif
            }
            // four spheres per step; every lane runs as many steps as the longest run of the round needs
            while (__ballot(eb < ee) != 0ull) {
                RT_SITE(G_STEP);
#ifndef RT_GRID_SPAN
#define RT_GRID_SPAN 0  // limits in the step loop's bound test: 0 none (default: the limits cost more than the exact tests they save, +5.2 % on grid10k), 1 the far limit (the closest hit so far), 2 near and far
#endif
#if RT_GRID_SPAN
                const uint32_t tbits = reinterpret_cast<const uint32_t*>(best + r)[1];  // far limit: the ray's closest hit so far
                float fu = tbits < 0x7f800000u ? (fa * __uint_as_float(tbits)) * (1.f + 0x1p-10f) : __builtin_inff();
                fu = __builtin_fminf(fu, fuf);
#endif
                uint32_t rb = 0u;
#pragma unroll
                for (uint32_t q = 0; q < kGridStep; ++q) {
                    const uint32_t e = eb + q;
                    // (entry 0 is always there; its result is masked)
                    const float4 B = kQ ? grid_quant_bound(Q.rec[e < ee ? e : 0u], uBaseQ, Q) : leaf[e < ee ? e : 0u];
#if RT_GRID_SPAN == 2
                    const int rej = e < ee ? bound_rejected_span(B, fg, fd, fa, fdO, fcr, fbt, fun, fu) : -1;
#elif RT_GRID_SPAN == 1
                    const int rej = e < ee ? bound_rejected_far(B, fg, fd, fa, fdO, fcr, fbt, fu) : -1;
#elif !defined(RT_GRID_BEHIND)
                    // b~^2 - t alone, without the "behind the origin" half of bound_rejected (a min and the fetch of its threshold): the
                    // walk lists few cells behind the origin, and the exact test rejects their spheres anyway (+0.3 % on grid10k)
                    const float bb = __builtin_fmaf(-fd.z, B.z, __builtin_fmaf(-fd.y, B.y, __builtin_fmaf(-fd.x, B.x, fdO)));
                    const float tt = __builtin_fmaf(fg.z, B.z, __builtin_fmaf(fg.y, B.y, __builtin_fmaf(fg.x, B.x, __builtin_fmaf(fa, B.w, fcr))));
                    const int rej = e < ee ? __float_as_int(__builtin_fmaf(bb, bb, -tt)) : -1;
#else
                    const int rej = e < ee ? bound_rejected(B, fg, fd, fa, fdO, fcr, fbt) : -1;
#endif
                    rb = __builtin_amdgcn_alignbit(rb, (uint32_t)rej, 31);
                }
                const uint32_t m = ~rb & ((1u << kGridStep) - 1u);  // bit kGridStep-1-q = entry eb + q
                if (nExact >= kGridDrain) drainExact();  // early and often: every exact round may pull the far limits in
                // one prefix sum over the lanes' survivor counts, then every lane appends its own entries
                auto pushSurvivors = [&](uint32_t mask) __attribute__((always_inline)) {
                    const uint32_t nh = (uint32_t)__builtin_popcount(mask);
                    const uint32_t incl = wave_inclusive_sum(nh);
                    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    if (kGridStep > 4u && nExact + tot > kGridExact) drainExact();  // (four per step: the drain above keeps a step's room)
                    uint32_t* wp = exact + nExact + (incl - nh);
                    uint32_t mm = mask;
                    while (mm != 0u) {
                        RT_SITE(G_PUSH);
                        const uint32_t bit = 31u - (uint32_t)__builtin_clz(mm);
                        mm &= ~(1u << bit);
                        *wp++ = r << 16 | (eb + (kGridStep - 1u - bit));
                    }
                    nExact += tot;
                };
                if (kGridStep > 4u && (uint32_t)__popcll(__ballot(m != 0u)) * kGridStep > kGridExact) {
                    // (more survivors than the list holds are possible in principle: eight per lane; then the step is pushed in halves)
                    pushSurvivors(m & 0xf0u);
                    pushSurvivors(m & 0x0fu);
                } else {
                    pushSurvivors(m);
                }
                eb += kGridStep;
            }
        }
        nWork = 0;
        wave_lds_handoff();  // the list is free for the next feed
    }
    drainExact();
    const unsigned long long mineKey = best[lane];  // behind drainExact's closing hand-off
    const uint32_t tb = (uint32_t)(mineKey >> 32);
    if (tb < 0x7f800000u) {
        tmin = __uint_as_float(tb);
        idx = (int)(mineKey & 0xffffull);
    }
}

}  // namespace rtd

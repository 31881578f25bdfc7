// rt_params.h — launch parameters, LDS-resident scene constants, row-set arithmetic and primary-ray generation of the
// render-loop kernels (see rt_kernels.h for the map from reference functions to kernels).
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

#ifdef RT_TIMELINE
static __device__ unsigned long long g_tl[16];
static __device__ unsigned long long g_tlRing[4096 * 16];  // per wave: the last 16 (claim time << 24 | block index / 64) records
static __device__ unsigned int g_tlLast[4096 * 4];  // per wave, the last path to finish: start time (low word), depth, slot, traversals
static __device__ unsigned long long g_tlWave[4096 * 8];  // per wave: start, drain start, exit, blocks, last block, last claim, iterations, drain iterations
static __device__ unsigned int g_tlHist[1024];  // waves leaving the loop per 25-us bucket of their own lifetime  // diagnostic build only: wall-clock landmarks of the last trace kernel
#endif

constexpr int kWaveSize = 64;
constexpr uint32_t kMaxLevels = 6;  // levels of group bounds (4-ary): 128 * 4^5 groups at most
constexpr uint32_t kQueueBlock = 128;  // paths a wave takes from the global queue per claim (claimed one block ahead: rt_kernels.h)
constexpr uint32_t kCarryQueueBlock = 128; // ... in the frame-pipelining kernel (rt_kernels.h): two batches of the path cache

// ---- frame pipelining (progressive use: SpheresApp::OnRender adds ONE sample per frame, spheres-app.cpp:163-184).
// A 1-spp frame is 0.13 ms of work followed by the tail of its few 51-segment paths; with pipelining the trace kernel of
// frame j ends as soon as the fresh queue is empty and CARRIES its unfinished paths (80-byte entries) into the kernel of
// frame j+1.  Each rt_render call owns one region of a ring of sample buffers; regions are added to the HDR strip strictly in
// call order, and only once no carried path belongs to them (FrameCtl::oldest_open).
// The queue of fresh paths (every trace kernel): a wave owns TraceParams::static_blocks blocks statically (block index =
// wave index: no claim, and no rush of 4,096 simultaneous atomics when the kernel starts); the rest of the launch's paths
// sits in eight INTERLEAVED shards -- shard k owns the blocks k, k + 8, k + 16, ... of the work order, so all shards move
// through the launch's work order together and the tiles sorted to its end (the sky) really are what the launch ends with;
// a workgroup prefers shard blockIdx & 7 (workgroups are dealt round-robin over the 8 XCDs) and steals from the others when
// its own is dry.  The cursors are 128 bytes apart, and a wave LOOKS at a cursor before it claims: same-line atomics are
// served one behind the other (~11 ns each), loads are not.
constexpr uint32_t kQueueShards = 8;
#ifndef RT_SHARD_STRIDE_WORDS
#define RT_SHARD_STRIDE_WORDS 32
#endif
constexpr uint32_t kShardStrideWords = RT_SHARD_STRIDE_WORDS;  // distance of the cursors in 32-bit words

// n / d for a divisor fixed per launch, without the ~28-operation expansion of a 32-bit division (Granlund & Montgomery,
// "Division by invariant integers using multiplication", unsigned case): with l = ceil(log2 d) and
// m = floor(2^32 (2^l - d) / d) + 1, t = mulhi(m, n) and q = (t + ((n - t) >> min(l, 1))) >> max(l - 1, 0) for every
// 32-bit n.  Five operations; a prepared path needs three such quotients (tile, row, row block).
struct FastDiv {
    uint32_t m, sh;  // sh = sh1 | sh2 << 8
};
RT_DEV FastDiv make_fastdiv(uint32_t d) {
    uint32_t l = 0;
    while (l < 32u && (1ull << l) < (unsigned long long)d) ++l;
    FastDiv f;
    f.m = (uint32_t)((((1ull << l) - (unsigned long long)d) << 32) / (unsigned long long)d + 1ull);
    f.sh = (l < 1u ? l : 1u) | ((l > 0u ? l - 1u : 0u) << 8);
    return f;
}
RT_DEV uint32_t fastdiv(uint32_t n, FastDiv f) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t t = __umulhi(f.m, n);
#else
    const uint32_t t = (uint32_t)(((unsigned long long)f.m * (unsigned long long)n) >> 32);
#endif
    return (t + ((n - t) >> (f.sh & 255u))) >> (f.sh >> 8);
}

struct ContEntry {
    float4 a, b, c, d, e;  // ro.xyz rd.x | rd.yz thr.xy | thr.z rad.xyz | xoshiro s0..s3 | slot, depth, region sequence & 255, traversals
    float4 f, g;           // a lane waiting for its shadow scan (far hit point): pend.xyz nextDir.x | nextDir.yz, state and flags, -
};
// Carried paths stay with their wave: wave w of the (always full) grid writes its <= 64 entries to slots [64 w, 64 w + n) of
// the outgoing buffer and n to the outgoing count array, and wave w of the next kernel restores exactly those -- no cursor,
// no atomics.  (A first version claimed entries, fresh blocks and output slots with atomics on one control block: ~27,000
// same-line atomics per 1-spp frame, which one L2 channel serialises at ~11 ns each -- 0.3 ms per frame.)
struct FrameCtl {            // device control block of one context's pipeline
    uint32_t oldest_open;    // min region sequence among the paths carried out of the last trace kernel (0xffffffff: none)
    // The commit point, twice: commit kernel number k reads entry k & 1 and its first block writes entry (k + 1) & 1, so no
    // block can see the new value early and nothing has to wait for "the last block" (a closing ticket was one same-address
    // atomic per block, which held the kernel to a few hundred blocks; it now runs one pixel per thread).
    uint32_t committed_seq[2];      // regions with sequence <= this are in the HDR strip
    uint32_t committed_samples[2];  // samples per pixel in the HDR strip
};
constexpr uint32_t kMaxFramesInFlight = 16;  // regions of the sample ring
// The shadow index's GLOBAL list (spheres whose footprint covers much of the grid: the floor, the big spheres; <= 64) is walked by
// every query: its spheres and ids sit in LDS in every variant -- 18 bytes each -- so that the walk reads them directly instead of
// chasing id -> sphere through the tables (two dependent L2 reads per pair of entries in the large-scene variants).
RT_DEV uint32_t sg_glob_slots(uint32_t nGlobal) { return nGlobal ? nGlobal + (nGlobal * 2u + 15u) / 16u : 0u; }

// Lights 1 .. n_lights-1 of the scene's list (light 0 lives in TraceParams / SceneConsts as before): direction, radiance and
// the light's own shadow index, in global memory.  Member names equal SceneConsts' so that shade_value / shadow_query take either.
struct LightRec {
    float sun_dir[3], sun_rad[3];
    float cam_o[3];
    float sg_e1[3], sg_e2[3], sg_u0, sg_v0, sg_inv_cell, sg_p0sq;
    uint32_t sg_nx, sg_ny, sg_nglobal, sg_enabled;
    const uint16_t *cell_start, *entries, *global;
};

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
    // ... and the same table PACKED into 16 bytes per entry (rt_shade.h load_material16) when every material of the scene can be
    // (colours are 8-bit by construction: XMLoadColor of an XMCOLOR): one 16-byte read per hit instead of three, and small enough
    // for LDS next to a full hit stash.  mats16_mode: 0 = not used, 1 = read from global memory, 2 = staged into LDS (kMatsL2 variants)
    const uint4* mats16;
    uint32_t mats16_mode;
    uint32_t n;                // real spheres
    uint32_t n_groups;         // groups of four entries (even)
    uint32_t mats_in_lds;      // stage the material table into LDS (else it is read through L2)
    float bound_norm;          // max over groups of |C| + R (scale of the filter's behind-the-origin threshold)
    unsigned long long single_mask[2];  // flat scan: groups of ONE sphere, as bits of the two candidate-bitmap halves (rt_scan.h)
    float tree_box[9];         // hierarchy scan: lo.xyz, hi.xyz of the box around every sphere IN the hierarchy (radii included), max |coordinate|,
                               // [7] = 3 A^2 (A = max |c| + r of those spheres), [8] = 1 / (2 r_min): the per-ray padding's constants (rt_scan.h)
    uint32_t tree_box_on;      // ... valid: rays are clipped to it (near and far limits of the descent, rt_scan.h)
    uint32_t n_always;         // hierarchy scan: the first n_always groups hold one big sphere each and are NOT in the hierarchy:
                               // every live ray tests them exactly (a floor inside the bounds makes every node above it a candidate)
    uint32_t n_padded;         // 4 * n_groups + 4
    // Cell-grid scan (rt_scan.h scan_list_grid; large flat scenes): the small spheres are stored sorted by home cell behind the big ones
    const uint16_t* grid_cell_start;  // [grid_nu * grid_nv + 1] first scan entry of each cell, or null: no grid for this scene
    uint32_t grid_nu, grid_nv, grid_ax_u, grid_ax_v, grid_in_lds;
    float grid_g0u, grid_g0v, grid_inv_h, grid_rmax_over_h, grid_big_norm;
    // ... and its QUANTISED one-sphere bounds (rt_scan.h GridQuant): 4 bytes per scan entry, staged into LDS by the kGridQ kernels, so that the
    // step loop's conservative tests read no global memory; null: the scene has none (the float4 bounds `leaf` are read through L1/L2)
    const uint32_t* grid_qrec;
    float grid_q[8];                  // su, uBase0, h, vBase, wBase, wstep, rstep, s_max^2 (GridQuant's constants)
    // Exact shadow index for the (single, directional) sun: spheres binned by their footprint in the plane
    // perpendicular to the light.  Valid for hit points with |p|^2 <= sg_p0sq.
    const uint16_t* sg_cell_start;  // [sg_nx * sg_ny + 1]
    const uint16_t* sg_entries;     // clustered entry indices per cell
    const uint16_t* sg_global;      // entries tested for every query (footprints covering much of the grid)
    const float4* sg_sph;           // large scenes (index in global memory): the scan record of every sg_entries element, so that a walk
                                    // requests ids and spheres together instead of id -> sphere (two dependent L2 reads); else null
    uint32_t sg_nx, sg_ny, sg_nglobal, sg_nentries, sg_enabled, sg_in_lds;
    uint32_t sg_glob16;             // float4 slots of the global list's LDS copy (spheres, then ids) in front of the tables; 0: none (unit kernels)
    float sg_e1[3], sg_e2[3], sg_u0, sg_v0, sg_inv_cell, sg_p0sq;
    float cam_o[3], cam_x[3], cam_y[3], cam_oip[3];
    float aperture, focal;
    float sun_dir[3], sun_rad[3];  // light 0: sun_rad = luminance * colour (light.cpp:27, left factor)
    uint32_t n_lights;             // lights in the scene's list (spheres-app.h:38); 1 = the reference's scene
    const LightRec* extra_lights;  // [n_lights - 1] lights 1 .. (global memory)
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
    uint32_t sampler;     // RT_SAMPLER_* flags (rt_api.h): 0 = the reference's mappings
    uint64_t seed;
    const uint32_t* path_list;  // optional explicit (i, j, s) triples (unit tests)
    const uint32_t* tile_order; // optional: full tile of WORK position w is image tile tile_order[w] (expensive tiles first)
    const float2* jitter_tab;   // [spp_pass] Halton2D(s;2,3) for s = s0.. (rt_raygen_tables_kernel), or null
    const float2* lens_tab;     // [.] HaltonSampleDisk(k;4,5) for k = lens_k0.., or null
    uint32_t lens_k0;
    float* samples;             // [total_paths][3] radiance * exposure
    uint32_t* trav_out;         // optional per-path traversal counts
    // frame pipelining (kCarry kernels only; all zero otherwise)
    FrameCtl* ctl;
    const ContEntry* cont_in;   // [waves][64] entries carried out of the previous trace kernel ...
    const uint32_t* cont_in_n;  // ... and how many of them each wave left
    ContEntry* cont_out;        // where this kernel carries its unfinished paths
    uint32_t* cont_out_n;
    uint32_t carry;             // 1: end when the fresh queue is empty and carry the live paths; 0: run every path to its end
    uint32_t region_seq;        // sequence number of this call's region
    uint32_t max_carry_age;     // a path whose region is this many calls old (or older) is finished in place, not carried
    uint32_t min_iters;         // iterations a wave runs before it may carry its paths out
    uint32_t queue_block;       // paths per queue block in the carrying kernel (a multiple of 64)
    uint32_t static_blocks;     // blocks every wave owns without asking (block index = wave index)
    uint32_t dyn_begin, dyn_blocks;  // the shards' part of the work order: first path and number of blocks
    uint32_t sample_base;       // first 12-byte slot of this call's region in the sample ring (0 without pipelining)
    FastDiv fd_tile, fd_w, fd_rows;  // division by 64 * spp_pass, W and rs.block_rows (set per launch)
    uint32_t ray_cache_off16;  // float4 offset of the per-wave prepared-path caches (or hit stashes) in dynamic LDS, 0 = none
    uint32_t ray_cache_stride16;  // ... and the size of one wave's region in float4
    uint32_t stash_cap;        // kStash variants (rt_kernels.h): records a wave's hit stash holds (<= 63)
    uint32_t stash_process;    // ... hits are processed once MORE than this many lanes hold one (<= stash_cap: what is not
                               // processed must fit the stash)
    uint32_t* shard_heads;      // [kQueueShards][kShardStrideWords]: [k][0] = cursor of queue shard k (blocks claimed from it; zeroed before launch)
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
    uint32_t sampler;
    const float2* jitter_tab;
    const float2* lens_tab;
    uint32_t exit_age_max;  // frame pipelining: 1 + the largest region age among the paths this workgroup carried out (0: none)
    uint32_t exit_ticket;   // waves of this workgroup that have left the loop
    uint32_t dry_mask;      // bit k: some wave of this workgroup has seen queue shard k empty (nobody asks memory again)
};
constexpr uint32_t kSceneConstBytes = 256;                         // SceneConsts at the start of the dynamic LDS image ...
constexpr uint32_t kConstBytes = kSceneConstBytes + 8 * 256;       // ... followed by the elementary functions' tables (255 words)
static_assert(sizeof(SceneConsts) <= kSceneConstBytes, "SceneConsts must fit its LDS slot");
static_assert(kMathTabWords <= 256, "math tables must fit their LDS slot");
RT_DEV void fill_consts(const TraceParams& p, SceneConsts& k) {
    for (int i = 0; i < 3; ++i) {
        k.cam_o[i] = p.cam_o[i]; k.cam_x[i] = p.cam_x[i]; k.cam_y[i] = p.cam_y[i]; k.cam_oip[i] = p.cam_oip[i];
        k.sun_dir[i] = p.sun_dir[i]; k.sun_rad[i] = p.sun_rad[i]; k.sky_emit[i] = p.sky_emit[i];
        k.sg_e1[i] = p.sg_e1[i]; k.sg_e2[i] = p.sg_e2[i];
    }
    k.aperture = p.aperture; k.focal = p.focal; k.exposure = p.exposure;
    k.sg_u0 = p.sg_u0; k.sg_v0 = p.sg_v0; k.sg_inv_cell = p.sg_inv_cell; k.sg_p0sq = p.sg_p0sq;
    k.sg_nx = p.sg_nx; k.sg_ny = p.sg_ny; k.sg_nglobal = p.sg_nglobal; k.sg_enabled = p.sg_enabled;
    k.W = p.W; k.H = p.H; k.s0 = p.s0; k.lens_k0 = p.lens_k0; k.sampler = p.sampler;
    k.jitter_tab = p.jitter_tab; k.lens_tab = p.lens_tab;
    k.exit_age_max = 0u; k.exit_ticket = 0u; k.dry_mask = 0u;
}

// --------------------------------------------------------------------------- row sets
RT_DEV uint32_t rowset_global_row(const rt_rowset& rs, uint32_t lr, FastDiv byBlockRows) {
    const uint32_t lb = fastdiv(lr, byBlockRows);
    const uint32_t k = lr - lb * rs.block_rows;
    return rs.first_row + (lb * rs.nshards + rs.shard) * rs.block_rows + k;
}
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
RT_DEV void halton_disk_4_5(uint32_t k, float& lensx, float& lensy, uint32_t sampler = 0u) {  // quasi-random.cpp:52-61
    const float theta = (2.f * 3.141592654f) * halton(k, 4);
    float r = halton(k, 5);  // the reference does not take the square root: a centre-weighted disk
    if (sampler & RT_SAMPLER_SQRT_DISK) r = sqrt_rn(r);
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
        // the pointers may come out of LDS (SceneConsts): say that they point to global memory, or the loads become FLAT
        // ones, which wait for vmcnt(0) and lgkmcnt(0) -- i.e. for the wave's outstanding sample stores as well
        typedef const float __attribute__((address_space(1)))* GlobalF;
        const GlobalF jt = (GlobalF)(p.jitter_tab + (s - p.s0));
        const GlobalF lt = (GlobalF)(p.lens_tab + (li - p.lens_k0));
        jx = jt[0]; jy = jt[1]; lensx = lt[0]; lensy = lt[1];
    } else {
        RT_SITE(P_HALTON);
        jx = halton(s, 2);
        jy = halton(s, 3);
        halton_disk_4_5(li, lensx, lensy, p.sampler);
    }
    const float uvx = ((float)(int)i + jx) / xsize;
    const float uvy = ((float)(int)j + jy) / ysize;

    camera_get_ray(p, uvx, uvy, lensx, lensy, origin, dir);
}


}  // namespace rtd

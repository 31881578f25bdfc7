// rt_kernels.h — HIP kernels of the render-loop hot path for MI355X (gfx950 / CDNA4).
// Device functions live in rt_params.h (parameters, ray generation), rt_scan.h (filter + resolve), rt_shade.h (hit
// processing, shadow index) and rt_device_math.h (DirectXMath restatements, Halton, RNG, elementary functions).
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

#include "rt_params.h"
#include "rt_scan.h"
#include "rt_shade.h"

namespace rtd {

enum : uint32_t { kIdle = 0u, kNeedClosest = 1u, kNeedShadow = 2u, kHaveHit = 3u };  // kHaveHit: hit-stash variant only (below)

// Hit stash (kStash variants of rt_trace_kernel): one record per closest hit that waits for its hit processing --
// hit position, ray direction, throughput, radiance, stream state, sample slot, depth | scan entry << 16, traversals.
// Field k of record r sits at dword k * cap + r of the wave's region: consecutive records, consecutive banks.
constexpr uint32_t kStashDwords = 19;

// Work-item index -> (column i, global row j, sample s) and the path's slot in the sample buffer: the explicit path
// list of the unit tests, or the tiled order below over the rows of this shard.  slot = index either way.
RT_DEV void path_coordinates(const TraceParams& p, uint32_t q, uint32_t& i, uint32_t& j, uint32_t& s, uint32_t& slot) {
    if (p.path_list) {
        RT_SITE(K_PATHLIST);
        i = p.path_list[3 * q];
        j = p.path_list[3 * q + 1];
        s = p.path_list[3 * q + 2];
        slot = q;
    } else {
        // storage order = [tile of 64 local pixels][sample of the pass][pixel in tile] (the last tile is as wide as the
        // pixels left): a wave's block of 128 consecutive work-items is two sample planes of one tile, so its 12-byte
        // results fill whole cache lines, and rt_accumulate_kernel reads 768 contiguous bytes per plane.  The WORK order
        // is the same with the full tiles permuted by tile_order (expensive tiles first, sky last: rt_tile_order_kernel),
        // so that the 51-segment paths are not the last ones a launch starts; the partial tile stays last.
        const uint32_t tileSpan = 64u * p.spp_pass;
        const uint32_t nFull = p.npix_local >> 6;
        const uint32_t tileW = fastdiv(q, p.fd_tile);
        uint32_t pl, k;
        slot = q;
        if (tileW < nFull) {
            const uint32_t rem = q - tileW * tileSpan;
            const uint32_t tile = p.tile_order ? p.tile_order[tileW] : tileW;
            k = rem >> 6;
            pl = (tile << 6) + (rem & 63u);
            slot = tile * tileSpan + rem;
        } else {
            RT_SITE(K_PARTIAL);
            const uint32_t rem = q - nFull * tileSpan;
            const uint32_t wl = p.npix_local - (nFull << 6);
            k = rem / wl;
            pl = (nFull << 6) + (rem - k * wl);
        }
        s = p.s0 + k;
        const uint32_t lr = fastdiv(pl, p.fd_w);
        i = pl - lr * p.W;
        j = rowset_global_row(p.rs, lr, p.fd_rows);
        slot += p.sample_base;  // the call's region of the sample ring (0 without frame pipelining)
    }
}

// Scene tables as the scan and the hit processing read them: LDS copies when the launch staged them, else global (L2).
struct SceneTabs {
    const float4* scan;
    const uint32_t* orig;
    const float4* leaf;
    const float* rad;
    const rt_material* mats;
    const uint4* mats16;    // packed materials in LDS (kMatsL2 variants with p.mats16_mode == 2), else null
    const float* ops;       // filter operand image of the top level (matrix-core scans)
    const float4* tree;
    const uint16_t *sgCell, *sgEntries, *sgGlobal;
    const uint16_t* gridCells;  // cell-grid scan: first scan entry per cell
    const uint32_t* gridQ;      // ... and (kGridQ) the quantised one-sphere bounds in LDS
    uint32_t nTop, nTiles;
};

// Workgroup prologue shared by the megakernel and the closest-hit unit kernel: copy the scene tables the variant keeps in
// LDS to tabBase (16-byte aligned pieces: scan | one-sphere bounds (flat matrix-core scan only) | orig | materials (48 B =
// 3 float4) | radii | filter operands | shadow index; tree mode: operands | all levels of bounds) and build the filter's
// operand image.  Every thread of the workgroup must call it.
// kHitLds: flat matrix-core variant -- the material table and the shadow index are in LDS too; hierarchy variant -- all
// levels of bounds are in LDS.  Their pointers are assigned unconditionally so that the compiler can prove the address space -- a pointer that is LDS or global
// depending on a run-time flag becomes a FLAT load, and every flat load waits for vmcnt(0) AND lgkmcnt(0), i.e. also
// for the previous iteration's sample stores.  The host launches it only when p.mats_in_lds and (p.sg_in_lds or no index).
// kMatsL2 (with kHitLds): the material table is NOT staged -- it is read through L2 with a global-address-space pointer, and
// the 48 bytes per sphere it would take go to the hit stash (on the cover scene 44 -> 63 records per wave).
// kSgLds (cell-grid scan with its tables in global memory): the shadow index is staged behind the cells (typed LDS pointers).
template <bool kLds, int kScan, bool kHitLds = false, bool kMatsL2 = false, bool kSgLds = false, bool kGridQ = false>
RT_DEV void stage_scene(const TraceParams& p, float4* tabBase, SceneTabs& T) {
    constexpr bool kMfma = kScan == 1 || kScan == 2;
    T.gridCells = p.grid_cell_start;
    T.gridQ = nullptr;
    T.scan = p.scan;
    T.orig = p.orig;
    T.leaf = p.leaf;
    T.rad = p.radius;
    T.mats = p.mats;
    T.mats16 = nullptr;
    T.ops = nullptr;
    T.tree = p.tree;
    T.sgCell = p.sg_cell_start;
    T.sgEntries = p.sg_entries;
    T.sgGlobal = p.sg_global;
    const uint32_t topLevel = p.n_levels - 1u;
    const uint32_t nTop = p.level_cnt[topLevel];
    const uint32_t nTiles = mfma_tiles_for(nTop);  // even; nTop <= 128 => at most four
    T.nTop = nTop;
    T.nTiles = nTiles;
    if (kLds) {
        float4* ldsScan = tabBase;
        float4* ldsLeaf = ldsScan + p.n_padded;
        constexpr bool kLeafLds = kScan == 1 || kScan == 3;  // the scans that test one-sphere bounds from LDS
        // (the flat scan keeps the bounds at kFlatLeafStride float4 per group of four: rt_scan.h)
        constexpr uint32_t kLeafStride = kScan == 1 ? kFlatLeafStride : 4u;
        uint32_t* ldsOrig = reinterpret_cast<uint32_t*>(ldsLeaf + (kLeafLds ? (p.n_padded / 4u) * kLeafStride : 0u));
        float4* ldsMat = reinterpret_cast<float4*>(ldsOrig + p.n_padded);  // n_padded is a multiple of 4
        const uint32_t nMatLds = kMatsL2 ? 0u : ((kHitLds || p.mats_in_lds) ? p.n_padded : 0u);
        // (kMatsL2: the 48-byte table stays in global memory; its 16-byte packed form takes the place when the launch says so)
        const uint32_t nMat16 = (kMatsL2 && p.mats16_mode == 2u) ? p.n_padded : 0u;
        float* ldsRad = reinterpret_cast<float*>(ldsMat + (size_t)nMatLds * 3 + nMat16);
        const float4* gMat = reinterpret_cast<const float4*>(p.mats);
        for (uint32_t k = threadIdx.x; k < p.n_padded; k += blockDim.x) {
            ldsScan[k] = p.scan[k];
            ldsOrig[k] = p.orig[k];
            if (kLeafLds) ldsLeaf[(k >> 2) * kLeafStride + (k & 3u)] = p.leaf[k];
        }
        for (uint32_t k = threadIdx.x; k < nMatLds * 3; k += blockDim.x) ldsMat[k] = gMat[k];
        if (kMatsL2) {
            const float4* g16 = reinterpret_cast<const float4*>(p.mats16);
            for (uint32_t k = threadIdx.x; k < nMat16; k += blockDim.x) ldsMat[k] = g16[k];
            T.mats16 = reinterpret_cast<const uint4*>(ldsMat);
        }
        for (uint32_t k = threadIdx.x; k < p.n_padded; k += blockDim.x) ldsRad[k] = p.radius[k];
        if (kScan == 3) {
            // cell-grid scan with every table in LDS (kHitLds by construction): the cells' first entries, then the shadow index
            uint16_t* gc = reinterpret_cast<uint16_t*>(ldsRad + p.n_padded);
            const uint32_t ng = p.grid_nu * p.grid_nv + 1u;
            for (uint32_t k = threadIdx.x; k < ng; k += blockDim.x) gc[k] = p.grid_cell_start[k];
            T.gridCells = gc;
            uint16_t* g = gc + ((ng + 7u) & ~7u);  // 16-byte steps
            const uint32_t nc = p.sg_nx * p.sg_ny + 1u;
            if (p.sg_enabled) {
                for (uint32_t k = threadIdx.x; k < nc; k += blockDim.x) g[k] = p.sg_cell_start[k];
                for (uint32_t k = threadIdx.x; k < p.sg_nentries; k += blockDim.x) g[nc + k] = p.sg_entries[k];
                for (uint32_t k = threadIdx.x; k < p.sg_nglobal; k += blockDim.x) g[nc + p.sg_nentries + k] = p.sg_global[k];
            }
            T.sgCell = g;
            T.sgEntries = g + nc;
            T.sgGlobal = g + nc + p.sg_nentries;
        }
        if (kMfma) {
            float* ldsOps = ldsRad + p.n_padded;  // a multiple of 4
            build_mfma_operands(p.tree + p.level_off[topLevel], nTop, nTiles, ldsOps, threadIdx.x, blockDim.x);
            T.ops = ldsOps;
            if (kHitLds) {
                // typed LDS pointers, staged only when the scene has an index (the pointers are never read otherwise)
                uint16_t* g = reinterpret_cast<uint16_t*>(ldsOps + (size_t)nTiles * kOpsPerTile);
                const uint32_t nc = p.sg_nx * p.sg_ny + 1u;
                if (p.sg_enabled) {
                    for (uint32_t k = threadIdx.x; k < nc; k += blockDim.x) g[k] = p.sg_cell_start[k];
                    for (uint32_t k = threadIdx.x; k < p.sg_nentries; k += blockDim.x) g[nc + k] = p.sg_entries[k];
                    for (uint32_t k = threadIdx.x; k < p.sg_nglobal; k += blockDim.x) g[nc + p.sg_nentries + k] = p.sg_global[k];
                }
                T.sgCell = g;
                T.sgEntries = g + nc;
                T.sgGlobal = g + nc + p.sg_nentries;
            } else if (p.sg_enabled && p.sg_in_lds) {  // shadow index after the operand image
                uint16_t* g = reinterpret_cast<uint16_t*>(ldsOps + (size_t)nTiles * kOpsPerTile);
                const uint32_t nc = p.sg_nx * p.sg_ny + 1u;
                for (uint32_t k = threadIdx.x; k < nc; k += blockDim.x) g[k] = p.sg_cell_start[k];
                for (uint32_t k = threadIdx.x; k < p.sg_nentries; k += blockDim.x) g[nc + k] = p.sg_entries[k];
                for (uint32_t k = threadIdx.x; k < p.sg_nglobal; k += blockDim.x) g[nc + p.sg_nentries + k] = p.sg_global[k];
                T.sgCell = g;
                T.sgEntries = g + nc;
                T.sgGlobal = g + nc + p.sg_nentries;
            }
        }
        __syncthreads();
        T.scan = ldsScan;
        T.orig = ldsOrig;
        if (kLeafLds) T.leaf = ldsLeaf;
        T.rad = ldsRad;
        if (kMatsL2) T.mats = p.mats;  // (a kernel argument: global address space, no flat loads)
        else if (kHitLds) T.mats = reinterpret_cast<const rt_material*>(ldsMat);
        else if (p.mats_in_lds) T.mats = reinterpret_cast<const rt_material*>(ldsMat);
    } else if (kScan == 3) {
        // cell-grid scan: the exact tables stay in global memory (L2); the cells' first entries live in LDS (kHitLds: typed pointer)
        if (kHitLds) {
            uint16_t* g = reinterpret_cast<uint16_t*>(tabBase);
            const uint32_t ng = p.grid_nu * p.grid_nv + 1u;
            for (uint32_t k = threadIdx.x; k < ng; k += blockDim.x) g[k] = p.grid_cell_start[k];
            T.gridCells = g;
            if (kGridQ) {  // the quantised bounds behind the cells (16-byte steps): 4 bytes per scan entry
                uint32_t* q = reinterpret_cast<uint32_t*>(g + ((ng + 7u) & ~7u));
                for (uint32_t k = threadIdx.x; k < p.n_padded; k += blockDim.x) q[k] = p.grid_qrec[k];
                T.gridQ = q;
            }
            if (kSgLds) {  // the shadow index behind the cells (16-byte steps); the host launches this flavour only when the scene has one
                uint16_t* sgl = g + ((ng + 7u) & ~7u);
                const uint32_t nc = p.sg_nx * p.sg_ny + 1u;
                for (uint32_t k = threadIdx.x; k < nc; k += blockDim.x) sgl[k] = p.sg_cell_start[k];
                for (uint32_t k = threadIdx.x; k < p.sg_nentries; k += blockDim.x) sgl[nc + k] = p.sg_entries[k];
                for (uint32_t k = threadIdx.x; k < p.sg_nglobal; k += blockDim.x) sgl[nc + p.sg_nentries + k] = p.sg_global[k];
                T.sgCell = sgl;
                T.sgEntries = sgl + nc;
                T.sgGlobal = sgl + nc + p.sg_nentries;
            }
            __syncthreads();
        }
    } else if (kMfma) {
        // exact tables stay in global memory (L2); the top level's operand image and, when they fit, all bounds live in LDS
        float* ldsOps = reinterpret_cast<float*>(tabBase);
        build_mfma_operands(p.tree + p.level_off[topLevel], nTop, nTiles, ldsOps, threadIdx.x, blockDim.x);
        T.ops = ldsOps;
        if (kHitLds) {  // hierarchy scan: every level of bounds is in LDS (typed pointer: the descent's node reads are ds_read, not flat)
            float4* ldsTree = reinterpret_cast<float4*>(ldsOps + (size_t)nTiles * kOpsPerTile);
            const uint32_t nNodes = p.level_off[topLevel] + nTop;
            for (uint32_t k = threadIdx.x; k < nNodes; k += blockDim.x) ldsTree[k] = p.tree[k];
            T.tree = ldsTree;
        }
        __syncthreads();
    }
}

// Next block of fresh paths for this wave (rt_params.h, "the queue of fresh paths"); false when every shard is empty.
// What one wave learns about an empty shard it tells its workgroup (SceneConsts::dry_mask, LDS): when a launch runs out of
// work, 4,096 waves asking 8 cursors each are 32,000 requests that the memory side serves one behind the other -- they
// alone were the last 0.4 ms of a launch (tools/timeline_waves.py: a claim took 160-400 us there).
RT_DEV uint32_t dry_shards(SceneConsts* ldsK) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ldsK->dry_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
RT_DEV void mark_dry(SceneConsts* ldsK, uint32_t lane, uint32_t k) {
    if (lane == 0) __hip_atomic_fetch_or(&ldsK->dry_mask, 1u << k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
RT_DEV bool claim_block(const TraceParams& p, SceneConsts* ldsK, uint32_t lane, uint32_t qb, uint32_t& blkNext, uint32_t& blkEnd,
                        uint32_t& shard) {
    RT_SITE(K_CLAIM);
    uint32_t dry = dry_shards(ldsK);
    for (uint32_t a = 0; a < kQueueShards; ++a) {
        const uint32_t k = (blockIdx.x + a) & (kQueueShards - 1u);
        if ((dry >> k) & 1u) continue;
        uint32_t* head = p.shard_heads + kShardStrideWords * k;
        uint32_t c = 0;
        if (lane == 0) c = __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        if (c * kQueueShards + k >= p.dyn_blocks) {
            mark_dry(ldsK, lane, k);
            continue;
        }
        if (lane == 0) c = atomicAdd(head, 1u);
        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        const uint32_t b = c * kQueueShards + k;  // shard k owns the blocks k, k + 8, k + 16, ...
        if (b < p.dyn_blocks) {
            blkNext = p.dyn_begin + b * qb;
            blkEnd = (blkNext + qb) < p.total_paths ? (blkNext + qb) : p.total_paths;
            shard = k;  // where the claim succeeded (the caller claims ahead from the same shard)
            return true;
        }
        mark_dry(ldsK, lane, k);
        dry = dry_shards(ldsK);  // what the others found in the meantime
    }
    shard = kQueueShards;
    return false;
}

// ============================================================================ megakernel
// Persistent threads: every wave loops { refill idle lanes from the queue; one list scan for all
// lanes; per-lane state transition } until the queue is empty and all its lanes are idle.  Waves
// never synchronise with each other after the LDS staging barrier, and every wave's loop ends when
// the (bounded, monotonically consumed) queue is exhausted and its at most 64 paths of at most
// max_depth+1 segments have finished.
// kScan: 0 = VALU sign filter per group (any scene size), 1 = matrix-core filter over the groups (tables in LDS),
// 2 = matrix-core filter over the top level of the bounds hierarchy + per-lane descent (tables through L2).
// kCarry: frame pipelining (rt_params.h): resume the paths the previous trace kernel carried out, end as soon as the
// fresh queue is empty and carry the unfinished paths into the next kernel.
// kStash: REGROUPED HIT PROCESSING (within the wave).  Four closest-hit scans in ten end in the sky, so hit
// processing -- Material::Scatter, the shadow index, DirectionalLight::Shade: 40 % of an iteration -- ran with 36 of 64
// lanes.  Here a lane whose scan found a hit only RECORDS it (state kHaveHit: position in `ro`, scan entry in the high half
// of `depth`) and the wave processes hits when at least stash_cap + 1 lanes hold one; with fewer it pushes them to a
// per-wave stash in LDS (the region the prepared-path cache occupies in the other variants), all 64 lanes generate the next
// 64 fresh paths straight into registers, scan, and the lanes that missed pop stashed hits until the wave is full of
// hits.  Scans run with ~59 live rays, hit processing with ~58 lanes instead of 36.  Every path sees the same sequence of
// operations on the same values as before: only WHEN a hit is processed changes, never what is computed.
// kLights (trace_body only): false = ONE light, the reference's scene (rt_trace_kernel); true = the scene's LIST of lights
// (rt_trace_kernel_lights: Material::Shade's loop, material.cpp:4-13).  Two entry kernels over one body, so that the single-light
// kernels' code and register allocation are exactly what they were without the list.
// kGridQ (cell-grid scan, global tables): the quantised one-sphere bounds are staged into LDS behind the cells (rt_scan.h GridQuant).
template <bool kLights, bool kLds, int kThreads, int kScan, bool kCache, bool kHitLds, bool kCarry, bool kStash, bool kMatsL2, bool kSgLds, bool kGridQ>
__device__ __forceinline__ void trace_body(const TraceParams& p) {
    static_assert(!kGridQ || (kScan == 3 && !kLds && kHitLds && !kSgLds), "quantised bounds: the global-tables grid variant");
    static_assert(!kSgLds || (kScan == 3 && !kLds && kHitLds), "shadow index in LDS next to the grid's cells: the global-tables grid variant");
    static_assert(!kMatsL2 || (kHitLds && kStash && kLds), "materials through L2: a flavour of the all-in-LDS stash variants");
    constexpr bool kMfma = kScan == 1 || kScan == 2;  // matrix-core filter; kScan == 3: cell-grid scan (rt_scan.h scan_list_grid)
    static_assert(!kCarry || (kCache && kHitLds && kScan == 1), "frame pipelining is built for the flat LDS variant only");
    static_assert(!kStash || (kCache && !kCarry), "the hit stash takes the LDS region of the prepared-path cache");
    static_assert(!kHitLds || kScan != 0, "kHitLds belongs to the matrix-core and grid variants");
#ifdef RT_TIMELINE
    const unsigned long long tl0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz; diagnostic build only
    uint32_t tlCarriedIn = 0;
#endif
    RT_SITE(K_WAVE);
    extern __shared__ float4 smem[];
    // [0, kConstBytes): the scene constants; then the per-wave regions; then the tables
    SceneConsts* ldsK = reinterpret_cast<SceneConsts*>(smem);
    if (threadIdx.x == 0) fill_consts(p, *ldsK);
    // the elementary functions' tables behind the constants (log | exp2 | sincos, as MathTabs indexes them)
    unsigned long long* ldsMath = reinterpret_cast<unsigned long long*>(smem + kSceneConstBytes / 16);
    {
        const MathTabs g = default_math_tabs();
        for (uint32_t k = threadIdx.x; k < kMathTabWords; k += blockDim.x)
            ldsMath[k] = k < 94u ? g.log[k] : (k < 127u ? g.exp2[k - 94u] : g.sincos[k - 127u]);
    }
#ifdef RT_NO_LDS_MATH
    const MathTabs mt = default_math_tabs();
#else
    const MathTabs mt{ldsMath, ldsMath + 94, ldsMath + 127};
#endif
    uint16_t* candBase = reinterpret_cast<uint16_t*>(smem + kConstBytes / 16);
    constexpr uint32_t kWaveRegion = wave_region_bytes<kScan>();  // the work lists of the flat, hierarchy or grid scan
    // the shadow index's global list (rt_params.h sg_glob_slots): spheres, then ids, in front of the tables
    float4* globSph = smem + kConstBytes / 16 + (kThreads / kWaveSize) * (kWaveRegion / 16);
    uint16_t* globIds = reinterpret_cast<uint16_t*>(globSph + p.sg_nglobal);
    if (p.sg_glob16 != 0u)
        for (uint32_t k = threadIdx.x; k < p.sg_nglobal; k += blockDim.x) {
            const uint32_t id = p.sg_global[k];
            globSph[k] = p.scan[id];
            globIds[k] = (uint16_t)id;
        }
    float4* tabBase = globSph + p.sg_glob16;
    SceneTabs T;
    stage_scene<kLds, kScan, kHitLds, kMatsL2, kSgLds, kGridQ>(p, tabBase, T);  // ends with a workgroup barrier when it staged anything
    const float4* scanTab = T.scan;
    const uint32_t* origTab = T.orig;
    const float4* leafTab = T.leaf;
    const float* radTab = T.rad;
    const rt_material* matTab = T.mats;
    const uint4* mat16Lds = T.mats16;
    const float* mfmaOps = T.ops;
    const float4* treeTab = T.tree;
    const uint16_t* sgCell = T.sgCell;
    const uint16_t* sgEntries = T.sgEntries;
    const uint16_t* sgGlobal = T.sgGlobal;
    const uint16_t* gridCells = T.gridCells;
    const uint32_t* gridQ = T.gridQ;
    const uint32_t nTop = T.nTop, nTiles = T.nTiles;

    const uint32_t lane = threadIdx.x & (kWaveSize - 1);
    uint16_t* waveCand = candBase + (threadIdx.x / kWaveSize) * (kWaveRegion / 2);
    uint16_t* cand = waveCand + lane;
    __syncthreads();  // the constants block (and, above, the staged tables) are visible to every wave from here on
#ifdef RT_TIMELINE
    const unsigned long long tl1 = __builtin_amdgcn_s_memrealtime();
#endif
    const SceneConsts& K = *ldsK;
    const V3 sunDir = v3(K.sun_dir[0], K.sun_dir[1], K.sun_dir[2]);
    const float aSun = dot3(sunDir, sunDir);  // the `a` of every shadow ray (ray-tracing.cpp:46)

    // per-lane path state
    V3 ro = v3(0.f, 0.f, 0.f), rd = v3(0.f, 0.f, 1.f);  // current ray
    V3 thr = v3(1.f, 1.f, 1.f), rad = v3(0.f, 0.f, 0.f);
    V3 pend = v3(0.f, 0.f, 0.f), nextDir = v3(0.f, 0.f, 0.f);
    StreamDraws draws{Rng{1u, 0u, 0u, 0u}};
    uint32_t q = 0, depth = 0, state = kIdle, pathTrav = 0;  // q = the path's slot in the sample buffer
    uint32_t seq8 = 0;  // kCarry: low byte of the sequence number of the path's region (its call)
    uint32_t carried = 0;  // kCarry: paths this wave carries out (wave-uniform)
    uint32_t itersHere = 0;  // kCarry: iterations of this wave in this launch
    const uint32_t gwave = blockIdx.x * (kThreads / kWaveSize) + threadIdx.x / kWaveSize;  // this wave's index in the grid
    const uint32_t kBlk = p.queue_block;  // paths per queue block (a multiple of 64)
    bool contAfterShadow = false, pathScattered = false;
    uint32_t nTrav = 0, nSeg = 0;

    // wave-uniform queue window and prepared-path cache (48-byte slots: origin, direction, stream state, path index)
    uint32_t blkNext = 0, blkEnd = 0, cachePos = 0, cacheCnt = 0;
    bool queueEmpty = false;
    uint32_t pendShard = kQueueShards;  // shard of the block claimed ahead (kQueueShards: none), wave-uniform
    uint32_t pendCount = 0;             // lane 0: what that claim returned
    if (kCarry) {
        // the wave's own carried paths (rt_params.h): no cursor, no atomics
        const uint32_t nIn = p.cont_in_n[gwave];
#ifdef RT_TIMELINE
        tlCarriedIn = nIn;
#endif
        if (lane < nIn) {
            const ContEntry* e = p.cont_in + ((size_t)gwave * kWaveSize + lane);
            const float4 A = e->a, B = e->b, Cc = e->c, D = e->d, E = e->e;
            ro = v3(A.x, A.y, A.z);
            rd = v3(A.w, B.x, B.y);
            thr = v3(B.z, B.w, Cc.x);
            rad = v3(Cc.y, Cc.z, Cc.w);
            draws.rng = Rng{__float_as_uint(D.x), __float_as_uint(D.y), __float_as_uint(D.z), __float_as_uint(D.w)};
            q = __float_as_uint(E.x);
            depth = __float_as_uint(E.y);
            seq8 = __float_as_uint(E.z);
            pathTrav = __float_as_uint(E.w);
            const float4 F = e->f, G = e->g;
            pend = v3(F.x, F.y, F.z);
            nextDir = v3(F.w, G.x, G.y);
            const uint32_t fl = __float_as_uint(G.z);
            state = fl & 3u;
            contAfterShadow = (fl & 4u) != 0u;
            pathScattered = (fl & 8u) != 0u;
        }
    }
    {
        // the wave's first block(s) of fresh paths are static: block `gwave`; the shards start behind those.  (The carrying
        // kernel uses half-size blocks: a wave can only start paths as lanes fall idle, so a wave whose lanes are held by long
        // paths should own little unstarted work when the queue runs dry -- it must start all of it before it may leave.)
        const uint32_t b0 = gwave * (kBlk * p.static_blocks);
        blkNext = b0 < p.total_paths ? b0 : p.total_paths;
        blkEnd = (b0 + kBlk * p.static_blocks) < p.total_paths ? (b0 + kBlk * p.static_blocks) : p.total_paths;
    }
    // per-wave region behind the tables: the prepared-path cache, or (kStash) the hit stash, p.ray_cache_stride16 float4 per wave
    float4* rayCache = kCache ? smem + p.ray_cache_off16 + (threadIdx.x / kWaveSize) * p.ray_cache_stride16 : nullptr;
    uint32_t stashCnt = 0;  // kStash: records in the wave's stash (wave-uniform)

    unsigned long long dbgScan[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#ifdef RT_TIMELINE
    unsigned long long tlDrain = 0ull;  // when this wave had nothing left to start
    uint32_t drainIters = 0;            // iterations after that
    unsigned long long tlLastClaim = 0ull;
    uint32_t tlBorn = 0;  // per lane: when its path was started
    uint32_t tlBlocks = 0, tlLastBlock = 0, tlIters = 0;
#endif
    (void)dbgScan;
#ifdef RT_STAMPS
    unsigned long long cyRefill = 0, cyScan = 0, cyTrans = 0, cyIters = 0, cyHit[6] = {0, 0, 0, 0, 0, 0};
#endif
    // Next block of fresh paths for this wave; false when every shard is empty.  The block was claimed AHEAD, when the
    // previous one was taken: the atomic's round trip (two of them with the look, ~4 us under load, 117 times per wave on
    // C2) ran under the work on a whole block instead of stalling the wave.  A claim that came back beyond the shard's end
    // falls through to a fresh look.
    auto nextBlock = [&]() -> bool {
        RT_SITE(K_NEXTBLOCK);
        bool got = false;
        if (pendShard < kQueueShards) {
            const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)pendCount);
            const uint32_t b = c * kQueueShards + pendShard;
            if (b < p.dyn_blocks) {
                blkNext = p.dyn_begin + b * kBlk;
                blkEnd = (blkNext + kBlk) < p.total_paths ? (blkNext + kBlk) : p.total_paths;
                got = true;
            } else {
                mark_dry(ldsK, lane, pendShard);
                pendShard = kQueueShards;  // that shard is dry
            }
        }
        if (!got && !claim_block(p, ldsK, lane, kBlk, blkNext, blkEnd, pendShard)) return false;
        // (not in the frame-pipelining kernel: a wave must start everything it owns before it may carry out)
        if (kCarry) pendShard = kQueueShards;
        if (pendShard < kQueueShards && ((dry_shards(ldsK) >> pendShard) & 1u)) pendShard = kQueueShards;
        if (pendShard < kQueueShards && lane == 0) pendCount = atomicAdd(p.shard_heads + kShardStrideWords * pendShard, 1u);  // claim ahead
        return true;
    };
    // A finished path: GetHitColor * exposureAdjustment, spheres-app.cpp:183; one 12-byte store
    auto finishPath = [&]() {
        RT_SITE(K_FINISH);
        const float expo = K.exposure;
        *reinterpret_cast<float3*>(p.samples + (size_t)q * 3) = make_float3(rad.x * expo, rad.y * expo, rad.z * expo);
        if (p.trav_out) p.trav_out[q] = pathTrav;
        state = kIdle;
#ifdef RT_TIMELINE
        if (!kCarry && tlDrain != 0ull && gwave < 4096u) {
            unsigned int* w = g_tlLast + 4u * gwave;
            w[0] = tlBorn; w[1] = depth; w[2] = q; w[3] = pathTrav;
        }
#endif
    };
#ifdef RT_STAMPS
    unsigned long long thA = 0, thB = 0;
#endif
    // Hit processing of one closest hit (scan entry idx at pos) for a lane in state kNeedClosest: Scatter, the shadow
    // query, Emit + Shade, and the path's next state (GetHitColor, spheres-app.cpp:238-257).
    auto processHit = [&](int idx, V3 pos, bool& finished) {
        RT_SITE(H_PROCESS);
        const float4 S = scanTab[idx];
        const float radius = radTab[idx];  // radius and material tables are in scan-entry (clustered) order
        // the packed 16-byte record where the launch provides it: from LDS (kMatsL2 variants, mode 2) or from global memory (mode 1)
        const Mat m = (kMatsL2 && p.mats16_mode == 2u) ? load_material16(mat16Lds, idx)
                                                        : (p.mats16_mode == 1u ? load_material16(p.mats16, idx) : load_material(matTab, idx));
        const V3 center = v3(S.x, S.y, S.z);
        const V3 nrm = div3(pos - center, radius);  // ray-tracing.cpp:58 (true divide; radius > 0)
        V3 atten, local, localOcc, tex;
        RT_STAMP(th0);
        const bool scattered = scatter_only(m, rd, nrm, draws, atten, nextDir, tex, mt, K.sampler);  // Scatter first: it draws (spheres-app.cpp:246)
        RT_STAMP(th1);
        const bool cont = (depth < p.max_depth) && scattered;  // spheres-app.cpp:247
        if (kLights) {
            // Material::Shade over the scene's LIST of lights (material.cpp:4-13): directLighting = 0, += light->Shade(...) in list
            // order; every light casts its shadow ray (one traversal each), answered by the light's own footprint index -- or, for
            // a hit point outside it, by the any-hit over every entry.  (One light -- the reference's scene -- takes the path below.)
            RT_SITE(H_MULTI);
            V3 emitOnly;
            shade_value(K, m, tex, pos, nrm, false, emitOnly, localOcc, mt);  // Emit + 0
            V3 direct = v3(0.f, 0.f, 0.f);
            const float pp = dot3(pos, pos);
            for (uint32_t k = 0; k < p.n_lights; ++k) {
                bool occ;
                V3 sh;
                if (k == 0u) {
                    occ = (K.sg_enabled && pp <= K.sg_p0sq)
                              ? shadow_query(K, scanTab, sgCell, sgEntries, sgGlobal, p.sg_glob16 != 0u, globSph, globIds, (!kLds && kScan >= 2) ? p.sg_sph : nullptr, pos, sunDir, aSun)
                              : any_hit_all(scanTab, p.n_padded, pos, sunDir, aSun);
                    sh = occ ? v3(0.f, 0.f, 0.f) : shade_only(K, m, tex, pos, nrm, mt);
                } else {
                    const LightRec& Lk = p.extra_lights[k - 1u];
                    const V3 dirK = v3(Lk.sun_dir[0], Lk.sun_dir[1], Lk.sun_dir[2]);
                    const float aK = dot3(dirK, dirK);
                    occ = (Lk.sg_enabled && pp <= Lk.sg_p0sq)
                              ? shadow_query(Lk, scanTab, Lk.cell_start, Lk.entries, Lk.global, false, globSph, globIds, (const float4*)nullptr, pos, dirK, aK)
                              : any_hit_all(scanTab, p.n_padded, pos, dirK, aK);
                    sh = occ ? v3(0.f, 0.f, 0.f) : shade_only(Lk, m, tex, pos, nrm, mt);
                }
                ++nTrav;
                ++pathTrav;
                direct = direct + sh;  // (an occluded light's Shade is XM_Zero, light.cpp:15-18)
            }
            rad = rad + thr * (emitOnly + direct);  // radiance += throughput * (Emit + Shade)
            if (cont) {
                thr = thr * atten;
                ro = pos;
                rd = nextDir;
                ++depth;
            } else {
                finished = true;
            }
            return;
        }
        const bool useIndex = K.sg_enabled && dot3(pos, pos) <= K.sg_p0sq;
        bool occluded = false;
        if (useIndex) occluded = shadow_query(K, scanTab, sgCell, sgEntries, sgGlobal, p.sg_glob16 != 0u, globSph, globIds, (!kLds && kScan >= 2) ? p.sg_sph : nullptr, pos, sunDir, aSun);
        RT_STAMP(th2);
        // the Blinn-Phong value (two normalisations, two pows) is only needed when the sun is visible or unknown
        shade_value(K, m, tex, pos, nrm, !occluded, local, localOcc, mt);
        RT_STAMP(th3);
#ifdef RT_STAMPS
        thA = th0;
        thB = th3;
#endif
        RT_ACC(cyHit[0], th0, th1);
        RT_ACC(cyHit[1], th1, th2);
        RT_ACC(cyHit[2], th2, th3);
        if (useIndex) {
            RT_SITE(H_INDEXED);
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
            RT_SITE(H_FARHIT);
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
    };
    for (;;) {
        RT_SITE(K_ITER);
        RT_STAMP(ts0);
        // ------------------------------------------------ refill idle lanes (ballot + prefix)
        // New paths come from a per-wave cache of 64 prepared paths in LDS: when it runs empty ALL 64 lanes generate
        // the next 64 paths of the wave's queue block at once (index arithmetic, stream seeding, Camera::GetRay with
        // its two normalisations: ~350 instructions at full lane utilisation instead of once per handful of idle
        // lanes), and an idle lane just pops a 48-byte slot.  In-flight paths are untouched: the generation only uses
        // temporaries.  Without room for the cache in LDS (kCache == false) idle lanes generate their own path.
        uint64_t idleMask = __ballot(state == kIdle);
        if (kStash) {
            float* stash = reinterpret_cast<float*>(rayCache);
            const uint32_t cap = p.stash_cap;  // <= 63 records of kStashDwords dwords
            // (1) idle lanes take stashed hits (newest first)
            if (stashCnt != 0u && idleMask != 0ull) {
                RT_SITE(K_POP);
                const uint32_t nIdle = (uint32_t)__popcll(idleMask);
                const uint32_t n = stashCnt < nIdle ? stashCnt : nIdle;
                const uint32_t r = prefix_count(idleMask);
                if (state == kIdle && r < n) {
                    RT_SITE(K_POP_LANE);
                    const float* e = stash + (stashCnt - 1u - r);
                    ro = v3(e[0], e[cap], e[2u * cap]);
                    rd = v3(e[3u * cap], e[4u * cap], e[5u * cap]);
                    thr = v3(e[6u * cap], e[7u * cap], e[8u * cap]);
                    rad = v3(e[9u * cap], e[10u * cap], e[11u * cap]);
                    draws.rng = Rng{__float_as_uint(e[12u * cap]), __float_as_uint(e[13u * cap]), __float_as_uint(e[14u * cap]),
                                    __float_as_uint(e[15u * cap])};
                    q = __float_as_uint(e[16u * cap]);
                    depth = __float_as_uint(e[17u * cap]);
                    pathTrav = __float_as_uint(e[18u * cap]);
                    state = kHaveHit;
                }
                stashCnt -= n;
            }
            // (2) process the hits when (nearly) every lane holds one; with fewer and nothing to scan, make room for 64 fresh paths
            const uint64_t hitMask = __ballot(state == kHaveHit);
            const uint32_t nHit = (uint32_t)__popcll(hitMask);
            bool process = nHit > p.stash_process;
            if (!process && nHit != 0u && __ballot(state == kNeedClosest || state == kNeedShadow) == 0ull) {
                if (blkNext == blkEnd && !queueEmpty && !nextBlock()) queueEmpty = true;
                if (blkNext == blkEnd) {
                    process = true;  // no fresh paths left: the hits are processed as they are
                } else {
                    RT_SITE(K_PUSH);
                    // (the stash is empty here: had records been left after (1), no lane would be idle, so with nothing to
                    // scan all 64 would hold a hit)
                    wave_lds_handoff();  // the pops above have read their records
                    if (state == kHaveHit) {
                        RT_SITE(K_PUSH_LANE);
                        float* e = stash + prefix_count(hitMask);
                        e[0] = ro.x; e[cap] = ro.y; e[2u * cap] = ro.z;
                        e[3u * cap] = rd.x; e[4u * cap] = rd.y; e[5u * cap] = rd.z;
                        e[6u * cap] = thr.x; e[7u * cap] = thr.y; e[8u * cap] = thr.z;
                        e[9u * cap] = rad.x; e[10u * cap] = rad.y; e[11u * cap] = rad.z;
                        e[12u * cap] = __uint_as_float(draws.rng.s0); e[13u * cap] = __uint_as_float(draws.rng.s1);
                        e[14u * cap] = __uint_as_float(draws.rng.s2); e[15u * cap] = __uint_as_float(draws.rng.s3);
                        e[16u * cap] = __uint_as_float(q);
                        e[17u * cap] = __uint_as_float(depth);
                        e[18u * cap] = __uint_as_float(pathTrav);
                        state = kIdle;
                    }
                    stashCnt = nHit;
                    wave_lds_handoff();  // records are popped by other lanes than the ones that pushed them
                }
            }
            if (process) {
                RT_SITE(K_PROCESS);
                bool finished = false;
                if (state == kHaveHit) {
                    const int hidx = (int)(depth >> 16);
                    depth &= 0xffffu;
                    state = kNeedClosest;
                    processHit(hidx, ro, finished);
                }
                if (finished) finishPath();
            }
            // (3) fresh paths: whenever every lane is idle -- at the start and after each push -- all 64 lanes generate the
            // next 64 paths of the wave's queue block straight into their registers (index arithmetic, stream seeding,
            // Camera::GetRay with its two normalisations at full lane utilisation, no LDS round trip)
            if (__ballot(state != kIdle) == 0ull) {
                if (blkNext == blkEnd && !queueEmpty && !nextBlock()) queueEmpty = true;
                if (blkNext != blkEnd) {
                    RT_SITE(K_GEN);
#ifdef RT_TIMELINE
                    if (blkNext % kBlk == 0u) {
                        tlLastClaim = __builtin_amdgcn_s_memrealtime();
                        ++tlBlocks;
                        tlLastBlock = blkNext / kBlk;
                    }
#endif
                    const uint32_t nGen = (blkEnd - blkNext) < (uint32_t)kWaveSize ? (blkEnd - blkNext) : (uint32_t)kWaveSize;
                    if (lane < nGen) {
                        RT_SITE(K_GEN_LANE);
                        uint32_t i, j, sN;
                        path_coordinates(p, blkNext + lane, i, j, sN, q);
                        draws.rng = rng_seed(p.seed, j * p.W + i, sN);
                        gen_primary_ray(K, i, j, sN, ro, rd);
                        thr = v3(1.f, 1.f, 1.f);
                        rad = v3(0.f, 0.f, 0.f);
                        depth = 0;
                        pathTrav = 0;
                        state = kNeedClosest;
#ifdef RT_TIMELINE
                        tlBorn = (uint32_t)(__builtin_amdgcn_s_memrealtime() - tl0);
#endif
                    }
                    blkNext += nGen;
                }
            }
        } else if (kCache) {
            while (idleMask != 0ull) {
                if (cachePos == cacheCnt) {
                    if (queueEmpty) break;
                    if (blkNext == blkEnd && !nextBlock()) {
                        queueEmpty = true;
                        break;
                    }
#ifdef RT_TIMELINE
                    if (blkNext % kBlk == 0u) {
                        tlLastClaim = __builtin_amdgcn_s_memrealtime();
                        ++tlBlocks;
                        tlLastBlock = blkNext / kBlk;
#ifdef RT_TIMELINE_RING
                        if (!kCarry && gwave < 4096u && lane == 0)
                            g_tlRing[16u * gwave + (tlBlocks & 15u)] = ((tlLastClaim - tl0) << 24) | (unsigned long long)(__popcll(__ballot(state != kIdle)) & 127) << 17 | (tlLastBlock >> 3);
#endif
                    }
#endif
                    const uint32_t nGen = (blkEnd - blkNext) < (uint32_t)kWaveSize ? (blkEnd - blkNext) : (uint32_t)kWaveSize;
                    wave_lds_handoff();  // every slot of the previous batch has been popped
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
                    wave_lds_handoff();  // slots are popped by other lanes than the ones that filled them
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
                    if (kCarry) seq8 = p.region_seq & 255u;
                    state = kNeedClosest;
#ifdef RT_TIMELINE
                    tlBorn = (uint32_t)(__builtin_amdgcn_s_memrealtime() - tl0);
#endif
                }
                cachePos += want < avail ? want : avail;
                idleMask = __ballot(state == kIdle);
            }
        } else {
            while (idleMask != 0ull && !queueEmpty) {
                if (blkNext == blkEnd) {
                    uint32_t shardUnused = 0;
                    if (!claim_block(p, ldsK, lane, kBlk, blkNext, blkEnd, shardUnused)) {
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
        if (__ballot(state != kIdle) == 0ull) {
            // queue empty and every lane drained -- but hits may still wait in the stash: a wave of 64 hits whose paths ALL
            // ended at that hit (a tile looking at an emissive sphere) leaves every lane idle with records behind it
            if (kStash && stashCnt != 0u) continue;  // back to the top: the idle lanes pop them (at most 63: one more round)
            break;
        }
#ifdef RT_TIMELINE
        ++tlIters;
        if (queueEmpty && cachePos == cacheCnt) {
            if (tlDrain == 0ull) tlDrain = __builtin_amdgcn_s_memrealtime();
            ++drainIters;
        }
#endif
        if (kCarry) {
            ++itersHere;
            // Nothing left to start: carry the live paths into the next call's kernel instead of running their tail here.
            // A lane is carried between two scans, whichever it waits for (closest hit, or the shadow scan of a far hit point),
            // but not when its region is max_carry_age calls old (its slot of the sample ring is about to be reused): such
            // paths make the wave iterate on (<= 102 iterations).
            // (min_iters: a wave without fresh work must still advance the paths it was handed, or they would be passed from
            // kernel to kernel untouched until the age limit makes some wave run their whole tail)
            const bool nothingToStart = queueEmpty && cachePos == cacheCnt && p.carry != 0u && itersHere >= p.min_iters;
            const bool carriable = state == kIdle || ((p.region_seq - seq8) & 255u) < p.max_carry_age;
            if (nothingToStart && __ballot(!carriable) == 0ull) {
                const uint64_t liveMask = __ballot(state != kIdle);
                carried = (uint32_t)__popcll(liveMask);
                uint32_t age1 = 0;  // 1 + region age of this lane's carried path
                if (state != kIdle) {
                    ContEntry* e = p.cont_out + ((size_t)gwave * kWaveSize + prefix_count(liveMask));
                    e->a = make_float4(ro.x, ro.y, ro.z, rd.x);
                    e->b = make_float4(rd.y, rd.z, thr.x, thr.y);
                    e->c = make_float4(thr.z, rad.x, rad.y, rad.z);
                    e->d = make_float4(__uint_as_float(draws.rng.s0), __uint_as_float(draws.rng.s1), __uint_as_float(draws.rng.s2),
                                       __uint_as_float(draws.rng.s3));
                    e->e = make_float4(__uint_as_float(q), __uint_as_float(depth), __uint_as_float(seq8), __uint_as_float(pathTrav));
                    e->f = make_float4(pend.x, pend.y, pend.z, nextDir.x);
                    e->g = make_float4(nextDir.y, nextDir.z, __uint_as_float(state | (contAfterShadow ? 4u : 0u) | (pathScattered ? 8u : 0u)), 0.f);
                    age1 = 1u + ((p.region_seq - seq8) & 255u);
                }
                for (int off = 32; off > 0; off >>= 1) {
                    const uint32_t o2 = (uint32_t)__shfl_xor((int)age1, off);
                    age1 = o2 > age1 ? o2 : age1;
                }
                if (lane == 0) atomicMax(&ldsK->exit_age_max, age1);  // LDS: the workgroup's oldest carried region
                break;
            }
        }

        RT_STAMP(ts1);
        // ------------------------------------------------ one list scan for every live lane
        float tmin = 0.f;
        int idx = -1;
        const bool live = kStash ? (state == kNeedClosest || state == kNeedShadow) : (state != kIdle);  // (lanes holding a hit wait)
        if (kMfma) {
            // every lane takes part: lane l also supplies operands for, and filters half the spheres of,
            // the ray owned by lane l^32, whether or not its own ray is live
            scan_list_mfma<kScan == 2>(scanTab, leafTab, origTab, mfmaOps, nTiles, nTop, treeTab, p.level_off, p.n_levels, p.bound_norm, p.single_mask, p.n_always, p.tree_box_on ? p.tree_box : nullptr, ro, rd,
                                       live, tmin, idx, waveCand, lane, dbgScan);
        } else if (kScan == 3) {
            const GridParams G{gridCells, p.grid_nu, p.grid_nv, p.grid_ax_u, p.grid_ax_v, p.grid_g0u, p.grid_g0v, p.grid_inv_h, p.grid_rmax_over_h, p.grid_big_norm};
            const GridQuant Q{gridQ, p.grid_q[0], p.grid_q[1], p.grid_q[2], p.grid_q[3], p.grid_q[4], p.grid_q[5], p.grid_q[6], p.grid_q[7]};
            scan_list_grid<kGridQ>(scanTab, leafTab, origTab, G, gridCells, p.n_always, p.tree_box, p.bound_norm, ro, rd, live, tmin, idx, waveCand, lane, Q);
        } else if (live) {
            scan_list_deferred(scanTab, origTab, p.n_padded, ro, rd, tmin, idx, cand);
        }
        if (live) {
            ++nTrav;
            ++pathTrav;
        }

        RT_STAMP(ts2);
        // ------------------------------------------------ state transitions
        bool finished = false;
#ifdef RT_STAMPS
        thA = 0;
        thB = 0;
#endif
        if (state == kNeedClosest) {
            ++nSeg;
            if (idx < 0) {
                RT_SITE(K_TRANS_MISS);
                // miss: sky Emissive::Emit (spheres-app.cpp:255)
                const V3 sky = v3(K.sky_emit[0], K.sky_emit[1], K.sky_emit[2]);
                rad = rad + thr * sky;
                finished = true;
            } else if (kStash) {
                RT_SITE(K_TRANS_HIT);
                // the hit waits for a full wave of hits: position (XMVectorMultiplyAdd(t, dir, origin), ray-tracing.cpp:57) in
                // place of the origin, scan entry beside the depth
                ro = tmin * rd + ro;
                depth |= (uint32_t)idx << 16;
                state = kHaveHit;
            } else {
                processHit(idx, tmin * rd + ro, finished);  // XMVectorMultiplyAdd(t, dir, origin), ray-tracing.cpp:57
            }
        } else if (state == kNeedShadow) {
            RT_SITE(K_TRANS_SHADOW);
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
        if (finished) finishPath();
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

    if (kCarry) {
        // the wave that leaves last tells the commit kernel which region is the oldest with paths still in flight
        if (lane == 0) {
            p.cont_out_n[gwave] = carried;
            const uint32_t ticket = atomicAdd(&ldsK->exit_ticket, 1u);
            if (ticket == (uint32_t)(kThreads / kWaveSize) - 1u) {
                const uint32_t m = atomicMax(&ldsK->exit_age_max, 0u);
                if (m != 0u) atomicMin(&p.ctl->oldest_open, p.region_seq - (m - 1u));
            }
        }
    }
#ifdef RT_TIMELINE
    if (lane == 0) {
        const unsigned long long tl2 = __builtin_amdgcn_s_memrealtime();
        // one 64-byte record per wave, plain stores: shared counters here would be 4,096 x n same-line atomics at ~11 ns
        // each when the waves leave together -- the instrument would make the tail it is meant to measure
        if (gwave < 4096u) {
            unsigned long long* w = g_tlWave + 8u * gwave;
            w[0] = tl0; w[1] = kCarry ? tl1 : (tlDrain != 0ull ? tlDrain : tl2); w[2] = tl2; w[3] = tlBlocks; w[4] = kCarry ? carried : tlLastBlock;
            w[5] = kCarry ? ((tlLastClaim - tl0) << 8 | (unsigned long long)tlCarriedIn) : tlLastClaim; w[6] = tlIters; w[7] = drainIters;
        }
    }
#endif
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

template <bool kLds, int kThreads, int kScan, bool kCache, bool kHitLds = false, bool kCarry = false, bool kStash = false, bool kMatsL2 = false,
          bool kSgLds = false, bool kGridQ = false>
__global__ void __launch_bounds__(kThreads, 1) rt_trace_kernel(const TraceParams p) {
    trace_body<false, kLds, kThreads, kScan, kCache, kHitLds, kCarry, kStash, kMatsL2, kSgLds, kGridQ>(p);
}
// ... and for scenes whose light list has another length than one (rt_scene_upload, n_lights != 1)
template <bool kLds, int kThreads, int kScan, bool kCache, bool kHitLds = false, bool kCarry = false, bool kStash = false, bool kMatsL2 = false,
          bool kSgLds = false, bool kGridQ = false>
__global__ void __launch_bounds__(kThreads, 1) rt_trace_kernel_lights(const TraceParams p) {
    trace_body<true, kLds, kThreads, kScan, kCache, kHitLds, kCarry, kStash, kMatsL2, kSgLds, kGridQ>(p);
}

// ============================================================ ray-generation tables (A1, A9)
// jitter[k] = Halton2D(s0+k; 2,3) (spheres-app.cpp:140), lens[k] = HaltonSampleDisk(k0+k; 4,5) (:152)
// The first threads also set the queue cursors (and, frame pipelining, the control block) for the trace kernel that follows.
__global__ void __launch_bounds__(256) rt_raygen_tables_kernel(float2* jitter, uint32_t s0, uint32_t nJitter, float2* lens, uint32_t k0,
                                                               uint32_t nLens, uint32_t sampler, uint32_t* shardHeads = nullptr,
                                                               FrameCtl* ctl = nullptr) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (shardHeads && k < kQueueShards) shardHeads[kShardStrideWords * k] = 0u;
    if (ctl && k == 0) ctl->oldest_open = 0xffffffffu;
    if (k < nJitter) jitter[k] = make_float2(halton(s0 + k, 2), halton(s0 + k, 3));
    if (k < nLens) {
        float lx, ly;
        halton_disk_4_5(k0 + k, lx, ly, sampler);
        lens[k] = make_float2(lx, ly);
    }
}

// ====================================================== work order of the tiles (scheduling only)
// A persistent launch ends with the tail of whatever it started last; if those are 51-segment paths through the glass
// spheres, most of the chip waits for them (2 % of a C2 launch, 11 % at spp 16).  Three pilot rays per full tile (its first,
// middle and last pixel, sample 1) are traced through the production scan when an accumulation starts; the tile's class is
// the most expensive material among their first hits -- nothing, anything else, metal, glass -- and the launches of the
// accumulation take the tiles by descending class: the sky, whose paths end after one scan, comes last.  Results do not
// depend on the order (every path has its own slot and stream); only the schedule does.
#ifndef RT_PILOTS_PER_TILE
#define RT_PILOTS_PER_TILE 3
#endif
constexpr uint32_t kPilotsPerTile = RT_PILOTS_PER_TILE;  // evenly spaced over the tile's 64 pixels, both ends included
__global__ void __launch_bounds__(256) rt_pilot_rays_kernel(const TraceParams p, uint32_t nFull, float* rays) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nFull * kPilotsPerTile) return;
    const uint32_t tile = t / kPilotsPerTile, which = t - tile * kPilotsPerTile;
    const uint32_t pl = (tile << 6) + (which * 63u + (kPilotsPerTile - 1u) / 2u) / (kPilotsPerTile - 1u);
    const uint32_t lr = pl / p.W;
    V3 o, d;
    gen_primary_ray(p, pl - lr * p.W, rowset_global_row(p.rs, lr), 1u, o, d);
    float* w = rays + 6 * (size_t)t;
    w[0] = o.x; w[1] = o.y; w[2] = o.z; w[3] = d.x; w[4] = d.y; w[5] = d.z;
}
// hits: rt_unit_closest_hit records (10 floats; [1] = original sphere index as bits, < 0 = miss), kPilotsPerTile per tile
RT_DEV uint32_t pilot_class(const float* hits, uint32_t t, const uint32_t* matTypeByOrig) {
    uint32_t c = 0;
    for (uint32_t k = 0; k < kPilotsPerTile; ++k) {
        const int oidx = __float_as_int(hits[10 * ((size_t)t * kPilotsPerTile + k) + 1]);
        if (oidx >= 0) {
            const uint32_t ty = matTypeByOrig[oidx];
            const uint32_t ck = ty == RT_MAT_DIELECTRIC_TRANSPARENT ? 3u : (ty == RT_MAT_METAL ? 2u : 1u);
            c = ck > c ? ck : c;
        }
    }
    return c;
}
// A tile next to a glass tile -- left, right, in the row above or below -- counts as glass too: the cap of a glass sphere
// that sticks out into the sky can be narrower than the pilots' spacing, and its 51-segment paths, started with the sky in
// the launch's last half millisecond, were the last three waves of a C2 launch (+1 % of its duration; tools/timeline_bulk.py).
__global__ void __launch_bounds__(256) rt_tile_class_kernel(const float* hits, uint32_t nFull, uint32_t W, const uint32_t* matTypeByOrig,
                                                            uint8_t* cls) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nFull) return;
    uint32_t c = pilot_class(hits, t, matTypeByOrig);
    if (c != 3u) {
        const int64_t first = (int64_t)t << 6;
        const int64_t near[6] = {first - 64, first + 64, first - (int64_t)W, first - (int64_t)W + 63, first + (int64_t)W, first + (int64_t)W + 63};
        for (int k = 0; k < 6; ++k) {
            const int64_t n = near[k] >> 6;  // the tile of that pixel
            if (near[k] >= 0 && n < (int64_t)nFull && pilot_class(hits, (uint32_t)n, matTypeByOrig) == 3u) c = 3u;
        }
    }
    cls[t] = (uint8_t)c;
}
// Stable counting sort of the tiles by descending class; one workgroup, thread k owns a contiguous run of tiles.
__global__ void __launch_bounds__(1024) rt_tile_order_kernel(const uint8_t* cls, uint32_t nFull, uint32_t* order) {
    __shared__ uint32_t waveTot[16][4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint32_t per = (nFull + 1023u) / 1024u;
    const uint32_t t0 = tid * per < nFull ? tid * per : nFull, t1 = (t0 + per) < nFull ? (t0 + per) : nFull;
    uint32_t cnt[4] = {0u, 0u, 0u, 0u};
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t c = cls[t];
        for (uint32_t k = 0; k < 4u; ++k) cnt[k] += c == k ? 1u : 0u;
    }
    uint32_t pos[4];  // where this thread's tiles of class k start
    for (uint32_t k = 0; k < 4u; ++k) {
        uint32_t incl = cnt[k];
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, off);
            if ((int)lane >= off) incl += v;
        }
        if (lane == 63u) waveTot[w][k] = incl;
        pos[k] = incl - cnt[k];
    }
    __syncthreads();
    uint32_t base = 0;
    for (int k = 3; k >= 0; --k) {
        uint32_t before = 0, tot = 0;
        for (uint32_t v = 0; v < 16u; ++v) {
            before += v < w ? waveTot[v][k] : 0u;
            tot += waveTot[v][k];
        }
        pos[k] += base + before;
        base += tot;
    }
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t c = cls[t];
        uint32_t at = 0;
        for (uint32_t k = 0; k < 4u; ++k) {
            at = c == k ? pos[k] : at;
            pos[k] += c == k ? 1u : 0u;
        }
        order[at] = t;
    }
}

// ============================================================== ordered accumulation (A16)
// hdr[pixel] += sample(pixel, s) for s = s0 .. s0+spp-1 in that order (spheres-app.cpp:182-183).
__global__ void __launch_bounds__(256) rt_accumulate_kernel(const float* __restrict__ samples, float* __restrict__ hdr, uint32_t npix,
                                                            uint32_t spp, uint32_t first = 0, uint32_t count = 0xffffffffu) {
    const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    float r = hdr[3 * (size_t)pix], g = hdr[3 * (size_t)pix + 1], b = hdr[3 * (size_t)pix + 2];
    // tiled buffer [tile of 64 pixels][sample][pixel in tile] (path_coordinates): the 64 lanes of a wave read 768
    // contiguous bytes per sample plane; eight planes in flight, the adds stay sequential in s.  first / count: the planes
    // [first, first + count) of the buffer's spp (render-ahead batching adds the planes of one call at a time).
    const uint32_t nFull = npix >> 6, tile = pix >> 6;
    const uint32_t stride = tile < nFull ? 64u : npix - (nFull << 6);
    const float3* sp = reinterpret_cast<const float3*>(samples) + (size_t)tile * 64u * spp + (pix - (tile << 6));
    uint32_t s = first < spp ? first : spp;
    const uint32_t end = count > spp - s ? spp : s + count;
    for (; s + 8 <= end; s += 8) {
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
    for (; s < end; ++s) {
        const float3 v = sp[(size_t)s * stride];
        r += v.x;
        g += v.y;
        b += v.z;
    }
    hdr[3 * (size_t)pix] = r;
    hdr[3 * (size_t)pix + 1] = g;
    hdr[3 * (size_t)pix + 2] = b;
}

// ================================================== ordered accumulation of pipelined regions (A16)
// Frame pipelining: add every region that is complete -- sequence numbers committed+1 .. min(newest, oldest_open - 1) -- to
// the HDR strip, region by region and sample by sample in increasing s (spheres-app.cpp:182-183).  The commit point is kept
// TWICE (FrameCtl): commit kernel number k reads entry k & 1 and its block 0 writes the new point to entry (k + 1) & 1, which
// nobody reads before the next commit kernel -- no closing ticket, no "last block".  Every block therefore reads the same
// committed_seq / oldest_open: the entry it reads is not written during this kernel, oldest_open only by the trace kernels.
struct RegionTable {
    uint32_t seq[kMaxFramesInFlight];  // by ring slot (sequence % regions): the call that owns it
    uint32_t spp[kMaxFramesInFlight];  // ... and its samples per pixel
};
__global__ void __launch_bounds__(256) rt_commit_kernel(const float* __restrict__ ring, float* __restrict__ hdr, uint32_t npix,
                                                        uint32_t regionEntries, uint32_t nRing, RegionTable rt, FrameCtl* ctl,
                                                        uint32_t newestSeq, uint32_t commitNo) {
    const uint32_t committed = ctl->committed_seq[commitNo & 1u], open = ctl->oldest_open;
    uint32_t limit = newestSeq;
    if (open != 0xffffffffu && open - 1u < limit) limit = open - 1u;
    if (limit < committed) limit = committed;
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // the other entry: nobody reads it before the next commit kernel
        uint32_t add = 0;
        for (uint32_t q = committed + 1u; q <= limit; ++q) add += rt.spp[q % nRing];
        ctl->committed_seq[(commitNo + 1u) & 1u] = limit;
        ctl->committed_samples[(commitNo + 1u) & 1u] = ctl->committed_samples[commitNo & 1u] + add;
    }
    if (limit == committed) return;
    for (uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += gridDim.x * blockDim.x) {
        float r = hdr[3 * (size_t)pix], g = hdr[3 * (size_t)pix + 1], b = hdr[3 * (size_t)pix + 2];
        const uint32_t nFull = npix >> 6, tile = pix >> 6;
        const uint32_t stride = tile < nFull ? 64u : npix - (nFull << 6);
        for (uint32_t q = committed + 1u; q <= limit; ++q) {
            const uint32_t slot = q % nRing, spp = rt.spp[slot];
            // the region's buffer is tiled like the one-shot sample buffer: [tile of 64 pixels][sample][pixel in tile]
            const float3* sp = reinterpret_cast<const float3*>(ring) + (size_t)slot * regionEntries + (size_t)tile * 64u * spp + (pix - (tile << 6));
            for (uint32_t s = 0; s < spp; ++s) {
                const float3 v = sp[(size_t)s * stride];
                r += v.x;
                g += v.y;
                b += v.z;
            }
        }
        hdr[3 * (size_t)pix] = r;
        hdr[3 * (size_t)pix + 1] = g;
        hdr[3 * (size_t)pix + 2] = b;
    }
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
// committedSamples (frame pipelining): the number of samples in the strip is only known on the device
__global__ void __launch_bounds__(256) rt_resolve_kernel(const float* __restrict__ hdr, uint8_t* __restrict__ ldr, uint32_t npix,
                                                         uint32_t nSamples, const uint32_t* committedSamples = nullptr) {
    const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    if (committedSamples) nSamples = *committedSamples > 0u ? *committedSamples : 1u;
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
#if RT_MARKSTEIN
    else if (op == 4) r = recip_rn(x[k]);  // must equal 1.0f / x for normal x in [2^-100, 2^100]
    else if (op == 5) {                      // the quotient the hit processing forms (guarded): must equal x / y
        const V3 q = div3(v3(x[k], 0.f, -x[k]), y[k]);
        r = q.x;
    }
#else
    else if (op == 4) r = 1.0f / x[k];
    else if (op == 5) r = x[k] / y[k];
#endif
    else if (op == 6) r = x[k] / y[k];      // the compiler's IEEE division, on the device
    else if (op == 7) r = sqrt_rn(x[k]);    // the square root the path takes (guarded): must equal the IEEE root for every x
#if RT_FAST_SQRT
    else if (op == 8) r = sqrt_rn_core(x[k]);  // ... its fast form alone: the IEEE root for x in [2^-80, inf)
#else
    else if (op == 8) r = __builtin_sqrtf(x[k]);
#endif
    else if (op == 9) r = __builtin_sqrtf(x[k]);  // the compiler's IEEE square root, on the device
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
// Closest hit through the PRODUCTION scan of the uploaded scene: the same table staging (stage_scene) and the same
// scan function (scan_list_mfma flat / hierarchy, or scan_list_deferred) as the variant of rt_trace_kernel that
// rt_render launches for this scene, 64 rays per wave.  LDS image: per-wave work-list regions, then the tables.
template <bool kLds, int kScan>
__global__ void __launch_bounds__(256) k_unit_closest(const TraceParams p, const float* rays, uint32_t n, float* out) {
    extern __shared__ float4 smem[];
    constexpr uint32_t kWaveRegion = wave_region_bytes<kScan>();
    uint16_t* candBase = reinterpret_cast<uint16_t*>(smem);
    float4* tabBase = smem + (256 / kWaveSize) * (kWaveRegion / 16);
    SceneTabs T;
    stage_scene<kLds, kScan>(p, tabBase, T);
    __syncthreads();
    const uint32_t lane = threadIdx.x & (kWaveSize - 1);
    uint16_t* waveCand = candBase + (threadIdx.x / kWaveSize) * (kWaveRegion / 2);
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = k < n;
    V3 o = v3(0.f, 0.f, 0.f), d = v3(0.f, 0.f, 1.f);
    if (live) {
        const float* r = rays + 6 * (size_t)k;
        o = v3(r[0], r[1], r[2]);
        d = v3(r[3], r[4], r[5]);
    }
    float tmin = 0.f;
    int idx = -1;
    unsigned long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)dbg;
    if (kScan == 3) {
        const GridParams G{T.gridCells, p.grid_nu, p.grid_nv, p.grid_ax_u, p.grid_ax_v, p.grid_g0u, p.grid_g0v, p.grid_inv_h, p.grid_rmax_over_h, p.grid_big_norm};
        scan_list_grid(T.scan, T.leaf, T.orig, G, T.gridCells, p.n_always, p.tree_box, p.bound_norm, o, d, live, tmin, idx, waveCand, lane);
    } else if (kScan != 0) {
        scan_list_mfma<kScan == 2>(T.scan, T.leaf, T.orig, T.ops, T.nTiles, T.nTop, T.tree, p.level_off, p.n_levels, p.bound_norm, p.single_mask, p.n_always, p.tree_box_on ? p.tree_box : nullptr, o, d, live,
                                   tmin, idx, waveCand, lane, dbg);
    } else if (live) {
        scan_list_deferred(T.scan, T.orig, p.n_padded, o, d, tmin, idx, waveCand + lane);
    }
    if (!live) return;
    float* w = out + 10 * (size_t)k;
    for (int c = 0; c < 10; ++c) w[c] = 0.f;
    const int oidx = idx >= 0 ? (int)T.orig[idx] : -1;
    w[1] = __int_as_float(oidx);
    if (idx >= 0) {
        const float4 S = T.scan[idx];
        const V3 pos = tmin * d + o;
        const V3 nrm = (pos - v3(S.x, S.y, S.z)) / T.rad[idx];
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
    const bool sc = scatter_and_shade(p, m, v3(q[0], q[1], q[2]), v3(q[3], q[4], q[5]), v3(q[6], q[7], q[8]), draws, atten, dir, local, localOcc, p.sampler);
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

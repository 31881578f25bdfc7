// rt_capi.hip — C-ABI implementation (include/rt_api.h) over the HIP kernels in rt_kernels.h.
// Builds into librt_hip.so with hipcc --offload-arch=gfx950.  No CPU fallback exists: every entry
// point that needs the device fails with RT_ERR_NO_DEVICE / RT_ERR_HIP when it is absent.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <array>
#include <utility>
#include <vector>

#include "../../include/rt_api.h"
#include "rt_kernels.h"

namespace {

thread_local std::string g_err;

int Fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define RT_HIP(call)                                                                                             \
    do {                                                                                                         \
        hipError_t e_ = (call);                                                                                  \
        if (e_ != hipSuccess) {                                                                                  \
            return Fail(e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP,                           \
                        std::string(#call) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
        }                                                                                                        \
    } while (0)

uint32_t EnvU32(const char* name, uint32_t dflt) {
    const char* v = std::getenv(name);
    if (!v || !*v) return dflt;
    return (uint32_t)std::strtoul(v, nullptr, 10);
}

// Largest top level the matrix-core filter takes (RT_TREE_TOP, default 128 = four tiles of 32): one reader for rt_create and for
// the GPU-less layout queries (rt_unit_layout, rt_unit_layout_info), so that they describe the layout an upload would build.
uint32_t TreeTopFromEnv() {
    const uint32_t t = EnvU32("RT_TREE_TOP", 128);
    return (t < 4 || t > 128) ? 128u : t;
}

template <typename T>
struct DevBuf {
    T* ptr = nullptr;
    size_t count = 0;
    int Reserve(size_t n) {
        if (n <= count) return RT_OK;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T));
        if (e != hipSuccess) {
            (void)hipGetLastError();  // the failure is reported here; later hipGetLastError() checks must not see it again
            return Fail(RT_ERR_OUT_OF_MEMORY, std::string("hipMalloc: ") + hipGetErrorString(e));
        }
        count = n;
        return RT_OK;
    }
    void Release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
};

}  // namespace

struct rt_ctx {
    int device = 0;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> passEv;  // three timing events per sample-range pass of rt_render, grown on demand
    int cuCount = 0;
    size_t ldsPerBlockMax = 0;
    uint64_t workspaceLimit = 8ull << 30;

    // scene
    bool hasScene = false;
    uint32_t n = 0;
    DevBuf<float4> scan, tree, leaf;
    DevBuf<uint32_t> orig;
    DevBuf<uint16_t> sgCells, sgEntries, sgGlobal;
    DevBuf<float4> sgSph;  // large scenes: the scan record of every shadow-index entry (rt_params.h sg_sph)
    // lights 1 .. n_lights-1 (rt_params.h LightRec): their records and their shadow indices, in global memory
    struct ExtraLight {
        DevBuf<uint16_t> cells, entries, global;
    };
    ExtraLight extraIdx[RT_MAX_LIGHTS];
    DevBuf<rtd::LightRec> lightRecs;
    DevBuf<uint16_t> gridCells;  // cell-grid scan: first scan entry per cell
    DevBuf<uint32_t> gridQ;      // ... and the quantised one-sphere bounds per scan entry (rt_scan.h GridQuant)
    bool useShadowGrid = true;  // RT_SHADOW_GRID=0 keeps every shadow ray on the scan
    DevBuf<float> radius;
    DevBuf<rt_material> mats;
    DevBuf<uint4> mats16;     // the same table packed into 16 bytes per entry (rt_shade.h load_material16), when the scene allows it
    bool mats16Ok = false;
    rtd::TraceParams base{};  // scene part filled at upload

    // accumulation state
    uint32_t W = 0, H = 0, rows = 0;
    rt_rowset rs{};
    uint32_t accumulated = 0;  // samples accumulated so far (next s0 must be accumulated + 1)
    uint32_t sampler = 0;      // RT_SAMPLER_* flags of the next / running accumulation (rt_set_sampler)
    DevBuf<float> hdr;         // [W*rows*3]
    DevBuf<uint8_t> ldr;       // [W*rows*3]
    DevBuf<float> samples;     // workspace [pixels*spp_pass*3]
    DevBuf<uint32_t> queue;    // [1]
    DevBuf<unsigned long long> counters;  // [2]
    DevBuf<float2> jitterTab, lensTab;    // ray-generation tables of the current pass
    double lastResolveMs = 0.0;

    // tuning (env: RT_BLOCKS_PER_CU, RT_FORCE_GLOBAL_TABLES)
    uint32_t blocksPerCu = 4;
    uint32_t blockThreads = 256;
    bool useMfma = true;  // matrix-core pre-filter for the list scan (RT_SCAN=valu disables)
    bool matsInLds = true;   // RT_MATS_LDS=0 leaves the material table in global memory (frees 48 B/sphere of LDS)
    bool useRayCache = true;
    bool useStash = true;  // regrouped hit processing through a per-wave hit stash (rt_kernels.h kStash); RT_STASH=0: off
    bool treeInLds = true;      // RT_TREE_LDS=0: the hierarchy's bounds are read through L2
    uint32_t treeTop = 128;  // largest top level the matrix-core filter takes (4 tiles of 32); RT_TREE_TOP overrides
    bool forceGlobal = false;

    // work order of the tiles (BuildTileOrder): valid for the running accumulation's strip
    bool useTileOrder = true;  // RT_TILE_ORDER=0 keeps the image order
    bool tileOrderValid = false;
    uint32_t tileW = 0, tileH = 0;  // ... and the image and strip it was built for (with the scene, all it depends on)
    rt_rowset tileRs{};
    DevBuf<float> pilotRays, pilotHits;
    DevBuf<uint32_t> tileOrder, matType;
    DevBuf<uint8_t> tileClass;

    // frame pipelining (rt_set_frame_pipelining; rt_params.h): regions of a sample ring, two continuation buffers
    uint32_t pipeDepth = 0;     // calls a path may be carried across (0 = off)
    bool pipeOpen = false;      // pipelined calls have been submitted since the last flush
    uint32_t pipeSeq = 0;       // sequence number of the newest region
    uint32_t pipeSppCap = 0, pipeNpix = 0, pipeRing = 0;  // geometry of the ring: samples per region, pixels, regions
    uint32_t pipeInSel = 0;     // which continuation buffer the next trace kernel reads
    uint32_t pipeCommits = 0;   // commit kernels launched since the pipeline started (FrameCtl: which entry is current)
    // frame batching (rt_set_frame_batch): stats-less rt_render calls wait here until batchFrames sample planes are pending
    uint32_t batchFrames = 1;
    bool pendOn = false;
    uint32_t pendW = 0, pendH = 0, pendS0 = 0, pendS1 = 0, pendDepth = 0;
    rt_rowset pendRs{};
    uint64_t pendSeed = 0;
    // render-ahead (rt_set_frame_lookahead): sample planes [aheadBase, aheadBase + aheadSpp) of the running accumulation sit in the
    // sample buffer, traced by ONE launch; planes below aheadNext have been added to the strip
    uint32_t lookahead = 1;
    bool aheadValid = false;
    uint32_t aheadW = 0, aheadH = 0, aheadBase = 0, aheadSpp = 0, aheadNext = 0, aheadDepth = 0;
    rt_rowset aheadRs{};
    uint64_t aheadSeed = 0;
    uint32_t pipeMaxDepth = 0;  // max_depth and seed of the running pipeline (a flush re-launches with them)
    uint64_t pipeSeed = 0;
    rtd::RegionTable pipeRegions{};
    DevBuf<float> ring;
    DevBuf<rtd::ContEntry> cont[2];
    DevBuf<uint32_t> contN[2];  // carried paths per wave
    DevBuf<rtd::FrameCtl> ctl;
};

static uint32_t RowsetLocalRows(rt_rowset rs) {
    if (rs.block_rows == 0 || rs.nshards == 0 || rs.shard >= rs.nshards) return 0;
    uint32_t rows = 0;
    const uint32_t nblocks = (rs.num_rows + rs.block_rows - 1) / rs.block_rows;
    for (uint32_t b = rs.shard; b < nblocks; b += rs.nshards) {
        const uint32_t r0 = b * rs.block_rows;
        const uint32_t left = rs.num_rows - r0;
        rows += left < rs.block_rows ? left : rs.block_rows;
    }
    return rows;
}

// ---------------------------------------------------------------------------------- scene layout
// Clustered storage for the scan (rt_scan.h): spheres split by a k-d median tree into groups of four, unusually
// large spheres alone, every group with a conservative bounding sphere for the matrix-core filter.
struct SceneLayout {
    std::vector<float4> scan;     // 4 * nGroups + 4 entries
    std::vector<uint32_t> orig;   // same length
    std::vector<float4> leaf;     // same length: conservative bound of each single sphere (sphere-level filter)
    std::vector<float4> tree;     // bounds of every level, level 0 (the groups) first
    uint32_t levelOff[rtd::kMaxLevels] = {0}, levelCnt[rtd::kMaxLevels] = {0};
    uint32_t nLevels = 1;         // level nLevels-1 is the top level (<= topMax nodes), filtered on the matrix cores
    uint32_t nGroups = 0;         // = levelCnt[0], a multiple of 4
    float boundNorm = 0.f;        // max |C| + R
    unsigned long long singleMask[2] = {0ull, 0ull};  // groups of one sphere, in the flat scan's bitmap coordinates
    uint32_t nAlways = 0;         // hierarchy scan: leading big-sphere groups kept out of the hierarchy (tested for every ray)
    float treeBox[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // box around the spheres in the hierarchy (lo, hi, max |coordinate|)
    bool treeBoxOn = false;
    // cell-grid scan (rt_scan.h scan_list_grid): the small spheres sorted by home cell behind the big ones
    bool gridOn = false;
    std::vector<uint16_t> gridCellStart;  // [nu * nv + 1]
    uint32_t gridNu = 0, gridNv = 0, gridAxU = 0, gridAxV = 2;
    float gridG0u = 0.f, gridG0v = 0.f, gridInvH = 0.f, gridRmaxOverH = 0.f, gridBigNorm = 0.f;
    std::vector<uint32_t> gridQ;  // quantised one-sphere bounds per scan entry (rt_scan.h GridQuant), empty: none
    float gridQc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// Conservative bounding sphere of a set of spheres, in the filter's (C, |C|^2 - Rf^2) form.
static float4 BoundOf(const rt_sphere* sp, const std::vector<uint32_t>& ids, float* normOut, float marginK) {
    if (ids.empty()) return make_float4(0.f, 0.f, 0.f, 1e30f);  // never a candidate
    const double kEps = (double)marginK * 5.9604644775390625e-08;  // K * eps: K is that of the unit testing this bound (rt_scan.h)
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t k : ids) {
        const double c[3] = {sp[k].cx, sp[k].cy, sp[k].cz};
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], c[a] - (double)sp[k].r);
            hi[a] = std::max(hi[a], c[a] + (double)sp[k].r);
        }
    }
    // centre: the enclosing radius max_i(|c_i - C| + r_i) is convex in C, so a pattern search from the box centre
    // (axis steps, halved when none improves) finds the near-minimal enclosing sphere, typically 5-15 % smaller than the
    // box-centred one.  Any centre is valid: R below is measured from the centre actually used.  The bound centre is the
    // FLOAT the device will use, so its rounding is inside s_i below.
    double cc[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
    if (ids.size() > 1) {
        auto reachOf = [&](const double c[3]) {
            double far = 0.0;
            for (uint32_t k : ids) {
                const double dx = sp[k].cx - c[0], dy = sp[k].cy - c[1], dz = sp[k].cz - c[2];
                far = std::max(far, std::sqrt(dx * dx + dy * dy + dz * dz) + (double)sp[k].r);
            }
            return far;
        };
        double best = reachOf(cc);
        double step = 0.25 * best;
        for (int it = 0; it < 400 && step > 1e-7 * best; ++it) {
            bool improved = false;
            for (int ax = 0; ax < 3; ++ax)
                for (int sgn = -1; sgn <= 1; sgn += 2) {
                    double t[3] = {cc[0], cc[1], cc[2]};
                    t[ax] += sgn * step;
                    const double r = reachOf(t);
                    if (r < best) {
                        best = r;
                        cc[0] = t[0]; cc[1] = t[1]; cc[2] = t[2];
                        improved = true;
                    }
                }
            if (!improved) step *= 0.5;
        }
    }
    const float Cf[3] = {(float)cc[0], (float)cc[1], (float)cc[2]};
    double R = 0, smax = 0;
    for (uint32_t k : ids) {
        const double dx = sp[k].cx - (double)Cf[0], dy = sp[k].cy - (double)Cf[1], dz = sp[k].cz - (double)Cf[2];
        const double si = std::sqrt(dx * dx + dy * dy + dz * dz);
        smax = std::max(smax, si);
        R = std::max(R, si + (double)sp[k].r);
    }
    const double C2 = (double)Cf[0] * Cf[0] + (double)Cf[1] * Cf[1] + (double)Cf[2] * Cf[2];
    const double Cn = std::sqrt(C2);
    const double Rf2 = R * R * (1.0 + 1e-5) + 0.01 * smax * smax + kEps * (2.0 * (Cn + R) * (Cn + R) + R * R);
    float w = (float)(C2 - Rf2);
    w = std::nextafterf(std::nextafterf(w, -INFINITY), -INFINITY);  // err towards "more candidates"
    if (normOut) *normOut = std::max(*normOut, (float)((Cn + R) * 1.001));
    return make_float4(Cf[0], Cf[1], Cf[2], w);
}

// Enclosing radius of a set of spheres about its near-optimal centre (the same pattern search BoundOf uses).
static double EnclosingRadius(const rt_sphere* sp, const uint32_t* ids, size_t n) {
    if (n == 0) return 0.0;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t q = 0; q < n; ++q) {
        const rt_sphere& s = sp[ids[q]];
        const double c[3] = {s.cx, s.cy, s.cz};
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], c[a] - (double)s.r);
            hi[a] = std::max(hi[a], c[a] + (double)s.r);
        }
    }
    double cc[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
    auto reachOf = [&](const double c[3]) {
        double far = 0.0;
        for (size_t q = 0; q < n; ++q) {
            const rt_sphere& s = sp[ids[q]];
            const double dx = s.cx - c[0], dy = s.cy - c[1], dz = s.cz - c[2];
            far = std::max(far, std::sqrt(dx * dx + dy * dy + dz * dz) + (double)s.r);
        }
        return far;
    };
    double best = reachOf(cc);
    if (n == 1) return best;
    double step = 0.25 * best;
    for (int it = 0; it < 60 && step > 1e-4 * best; ++it) {
        bool improved = false;
        for (int ax = 0; ax < 3; ++ax)
            for (int sgn = -1; sgn <= 1; sgn += 2) {
                double t[3] = {cc[0], cc[1], cc[2]};
                t[ax] += sgn * step;
                const double r = reachOf(t);
                if (r < best) { best = r; cc[0] = t[0]; cc[1] = t[1]; cc[2] = t[2]; improved = true; }
            }
        if (!improved) step *= 0.5;
    }
    return best;
}

// Local refinement of the k-d groups: for pairs of spatially neighbouring groups, redistribute their (at most eight)
// members into two groups of the same sizes when that lowers R1^2 + R2^2 (the filter's candidate count per ray is
// proportional to the summed squared bound radii).  Groups keep their positions in the list, so the hierarchy above
// them (consecutive quadruples) stays spatially coherent.  Skipped for very large scenes (upload time).
static void RefineGroups(const rt_sphere* sp, std::vector<std::vector<uint32_t>>& groups) {
    const size_t G = groups.size();
    if (G < 2 || G > 4096) return;
    std::vector<double> R(G, 0.0);
    std::vector<std::array<double, 3>> C(G);
    auto update = [&](size_t g) {
        R[g] = EnclosingRadius(sp, groups[g].data(), groups[g].size());
        double c[3] = {0, 0, 0};
        for (uint32_t k : groups[g]) { c[0] += sp[k].cx; c[1] += sp[k].cy; c[2] += sp[k].cz; }
        const double inv = groups[g].empty() ? 0.0 : 1.0 / (double)groups[g].size();
        C[g] = {c[0] * inv, c[1] * inv, c[2] * inv};
    };
    for (size_t g = 0; g < G; ++g) update(g);
    for (int sweep = 0; sweep < 4; ++sweep) {
        bool any = false;
        for (size_t g = 0; g < G; ++g) {
            if (groups[g].size() < 2) continue;  // singletons (big spheres) and padding stay as they are
            for (size_t h = g + 1; h < G; ++h) {
                if (groups[h].size() < 2) continue;
                const double dx = C[g][0] - C[h][0], dy = C[g][1] - C[h][1], dz = C[g][2] - C[h][2];
                const double reachSum = R[g] + R[h];
                if (dx * dx + dy * dy + dz * dz > reachSum * reachSum) continue;  // bounds do not even touch
                const size_t ng = groups[g].size(), nh = groups[h].size(), nt = ng + nh;
                uint32_t all[8];
                for (size_t q = 0; q < ng; ++q) all[q] = groups[g][q];
                for (size_t q = 0; q < nh; ++q) all[ng + q] = groups[h][q];
                double bestCost = R[g] * R[g] + R[h] * R[h];
                uint32_t bestMask = 0;
                for (uint32_t mask = 1; mask < (1u << nt); ++mask) {
                    if ((size_t)__builtin_popcount(mask) != ng || !(mask & 1u)) continue;  // member 0 stays in g: no mirror splits
                    uint32_t a[8], b[8];
                    size_t na = 0, nb = 0;
                    for (size_t q = 0; q < nt; ++q) ((mask >> q) & 1u ? a[na++] : b[nb++]) = all[q];
                    const double ra = EnclosingRadius(sp, a, na);
                    if (ra * ra >= bestCost) continue;
                    const double rb = EnclosingRadius(sp, b, nb);
                    const double cost = ra * ra + rb * rb;
                    if (cost < bestCost * (1.0 - 1e-9)) { bestCost = cost; bestMask = mask; }
                }
                if (bestMask != 0 && bestMask != ((1u << ng) - 1u)) {
                    std::vector<uint32_t> a, b;
                    for (size_t q = 0; q < nt; ++q) ((bestMask >> q) & 1u ? a : b).push_back(all[q]);
                    groups[g] = a;
                    groups[h] = b;
                    update(g);
                    update(h);
                    any = true;
                }
            }
        }
        if (!any) break;
    }
}

// Box around a set of spheres (radii included) and the constants of the per-ray padding (rt_scan.h): treeBox[0..5] = lo, hi,
// [6] = max |coordinate|, [7] = 3 A^2 (A = max |c| + r), [8] = 1 / (2 r_min).
static bool BoxOf(const rt_sphere* sp, const std::vector<uint32_t>& ids, float treeBox[9]) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    double A = 0.0, rmin = 1e300;
    for (uint32_t k : ids) {
        const double c[3] = {sp[k].cx, sp[k].cy, sp[k].cz};
        A = std::max(A, std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) + (double)sp[k].r);
        rmin = std::min(rmin, (double)sp[k].r);
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], c[a] - (double)sp[k].r * (1.0 + 1e-5));
            hi[a] = std::max(hi[a], c[a] + (double)sp[k].r * (1.0 + 1e-5));
        }
    }
    if (!(lo[0] <= hi[0])) return false;
    double am = 0;
    for (int a = 0; a < 3; ++a) {
        treeBox[a] = std::nextafterf((float)lo[a], -INFINITY);
        treeBox[3 + a] = std::nextafterf((float)hi[a], INFINITY);
        am = std::max(am, std::max(std::fabs(lo[a]), std::fabs(hi[a])));
    }
    treeBox[6] = (float)(am * 1.001);
    treeBox[7] = (float)(3.0 * A * A * 1.001);
    treeBox[8] = (float)(1.001 / (2.0 * rmin));
    return true;
}

// Cell-grid layout (rt_scan.h scan_list_grid) for a scene of at most eight big spheres and a layer of many small ones: the big
// spheres keep the leading one-sphere groups (entries 4q; every ray tests them exactly), the small spheres follow SORTED BY
// HOME CELL of a uniform grid over the two long axes of their box -- cell = iu * nv + iv, so a run of cells of one u-slab is a
// run of scan entries -- about one sphere per cell, at most 254 x 254 cells.  false: the scene does not suit (the caller builds
// the bounds hierarchy instead).
static bool BuildGridLayout(const rt_sphere* sp, uint32_t n, const std::vector<uint32_t>& big, const std::vector<uint32_t>& small, SceneLayout& L) {
    if (big.size() > 8 || small.size() < 256) return false;
    if (!BoxOf(sp, small, L.treeBox)) return false;
    const double ext[3] = {(double)L.treeBox[3] - L.treeBox[0], (double)L.treeBox[4] - L.treeBox[1], (double)L.treeBox[5] - L.treeBox[2]};
    int w = 0;
    if (ext[1] < ext[w]) w = 1;
    if (ext[2] < ext[w]) w = 2;
    const int axU = (w + 1) % 3, axV = (w + 2) % 3;
    double rmax = 0;
    for (uint32_t k : small) rmax = std::max(rmax, (double)sp[k].r);
    double density = 1.0;  // spheres per cell aimed at (measured on grid10k: 0.5 -5 %, 0.75 -4 %, 1.5 -4 %, 2.5 -8 %, 4 -12 %; RT_GRID_DENSITY: experiments)
    if (const char* e = std::getenv("RT_GRID_DENSITY")) density = std::max(0.1, std::atof(e));
    double h = std::sqrt(std::max(ext[axU] * ext[axV], 1e-30) * density / (double)small.size());
    h = std::max(h, 2.5 * rmax);                                  // a sphere's neighbourhood stays within one cell of its home
    h = std::max(h, std::max(ext[axU], ext[axV]) / 250.0);         // at most 254 cells per axis
    const float g0u = L.treeBox[axU] - (float)(0.01 * h), g0v = L.treeBox[axV] - (float)(0.01 * h);
    const float invH = (float)(1.0 / h);
    const uint32_t nu = (uint32_t)std::floor(((double)L.treeBox[3 + axU] - g0u) * invH) + 2u;
    const uint32_t nv = (uint32_t)std::floor(((double)L.treeBox[3 + axV] - g0v) * invH) + 2u;
    if (nu > 254u || nv > 254u || (size_t)nu * nv > 60000u) return false;
    auto coord = [&](uint32_t k, int ax) { return ax == 0 ? sp[k].cx : (ax == 1 ? sp[k].cy : sp[k].cz); };
    std::vector<std::pair<uint32_t, uint32_t>> keyed;  // (home cell, sphere)
    keyed.reserve(small.size());
    for (uint32_t k : small) {
        // the device forms (x - g0) * invH in float; the host's double value differs by < 1e-4 cells, inside the walk's slack
        const double fu = ((double)coord(k, axU) - (double)g0u) * (double)invH, fv = ((double)coord(k, axV) - (double)g0v) * (double)invH;
        const uint32_t iu = (uint32_t)std::min(std::max(std::floor(fu), 0.0), (double)(nu - 1u));
        const uint32_t iv = (uint32_t)std::min(std::max(std::floor(fv), 0.0), (double)(nv - 1u));
        keyed.push_back({iu * nv + iv, k});
    }
    std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) { return a.first < b.first; });
    {   // a uniform grid suits a uniform layer: when the spheres are clumped (the average sphere shares its cell with more than
        // five others; a Poisson layer at one sphere per cell has one) a ray through a clump would test hundreds of spheres per
        // cell -- measured 3x slower than the bounds hierarchy on 1,500 spheres in a 3 x 3 patch of a 200 x 200 layer -- so the
        // hierarchy takes such scenes
        double sumSq = 0.0;
        for (size_t q = 0; q < keyed.size();) {
            size_t e = q;
            while (e < keyed.size() && keyed[e].first == keyed[q].first) ++e;
            sumSq += (double)(e - q) * (double)(e - q);
            q = e;
        }
        if (sumSq / (double)keyed.size() > 6.0) return false;
    }
    std::vector<std::vector<uint32_t>> groups;
    for (uint32_t k : big) groups.push_back({k});
    while (groups.size() & 3u) groups.push_back({});
    const uint32_t base = (uint32_t)groups.size() * 4u;  // first entry of the sorted small spheres
    for (size_t q = 0; q < keyed.size(); q += 4) {
        std::vector<uint32_t> g;
        for (size_t m = q; m < std::min(q + 4, keyed.size()); ++m) g.push_back(keyed[m].second);
        groups.push_back(g);
    }
    while (groups.size() & 3u) groups.push_back({});
    if (groups.size() * 4 + 4 >= 65536) return false;
    L.nGroups = (uint32_t)groups.size();
    const float4 never = make_float4(0.f, 0.f, 0.f, -1e30f);
    L.scan.assign((size_t)L.nGroups * 4 + 4, never);
    L.orig.assign((size_t)L.nGroups * 4 + 4, 0xffffffffu);
    for (uint32_t gi = 0; gi < L.nGroups; ++gi)
        for (size_t m = 0; m < groups[gi].size(); ++m) {
            const uint32_t k = groups[gi][m];
            L.scan[(size_t)gi * 4 + m] = make_float4(sp[k].cx, sp[k].cy, sp[k].cz, sp[k].r * sp[k].r);
            L.orig[(size_t)gi * 4 + m] = k;
        }
    L.gridCellStart.assign((size_t)nu * nv + 1, 0);
    {
        size_t q = 0;
        for (uint32_t c = 0; c <= nu * nv; ++c) {
            while (q < keyed.size() && keyed[q].first < c) ++q;
            L.gridCellStart[c] = (uint16_t)(base + q);
        }
    }
    L.singleMask[0] = L.singleMask[1] = 0ull;
    L.leaf.assign(L.scan.size(), BoundOf(sp, {}, nullptr, rtd::kMarginKLeaf));
    L.boundNorm = 0.f;
    for (size_t e = 0; e < L.orig.size(); ++e)
        if (L.orig[e] != 0xffffffffu) L.leaf[e] = BoundOf(sp, {L.orig[e]}, e >= base ? &L.boundNorm : &L.gridBigNorm, rtd::kMarginKLeaf);
    L.nAlways = (uint32_t)big.size();
    // one level of group bounds for rt_unit_layout's readers (the scan does not use them); the big spheres' groups are out of it
    L.tree.clear();
    L.nLevels = 1;
    L.levelOff[0] = 0;
    L.levelCnt[0] = L.nGroups;
    for (uint32_t gi = 0; gi < L.nGroups; ++gi) {
        float norm = 0.f;
        L.tree.push_back(gi < L.nAlways ? BoundOf(sp, {}, nullptr, rtd::kMarginKValu) : BoundOf(sp, groups[gi], &norm, rtd::kMarginKValu));
    }
    L.treeBoxOn = true;
    L.gridOn = true;
    L.gridNu = nu; L.gridNv = nv; L.gridAxU = (uint32_t)axU; L.gridAxV = (uint32_t)axV;
    L.gridG0u = g0u; L.gridG0v = g0v; L.gridInvH = invH; L.gridRmaxOverH = (float)(rmax * (double)invH * 1.0001);
    {   // Quantised bounds (rt_scan.h GridQuant): the device's four fmas, evaluated here with fmaf on the same constants, give the
        // bound centre C' bit for bit; the radius class covers r_i plus the sphere's own offset |C' - c_i|.
        const float hF = (float)h, su = hF / 256.f;
        const float uBase0 = g0u + 0.5f * su, vBase = g0v + 0.5f * su;
        double wlo = 1e300, whi = -1e300;
        for (uint32_t k : small) {
            wlo = std::min(wlo, (double)coord(k, w));
            whi = std::max(whi, (double)coord(k, w));
        }
        const float wstep = (float)((whi - wlo) / 16.0 * 1.0001 + 1e-30), wBase = (float)wlo + 0.5f * wstep;
        struct Q1 { uint32_t du, v16, kw; double err, r; };
        std::vector<Q1> q1(keyed.size());
        double smax = 0.0, Rmax = 0.0;
        bool ok = true;
        for (size_t q = 0; q < keyed.size() && ok; ++q) {
            const uint32_t k = keyed[q].second, cell = keyed[q].first, iu = cell / nv, iv = cell % nv;
            const double fu = ((double)coord(k, axU) - (double)g0u) * (double)invH - (double)iu;  // position in the home cell, [0, 1) unless clamped
            const double fv = ((double)coord(k, axV) - (double)g0v) * (double)invH - (double)iv;
            const uint32_t du = (uint32_t)std::min(255.0, std::max(0.0, std::floor(fu * 256.0)));
            const uint32_t dv = (uint32_t)std::min(255.0, std::max(0.0, std::floor(fv * 256.0)));
            const uint32_t kw = (uint32_t)std::min(15.0, std::max(0.0, std::floor(((double)coord(k, w) - wlo) / std::max((double)wstep, 1e-30))));
            const uint32_t v16 = iv * 256u + dv;
            const float uBase = std::fmaf((float)iu, hF, uBase0);
            const float cu = std::fmaf((float)du, su, uBase), cv = std::fmaf((float)v16, su, vBase), cw = std::fmaf((float)kw, wstep, wBase);
            const double eu = (double)cu - coord(k, axU), ev = (double)cv - coord(k, axV), ew = (double)cw - coord(k, w);
            const double err = std::sqrt(eu * eu + ev * ev + ew * ew);
            q1[q] = {du, v16, kw, err, (double)sp[k].r};
            smax = std::max(smax, err);
            Rmax = std::max(Rmax, (double)sp[k].r + err);
        }
        // (a sphere clamped into an edge cell can sit far from its cell: then the classes would be too coarse to be of use)
        if (ok && smax > 0.02 * h + 0.5 * (double)wstep) ok = false;
        if (ok) {
            const float rstep = (float)(Rmax * (1.0 + 1e-5) / 16.0);
            L.gridQ.assign(L.scan.size(), 0u);
            for (size_t q = 0; q < keyed.size(); ++q) {
                const double need = (q1[q].r + q1[q].err) * (1.0 + 1e-6) + 1e-30;
                uint32_t kr = 0;
                while (kr < 15u && (double)std::fmaf((float)kr, rstep, rstep) < need) ++kr;
                if ((double)std::fmaf((float)kr, rstep, rstep) < need) { ok = false; break; }
                L.gridQ[base + q] = q1[q].du | q1[q].v16 << 8 | q1[q].kw << 24 | kr << 28;
            }
            const float s2 = (float)(smax * smax * (1.0 + 1e-5) + 1e-30);
            const float qc[8] = {su, uBase0, hF, vBase, wBase, wstep, rstep, s2};
            for (int c = 0; c < 8; ++c) L.gridQc[c] = qc[c];
        }
        if (!ok) L.gridQ.clear();
    }
    return true;
}

static void BuildLayout(const rt_sphere* sp, uint32_t n, uint32_t topMax, SceneLayout& L) {
    std::vector<float> radii(n);
    for (uint32_t k = 0; k < n; ++k) radii[k] = sp[k].r;
    std::vector<float> sorted = radii;
    std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
    const float median = sorted[n / 2];
    std::vector<uint32_t> big, small;
    for (uint32_t k = 0; k < n; ++k) (radii[k] > 4.f * median ? big : small).push_back(k);
    // scenes beyond the flat matrix-core filter (more than topMax groups of four): a cell grid over the layer of small spheres
    // when the scene suits it (RT_GRID=0: always the bounds hierarchy)
    {
        const char* g = std::getenv("RT_GRID");
        const bool wantGrid = !(g && g[0] == '0') && std::getenv("RT_ALWAYS_BIG") == nullptr;
        const bool force = g && g[0] == '2';  // RT_GRID=2 (experiments): the grid for every scene it can be built for
        if (wantGrid && ((small.size() + 3) / 4 + big.size() > topMax || force) && BuildGridLayout(sp, n, big, small, L)) return;
        L = SceneLayout{};
    }
    // k-d median split down to leaves of four, emitted in tree order: compact, balanced groups whose
    // neighbours in the list are neighbours in space (Morton chunks of a jittered grid have 3x the summed R^2
    // and twice the filter candidates; tools/cluster_eval.py)
    std::vector<std::vector<uint32_t>> groups;
    for (uint32_t k : big) groups.push_back({k});
    while (!big.empty() && (groups.size() & 3u)) groups.push_back({});  // big spheres keep upper-level nodes of their own
    std::vector<std::pair<size_t, size_t>> stack;  // [begin, end) ranges of `small`
    if (!small.empty()) stack.push_back({0, small.size()});
    while (!stack.empty()) {
        const auto [b, e] = stack.back();
        stack.pop_back();
        if (e - b <= 4) {
            groups.push_back(std::vector<uint32_t>(small.begin() + b, small.begin() + e));
            continue;
        }
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t k = b; k < e; ++k) {
            const float c[3] = {sp[small[k]].cx, sp[small[k]].cy, sp[small[k]].cz};
            for (int a = 0; a < 3; ++a) {
                lo[a] = std::min(lo[a], c[a]);
                hi[a] = std::max(hi[a], c[a]);
            }
        }
        int ax = 0;
        if (hi[1] - lo[1] > hi[ax] - lo[ax]) ax = 1;
        if (hi[2] - lo[2] > hi[ax] - lo[ax]) ax = 2;
        std::stable_sort(small.begin() + b, small.begin() + e, [&](uint32_t x, uint32_t y) {
            const float cx[3] = {sp[x].cx, sp[x].cy, sp[x].cz}, cy[3] = {sp[y].cx, sp[y].cy, sp[y].cz};
            return cx[ax] < cy[ax];
        });
        // left part: the largest power-of-four multiple of 4 not above half, so whole subtrees stay aligned
        size_t half = ((e - b) / 2 + 3) / 4 * 4;
        size_t p4 = 4;
        while (p4 * 4 <= (e - b) / 2 + 3) p4 *= 4;
        if (p4 >= 16 && (e - b) > p4) half = std::max(p4, (size_t)(((e - b) / 2) / p4 * p4));
        if (half >= e - b) half = (e - b) / 2;
        stack.push_back({b + half, e});
        stack.push_back({b, b + half});
    }
    RefineGroups(sp, groups);
    while (groups.size() & 3u) groups.push_back({});  // whole nodes at the next level; also even for the VALU scan
    L.nGroups = (uint32_t)groups.size();
    const float4 never = make_float4(0.f, 0.f, 0.f, -1e30f);  // r*r = -1e30: discriminant negative for any ray
    L.scan.assign((size_t)L.nGroups * 4 + 4, never);
    L.orig.assign((size_t)L.nGroups * 4 + 4, 0xffffffffu);
    for (uint32_t gi = 0; gi < L.nGroups; ++gi) {
        for (size_t m = 0; m < groups[gi].size(); ++m) {
            const uint32_t k = groups[gi][m];
            // radius * radius is the float product Sphere::Intersect forms per call (ray-tracing.cpp:48)
            L.scan[(size_t)gi * 4 + m] = make_float4(sp[k].cx, sp[k].cy, sp[k].cz, sp[k].r * sp[k].r);
            L.orig[(size_t)gi * 4 + m] = k;
        }
    }
    // groups of one sphere (the big ones, mostly) as bitmap bits: bit 63 - N of half h is group 16 h + (N & 15) + 32 (N >> 4)
    // (rt_scan.h, next_candidate); only meaningful while the groups ARE the filter's top level (<= 128 of them)
    L.singleMask[0] = L.singleMask[1] = 0ull;
    if (L.nGroups <= 128u)
        for (uint32_t gi = 0; gi < L.nGroups; ++gi)
            if (groups[gi].size() == 1) {
                const uint32_t N = (gi & 15u) + 16u * (gi >> 5);
                L.singleMask[(gi >> 4) & 1u] |= 0x8000000000000000ull >> N;
            }
    L.leaf.assign(L.scan.size(), BoundOf(sp, {}, nullptr, rtd::kMarginKLeaf));  // padding entries are never candidates
    for (size_t e = 0; e < L.orig.size(); ++e)
        if (L.orig[e] != 0xffffffffu) L.leaf[e] = BoundOf(sp, {L.orig[e]}, nullptr, rtd::kMarginKLeaf);
    // levels: level 0 = the groups; level k+1 node j = level-k nodes 4j .. 4j+3; stop at <= topMax nodes.  The top
    // level is tested by the matrix-core filter (margin K = kMarginK), the levels below it on the VALU (kMarginKValu).
    std::vector<std::vector<std::vector<uint32_t>>> levels;
    levels.push_back(groups);
    // Hierarchy scan (more groups than the matrix-core level takes): the big spheres stay out of the bounds.  With the floor
    // inside, node 0 of every level is a candidate for every ray and drags its siblings into the descent; tested directly, a
    // big sphere costs one exact slot per live ray.  (At most eight; the flat scan keeps them as one-sphere groups.)
    L.nAlways = 0;
    if (groups.size() > topMax && std::getenv("RT_ALWAYS_BIG") == nullptr) {
        while (L.nAlways < 8u && L.nAlways < big.size() && groups[L.nAlways].size() == 1) {
            levels[0][L.nAlways].clear();
            ++L.nAlways;
        }
    }
    while (levels.back().size() > topMax && levels.size() < rtd::kMaxLevels) {
        std::vector<std::vector<uint32_t>>& cur = levels.back();
        while (cur.size() & 3u) cur.push_back({});  // pad this level to whole parents
        std::vector<std::vector<uint32_t>> up(cur.size() / 4);
        for (size_t j = 0; j < up.size(); ++j)
            for (int q = 0; q < 4; ++q) up[j].insert(up[j].end(), cur[4 * j + q].begin(), cur[4 * j + q].end());
        levels.push_back(std::move(up));
    }
    L.treeBoxOn = false;
    if (levels.size() > 1 && std::getenv("RT_TREE_BOX_OFF") == nullptr) {
        std::vector<uint32_t> inTree;
        for (const auto& ids : levels[0]) inTree.insert(inTree.end(), ids.begin(), ids.end());
        L.treeBoxOn = BoxOf(sp, inTree, L.treeBox);
    }
    L.tree.clear();
    L.nLevels = (uint32_t)levels.size();
    for (uint32_t lvl = 0; lvl < L.nLevels; ++lvl) {
        L.levelOff[lvl] = (uint32_t)L.tree.size();
        L.levelCnt[lvl] = (uint32_t)levels[lvl].size();
        const float K = lvl + 1 == L.nLevels ? rtd::kMarginK : rtd::kMarginKValu;
        for (const auto& ids : levels[lvl]) L.tree.push_back(BoundOf(sp, ids, &L.boundNorm, K));
    }
}
// ---------------------------------------------------------------------------------- shadow index
// Footprints of the spheres in the plane perpendicular to the sun, binned into a uniform grid (rt_shade.h
// shadow_query).  Conservative by construction: footprint radius rho = sqrt(r^2 + 64 eps (2 P0^2 + 2|c|^2 + r^2))
// (1 + 1e-4) + 1e-5 (P0 + |c| + 1) covers the reference test's own rounding for hit points with |p| <= P0 (E/a <= 16
// eps (...), 4x safety) and the rounding of the float projection; a sphere is listed in every cell its footprint's
// bounding square touches.
struct ShadowGrid {
    std::vector<uint16_t> cellStart, entries, global;
    uint32_t nx = 0, ny = 0;
    float e1[3] = {0, 0, 0}, e2[3] = {0, 0, 0}, u0 = 0, v0 = 0, invCell = 0, p0sq = 0;
    bool enabled = false;
};

// maxCells: cells per axis at most.  64 for the scenes whose index is staged into LDS next to the tables (the flat scan: 8-10 KB on
// the cover scene); 256 for the scenes whose index stays in global memory (cell-grid and hierarchy scans, 10,000-sphere class): at
// 64 x 64 a query of grid10k walked ~15 spheres -- 7 rounds of two dependent L2 reads -- at 256 x 256 (cells of about one
// footprint) ~5.
static void BuildShadowGrid(const rt_sphere* sp, const SceneLayout& L, const float sunDir[3], uint32_t maxCells, ShadowGrid& G) {
    G = ShadowGrid{};
    const double Lx = sunDir[0], Ly = sunDir[1], Lz = sunDir[2];
    const double ln = std::sqrt(Lx * Lx + Ly * Ly + Lz * Lz);
    if (!(ln > 0.5 && ln < 2.0)) return;  // not a direction: keep the scan
    // orthonormal basis of the plane perpendicular to L
    double ax[3] = {1, 0, 0};
    if (std::fabs(Lx) > std::fabs(Ly) && std::fabs(Lx) > std::fabs(Lz)) { ax[0] = 0; ax[1] = 1; }
    double e1[3] = {Ly * ax[2] - Lz * ax[1], Lz * ax[0] - Lx * ax[2], Lx * ax[1] - Ly * ax[0]};
    const double n1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    for (double& v : e1) v /= n1;
    double e2[3] = {(Ly * e1[2] - Lz * e1[1]) / ln, (Lz * e1[0] - Lx * e1[2]) / ln, (Lx * e1[1] - Ly * e1[0]) / ln};
    for (int k = 0; k < 3; ++k) {
        G.e1[k] = (float)e1[k];
        G.e2[k] = (float)e2[k];
    }
    const size_t nEnt = (size_t)L.nGroups * 4;
    // P0: twice the reach of the ordinary (non-huge) spheres, so that practically every hit point qualifies
    std::vector<double> rs;
    for (size_t e = 0; e < nEnt; ++e)
        if (L.orig[e] != 0xffffffffu) rs.push_back(sp[L.orig[e]].r);
    if (rs.empty()) return;
    std::nth_element(rs.begin(), rs.begin() + rs.size() / 2, rs.end());
    const double med = rs[rs.size() / 2];
    double reach = 0;
    for (size_t e = 0; e < nEnt; ++e) {
        if (L.orig[e] == 0xffffffffu) continue;
        const rt_sphere& q = sp[L.orig[e]];
        if (q.r > 4.0 * med) continue;
        reach = std::max(reach, std::sqrt((double)q.cx * q.cx + (double)q.cy * q.cy + (double)q.cz * q.cz) + q.r);
    }
    const double P0 = 2.0 * reach + 8.0 * med + 1.0;
    const double eps = 5.9604644775390625e-08;
    struct Foot { double u, v, rho; uint16_t entry; };
    std::vector<Foot> feet;
    double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
    std::vector<double> rhos;
    for (size_t e = 0; e < nEnt; ++e) {
        if (L.orig[e] == 0xffffffffu) continue;
        const rt_sphere& q = sp[L.orig[e]];
        // the device projects with the FLOAT basis; use the same vectors here
        const double u = q.cx * (double)G.e1[0] + q.cy * (double)G.e1[1] + q.cz * (double)G.e1[2];
        const double v = q.cx * (double)G.e2[0] + q.cy * (double)G.e2[1] + q.cz * (double)G.e2[2];
        const double cn = std::sqrt((double)q.cx * q.cx + (double)q.cy * q.cy + (double)q.cz * q.cz);
        const double rho = std::sqrt((double)q.r * q.r + 64.0 * eps * (2.0 * P0 * P0 + 2.0 * cn * cn + (double)q.r * q.r)) * (1.0 + 1e-4) +
                           1e-5 * (P0 + cn + 1.0);
        feet.push_back({u, v, rho, (uint16_t)e});
        if (q.r <= 4.0 * med) {
            lo[0] = std::min(lo[0], u - rho); hi[0] = std::max(hi[0], u + rho);
            lo[1] = std::min(lo[1], v - rho); hi[1] = std::max(hi[1], v + rho);
            rhos.push_back(rho);
        }
    }
    if (rhos.empty()) {  // only huge spheres: everything goes to the global list, a 1x1 grid
        lo[0] = lo[1] = -1.0;
        hi[0] = hi[1] = 1.0;
        rhos.push_back(1.0);
    }
    std::nth_element(rhos.begin(), rhos.begin() + rhos.size() / 2, rhos.end());
    const double ext = std::max(hi[0] - lo[0], hi[1] - lo[1]);
    double cell = std::max(2.0 * rhos[rhos.size() / 2], ext / (double)maxCells);
    G.nx = (uint32_t)std::min((double)maxCells, std::max(1.0, std::ceil((hi[0] - lo[0]) / cell)));
    G.ny = (uint32_t)std::min((double)maxCells, std::max(1.0, std::ceil((hi[1] - lo[1]) / cell)));
    G.u0 = (float)lo[0];
    G.v0 = (float)lo[1];
    G.invCell = (float)(1.0 / cell);
    // cell of a coordinate exactly as the device computes it (float), widened by one ulp-ish slack through rho
    auto cellOf = [&](double x, float x0, uint32_t n) -> long {
        const double f = (x - (double)x0) * (double)G.invCell;
        return (long)std::floor(f);
    };
    std::vector<std::vector<uint16_t>> cells((size_t)G.nx * G.ny);
    const size_t ncell = cells.size();
    for (const Foot& f : feet) {
        // no extra cell of slack: rho already carries 1e-5 (P0 + |c| + 1), at least 20x the error of the device's float
        // projection and cell arithmetic (<= ~5e-7 P0 for |p| <= P0), and floor() is monotone
        long x0 = cellOf(f.u - f.rho, G.u0, G.nx), x1 = cellOf(f.u + f.rho, G.u0, G.nx);
        long y0 = cellOf(f.v - f.rho, G.v0, G.ny), y1 = cellOf(f.v + f.rho, G.v0, G.ny);
        const bool outside = x1 < 0 || y1 < 0 || x0 >= (long)G.nx || y0 >= (long)G.ny;
        x0 = std::max(0L, x0); y0 = std::max(0L, y0);
        x1 = std::min((long)G.nx - 1, x1); y1 = std::min((long)G.ny - 1, y1);
        const bool spills = (f.u - f.rho < lo[0]) || (f.u + f.rho > hi[0]) || (f.v - f.rho < lo[1]) || (f.v + f.rho > hi[1]);
        const size_t covered = outside ? 0 : (size_t)(x1 - x0 + 1) * (size_t)(y1 - y0 + 1);
        // a footprint reaching beyond the grid can shadow points outside it: it must be tested for every query
        if (spills || covered * 8 > ncell) {
            G.global.push_back(f.entry);
            continue;
        }
        for (long y = y0; y <= y1; ++y)
            for (long x = x0; x <= x1; ++x) cells[(size_t)y * G.nx + x].push_back(f.entry);
    }
    size_t total = 0;
    for (const auto& c : cells) total += c.size();
    if (total >= 65535 && maxCells > 64u) {  // too many entries for 16-bit cell starts: the coarser grid
        BuildShadowGrid(sp, L, sunDir, maxCells / 2u, G);
        return;
    }
    if (total >= 65535 || G.global.size() > 64) return;  // pathological: keep the scan
    G.cellStart.resize(ncell + 1);
    G.entries.reserve(total);
    for (size_t c = 0; c < ncell; ++c) {
        G.cellStart[c] = (uint16_t)G.entries.size();
        G.entries.insert(G.entries.end(), cells[c].begin(), cells[c].end());
    }
    G.cellStart[ncell] = (uint16_t)G.entries.size();
    G.p0sq = (float)(P0 * P0 * (1.0 - 1e-6));
    G.enabled = true;
}

static size_t LdsBytesFor(uint32_t n, uint32_t nPadded, bool mats) {
    (void)n;  // radius and material tables are stored per scan entry (clustered order), like the scan table
    return (size_t)nPadded * (16 + 4) + (mats ? (size_t)nPadded * 48 : 0) + (size_t)nPadded * 4;
}
static size_t MfmaOpsBytesFor(uint32_t nGroups) { return (size_t)rtd::mfma_tiles_for(nGroups) * rtd::kOpsPerTile * 4; }

// Which instantiation of rt_trace_kernel a scene gets under this context's settings (also used by the closest-hit unit
// entry, so that it runs the scan rt_render runs): tree = hierarchy scan <false, ., 2>, flat = matrix-core filter over the
// groups with every table in LDS <true, ., 1>, else the VALU scan with (ldsTables) or without LDS tables.
struct TraceVariant {
    bool tree, flat, ldsTables, grid, gridLds;
    size_t candBytes, leafBytes;
};
static TraceVariant ChooseVariant(const rt_ctx* ctx, const rtd::TraceParams& tp) {
    TraceVariant V{};
    const size_t lds = LdsBytesFor(tp.n, tp.n_padded, ctx->matsInLds);
    const bool useLds = !ctx->forceGlobal && lds <= 48 * 1024 && tp.n_padded < 65536;
    const uint32_t wavesPerBlock = ctx->blockThreads / 64;
    V.grid = ctx->useMfma && tp.grid_cell_start != nullptr;  // cell-grid scan (the scene was laid out for it at upload): tables in global memory (L2)
    V.tree = ctx->useMfma && !V.grid && tp.n_levels > 1;  // deeper hierarchy: tables stay in global memory (L2)
    const uint32_t topCnt = tp.level_cnt[tp.n_levels - 1];
    // the scene constants' slot comes first in the image, then the per-wave regions
    V.candBytes = rtd::kConstBytes + (size_t)wavesPerBlock * (V.grid ? rtd::kWaveGridBytes : (V.tree ? rtd::kWaveCandBytes : rtd::kWaveListBytes)) +
                  (size_t)tp.sg_glob16 * 16;  // ... and the shadow index's global list in front of the tables (rt_params.h sg_glob_slots)
    // one-sphere bounds, staged next to the scan table: the flat scan's copy has kFlatLeafStride float4 per group (rt_scan.h), the grid's is dense
    V.leafBytes = (size_t)(tp.n_padded / 4u) * (V.grid ? 4u : rtd::kFlatLeafStride) * 16;
    V.flat = !V.tree && !V.grid && useLds && ctx->useMfma && (V.candBytes + lds + V.leafBytes + MfmaOpsBytesFor(topCnt)) <= 160 * 1024;
    V.ldsTables = useLds && !V.tree && !V.grid;
    // a grid scene whose tables all fit LDS next to the grid scan's work lists (small scenes laid out for the grid: RT_GRID=2)
    V.gridLds = V.grid && useLds && ctx->matsInLds && ctx->blockThreads == 1024 &&
                V.candBytes + lds + V.leafBytes + ((size_t)tp.grid_nu * tp.grid_nv + 1) * 2 + 16 <= 128 * 1024;
    return V;
}

// The queue of fresh paths (rt_params.h): the first static_blocks blocks of every launched wave are static, the remaining
// paths are cut into eight shards of whole blocks.
static void QueueShards(rtd::TraceParams& tp, uint32_t wavesLaunched, uint32_t queueBlock, uint32_t staticBlocks) {
    tp.queue_block = queueBlock;
    tp.static_blocks = staticBlocks;
    const uint64_t total = tp.total_paths, qb = queueBlock;
    uint64_t begin = (uint64_t)wavesLaunched * qb * staticBlocks;
    if (begin > total) begin = total;
    tp.dyn_begin = (uint32_t)begin;
    tp.dyn_blocks = (uint32_t)((total - begin + qb - 1) / qb);
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (device, function), process-wide: every kernel variant is raised
// ONCE per device to the 160 KiB a workgroup can have, so that two contexts with different scenes can never disagree about it
// (a per-context cache of "the last value I set" could skip the call after another context had lowered the limit), and a
// launch-bound 1-spp frame pays the attribute call only the first time.
static int RaiseLdsLimit(int device, const void* fn) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> raised;
    std::lock_guard<std::mutex> lock(mu);
    if (raised.count({device, fn})) return RT_OK;
    RT_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised.insert({device, fn});
    return RT_OK;
}

// Launch the megakernel over total paths described by tp.  carryMode 0: ordinary launch.  1: probe -- RT_OK iff this scene
// and these settings get the kernel variant that implements frame pipelining (nothing is launched).  2: launch that variant
// (tp.ctl etc. filled by the caller; the queue cursor lives in tp.ctl and is reset by the preparation kernel).
static int LaunchTrace(rt_ctx* ctx, rtd::TraceParams& tp, int carryMode = 0) {
    tp.shard_heads = ctx->queue.ptr;
    tp.fd_tile = rtd::make_fastdiv(tp.spp_pass ? 64u * tp.spp_pass : 1u);  // path_coordinates' divisors (rt_params.h, FastDiv)
    tp.fd_w = rtd::make_fastdiv(tp.W ? tp.W : 1u);
    tp.fd_rows = rtd::make_fastdiv(tp.rs.block_rows ? tp.rs.block_rows : 1u);
    tp.mats_in_lds = ctx->matsInLds ? 1u : 0u;
    // packed materials (decided below, once it is known where the variant reads its materials from).  RT_MATS16: 0 (default) never, 1
    // from global memory wherever the 48-byte table would be read from there, 2 also staged into LDS by the flat stash variant.
    // Measured (C2, three interleaved rounds): 48-byte records through L2 9,983 / packed from global memory 9,972 / packed in LDS
    // with a 56-record stash 9,977 Msamples/s -- the material read is not what a hit waits for; C5: 7,579 -> 7,596; the unpacking adds 62 lane-operations per
    // sample (+1.3 % of the kernel's VALU instructions, tools/phase_budget.py): same time for more instructions, so it stays off.
    const bool want16 = tp.mats16 != nullptr && EnvU32("RT_MATS16", 0u) != 0u;
    tp.mats16_mode = 0u;
    tp.sg_glob16 = tp.sg_enabled ? rtd::sg_glob_slots(tp.sg_nglobal) : 0u;
    const size_t lds = LdsBytesFor(tp.n, tp.n_padded, ctx->matsInLds);
    if (tp.n_padded >= 65536) return Fail(RT_ERR_INVALID_ARG, "scenes beyond 65,000 spheres are not supported by the 16-bit candidate lists");
    const uint32_t maxBlocks = (uint32_t)ctx->cuCount * ctx->blocksPerCu;
    // one wave holds 64 paths; do not launch more waves than there is work for
    const uint64_t wavesNeeded = ((uint64_t)tp.total_paths + 63) / 64;
    const uint32_t wavesPerBlock = ctx->blockThreads / 64;
    uint32_t blocks = (uint32_t)((wavesNeeded + wavesPerBlock - 1) / wavesPerBlock);
    if (blocks > maxBlocks) blocks = maxBlocks;
    if (blocks == 0) blocks = 1;
    if (carryMode == 2) blocks = maxBlocks;  // carried paths may outnumber the fresh ones (a flush has none)
    if (carryMode == 0) {
        // queue cursors for this launch (the pipelined path sets them in its preparation kernel)
        // 128-path blocks; 256 once a wave will take more than ~32 of them anyway (measured: +0.7 % at 1200x800 spp 128, +0.8 % on C4,
        // +0.9 % on grid10k, neutral at spp 48, -1.7 % at spp 16, where the finer blocks balance the short launch's end better)
        uint32_t qb = EnvU32("RT_QUEUE_BLOCK", 0u) / 64u * 64u;
        if (qb == 0u) qb = (uint64_t)tp.total_paths >= 8192ull * blocks * wavesPerBlock ? 2u * rtd::kQueueBlock : rtd::kQueueBlock;
        QueueShards(tp, blocks * wavesPerBlock, qb, EnvU32("RT_QUEUE_STATIC", 1u));
        hipLaunchKernelGGL(rtd::rt_raygen_tables_kernel, dim3(1), dim3(64), 0, ctx->stream, (float2*)nullptr, 0u, 0u, (float2*)nullptr, 0u, 0u, 0u,
                           ctx->queue.ptr, (rtd::FrameCtl*)nullptr);
        RT_HIP(hipGetLastError());
    }
    // dynamic LDS: per-wave candidate regions + the scene tables when they fit + the filter operand image
    const TraceVariant V = ChooseVariant(ctx, tp);
    const bool tree = V.tree, flat = V.flat, ldsTables = V.ldsTables;
    const uint32_t topCnt = tp.level_cnt[tp.n_levels - 1];
    const size_t candBytes = V.candBytes, leafBytes = V.leafBytes;
    bool gridLds = V.gridLds;
    size_t sgBytes = ((flat || gridLds) && tp.sg_enabled) ? (((size_t)tp.sg_nx * tp.sg_ny + 1 + tp.sg_nentries + tp.sg_nglobal) * 2 + 15) / 16 * 16 : 0;
    if (candBytes + lds + leafBytes + (flat ? MfmaOpsBytesFor(topCnt) : 0) + sgBytes + (gridLds ? ((size_t)tp.grid_nu * tp.grid_nv + 1) * 2 + 16 : 0) > 160 * 1024) {
        sgBytes = 0;  // index stays in global memory (L2)
        if (tp.sg_enabled) gridLds = false;  // ... and the all-in-LDS grid variant needs it there
    }
    tp.sg_in_lds = sgBytes ? 1u : 0u;
    size_t treeBytes = tree ? (size_t)(tp.level_off[tp.n_levels - 1] + topCnt) * 16 : 0;
    if (!ctx->treeInLds || candBytes + MfmaOpsBytesFor(topCnt) + treeBytes > 160 * 1024) treeBytes = 0;
    tp.tree_in_lds = treeBytes ? 1u : 0u;
    const bool grid = V.grid;
    // (the cells go to LDS in the 1024-thread variants; smaller workgroups -- a knob for experiments -- read them through L2)
    size_t gridBytes = (grid && ctx->blockThreads == 1024) ? (((size_t)tp.grid_nu * tp.grid_nv + 1) * 2 + 15) / 16 * 16 : 0;
    if (candBytes + gridBytes > 160 * 1024) gridBytes = 0;  // (cannot happen below 60,000 cells)
    tp.grid_in_lds = gridBytes ? 1u : 0u;
    // ... and, RT_GRID_QUANT=1 (experiments; BASELINE configs[4]'s "LDS-tiled sphere list"): in the stash variant with global tables
    // the QUANTISED one-sphere bounds behind the cells (rt_scan.h GridQuant: the step loop then reads no global memory) when they
    // leave room for a stash of at least 16 records.  Measured on grid10k (4096^2, spp 64): 40 KB of bounds leave 38 stash records
    // instead of 63; 7.59 -> 7.10 Gsamples/s, of which -3.8 % is the smaller stash (float4 bounds at 38 records: 7.31) and -2.8 %
    // the 17 extra operations per tested sphere that unpack the record -- the float4 bounds hit L1 95 % of the time, and what the
    // L1 serves per cycle was not the limit it looked like (profiles/r04_c5_placement.json).  Default off.
    size_t gridQBytes = 0;
    if (grid && !gridLds && gridBytes != 0 && tp.grid_qrec != nullptr && ctx->useStash && carryMode == 0 && ctx->blockThreads == 1024 &&
        tp.max_depth < 65536u && EnvU32("RT_GRID_QUANT", 0u) != 0u && EnvU32("RT_GRID_SG_LDS", 0u) == 0u && EnvU32("RT_STASH_CAP", 63u) >= 16u) {
        const size_t qb = ((size_t)tp.n_padded * 4 + 15) / 16 * 16;
        if (candBytes + gridBytes + qb + (size_t)wavesPerBlock * 16 * rtd::kStashDwords * 4 + 256 <= 160 * 1024) gridQBytes = qb;
    }
    // cell-grid scan with its tables in global memory, RT_GRID_SG_LDS=1 (experiments): the shadow index in LDS next to the cells when
    // that leaves a stash of at least 24 records.  Measured on grid10k: 48.6 KB of index leave 31 records instead of 63: -5 %
    // (6.47 vs 6.82 Gsamples/s) -- the full stash is worth more than the three dependent L2 reads per shadow query it would save.
    bool gridSgLds = false;
    if (grid && !gridLds && gridBytes != 0 && tp.sg_enabled && ctx->useStash && carryMode == 0 && ctx->blockThreads == 1024 && tp.max_depth < 65536u &&
        EnvU32("RT_GRID_SG_LDS", 0u) != 0u) {
        const size_t sgb = (((size_t)tp.sg_nx * tp.sg_ny + 1 + tp.sg_nentries + tp.sg_nglobal) * 2 + 15) / 16 * 16;
        const size_t used = candBytes + gridBytes + sgb;
        if (used + (size_t)wavesPerBlock * 24 * rtd::kStashDwords * 4 + 256 <= 160 * 1024) {
            gridSgLds = true;
            sgBytes = sgb;
        }
    }
    size_t ldsBytes = candBytes + ((ldsTables || gridLds) ? lds : 0) + ((flat || gridLds) ? leafBytes : 0) + ((flat || tree) ? MfmaOpsBytesFor(topCnt) : 0) + sgBytes + treeBytes + gridBytes + gridQBytes;
    // per-wave caches of prepared paths go last, when there is room left (RT_RAY_CACHE=0 disables them)
    ldsBytes = (ldsBytes + 15) / 16 * 16;
    tp.ray_cache_off16 = 0;
    tp.ray_cache_stride16 = 0;
    tp.stash_cap = 0;
    // Hit stash (rt_kernels.h, kStash): the matrix-core variants at 1024 threads regroup their hit processing through a per-wave
    // stash of up to 63 hit records in the room the path cache would take -- as many records as the LDS left over holds
    // (RT_STASH=0: the path-cache variants; RT_STASH_CAP: fewer records).  depth shares its register with the scan entry.
    bool useStash = false, matsL2 = false;
    {
        const size_t budget = 160 * 1024 / ctx->blocksPerCu;
        const uint32_t capEnv = EnvU32("RT_STASH_CAP", 63u);
        auto capFor = [&](size_t imageBytes) -> uint32_t {  // records per wave that fit behind an image of this size
            if (budget <= imageBytes) return 0u;
            const size_t room = (budget - imageBytes) / wavesPerBlock / 16 * 16;
            uint32_t c = (uint32_t)(room / (rtd::kStashDwords * 4));
            c = c > 63u ? 63u : c;
            return c > capEnv ? capEnv : c;
        };
        const bool stashKernel = ctx->useStash && carryMode == 0 && (flat || tree || grid) && ctx->blockThreads == 1024 && tp.max_depth < 65536u;
        uint32_t cap = capFor(ldsBytes);
        // Flat variant: when the material table's 48 bytes per sphere would buy at least eight more records per wave, the
        // materials are read through L2 instead (kMatsL2; measured on the cover scene: 44 -> 63 records, +1 %; RT_MATS_L2=0: never).
        // Decided on the candidate image WITHOUT touching ldsBytes: the smaller image is committed only together with the stash
        // variant that is built for it (the other flat kernels stage the material table and need its room).
        if (stashKernel && flat && tp.mats_in_lds && (sgBytes != 0 || !tp.sg_enabled) && EnvU32("RT_MATS_L2", 1u) != 0u) {
            const size_t matBytes = (size_t)tp.n_padded * 48;
            const uint32_t cap2 = matBytes <= ldsBytes ? capFor(ldsBytes - matBytes) : 0u;
            if (cap2 >= cap + 8u && cap2 >= 16u) {
                matsL2 = true;
                ldsBytes -= matBytes;
                cap = cap2;
                // ... and, RT_MATS16=2, their packed form into the room that leaves, when the stash keeps at least 48 records with it
                const size_t m16Bytes = (size_t)tp.n_padded * 16;
                const uint32_t cap3 = capFor(ldsBytes + m16Bytes);
                if (want16 && EnvU32("RT_MATS16", 0u) >= 2u && cap3 >= 48u) {
                    tp.mats16_mode = 2u;
                    ldsBytes += m16Bytes;
                    cap = cap3;
                }
            }
        }
        if (stashKernel && cap >= 16u) {
            useStash = true;
            tp.stash_cap = cap;
            const uint32_t procEnv = EnvU32("RT_STASH_PROCESS", 63u);  // experiments: process hits from this many + 1 lanes on
            tp.stash_process = procEnv < cap ? procEnv : cap;
            tp.ray_cache_off16 = (uint32_t)(ldsBytes / 16);
            tp.ray_cache_stride16 = (cap * rtd::kStashDwords * 4 + 15) / 16;
            ldsBytes += (size_t)wavesPerBlock * tp.ray_cache_stride16 * 16;
        }
        if (matsL2 && !useStash) return Fail(RT_ERR_HIP, "internal: materials through L2 without the stash variant");
        // every other variant that reads its materials from GLOBAL memory takes the packed record from there (one 16-byte read per
        // hit instead of three); variants with the 48-byte table in LDS keep it
        const bool matsGlobal = matsL2 || tree || (grid && !gridLds) || !tp.mats_in_lds || (!flat && !ldsTables && !gridLds);
        if (want16 && tp.mats16_mode == 0u && matsGlobal) tp.mats16_mode = 1u;
    }
    if (!useStash && ctx->useRayCache && ldsBytes + (size_t)wavesPerBlock * rtd::kRayCacheBytes <= 160 * 1024 / ctx->blocksPerCu) {
        tp.ray_cache_off16 = (uint32_t)(ldsBytes / 16);
        tp.ray_cache_stride16 = rtd::kRayCacheBytes / 16;
        ldsBytes += (size_t)wavesPerBlock * rtd::kRayCacheBytes;
    }
    if (std::getenv("RT_VERBOSE"))
        std::fprintf(stderr, "rt_trace launch: tree=%d flat=%d ldsTables=%d blocks=%u threads=%u lds=%zu B (cand %zu, tables %zu, leaf %zu, ops %zu, sg %zu, tree %zu, cache %s, stash %u)\n",
                     (int)tree, (int)flat + 2 * (int)grid + 4 * (int)gridLds, (int)ldsTables, blocks, ctx->blockThreads, ldsBytes, candBytes, ldsTables ? lds : (size_t)0, flat ? leafBytes : (size_t)0,
                     (flat || tree) ? MfmaOpsBytesFor(topCnt) : (size_t)0, sgBytes, treeBytes, (tp.ray_cache_off16 && !useStash) ? "yes" : "no", tp.stash_cap);
    // flat variant with the hit-processing tables provably in LDS (typed pointers: no flat loads) when they all fit
    const bool hitLds = flat && tp.mats_in_lds && (sgBytes != 0 || !tp.sg_enabled);
    if (carryMode != 0) {
        const bool ok = hitLds && tp.ray_cache_off16 != 0 && ctx->blockThreads == 1024;
        if (carryMode == 1) return ok ? RT_OK : RT_ERR_INVALID_ARG;
        if (!ok) return Fail(RT_ERR_INVALID_ARG, "frame pipelining needs the flat LDS variant of the trace kernel");
    }
#define RT_LAUNCH_FN(KERNEL)                                                                                                  \
    do {                                                                                                                      \
        const void* fn_ = reinterpret_cast<const void*>(&KERNEL);                                                             \
        /* the attribute belongs to the FUNCTION on this device, not to the context: raised once to what any launch can ask for, never lowered */ \
        if (ldsBytes > 48 * 1024 && RaiseLdsLimit(ctx->device, fn_) != RT_OK) return RT_ERR_HIP;                               \
        hipLaunchKernelGGL(KERNEL, dim3(blocks), dim3(ctx->blockThreads), ldsBytes, ctx->stream, tp);                         \
    } while (0)
/* the variant with these template arguments: the single-light kernel, or its twin over the light list (rt_kernels.h trace_body) */
#define RT_LAUNCH_K(...)                                                          \
    do {                                                                          \
        if (tp.n_lights == 1u) RT_LAUNCH_FN((rtd::rt_trace_kernel<__VA_ARGS__>)); \
        else RT_LAUNCH_FN((rtd::rt_trace_kernel_lights<__VA_ARGS__>));            \
    } while (0)
#define RT_LAUNCH(LDS, T, M)                                                                        \
    do {                                                                                            \
        if (((M) == 1 && hitLds) || ((M) == 2 && tp.tree_in_lds)) {                                 \
            if (tp.ray_cache_off16) RT_LAUNCH_K(LDS, T, M, true, (M) != 0);  \
            else RT_LAUNCH_K(LDS, T, M, false, (M) != 0);                    \
        } else {                                                                                    \
            if (tp.ray_cache_off16) RT_LAUNCH_K(LDS, T, M, true, false);    \
            else RT_LAUNCH_K(LDS, T, M, false, false);                      \
        }                                                                                           \
    } while (0)
#define RT_LAUNCH_T(LDS, M)                                            \
    do {                                                               \
        if (ctx->blockThreads == 1024) RT_LAUNCH(LDS, 1024, M);        \
        else if (ctx->blockThreads == 512) RT_LAUNCH(LDS, 512, M);     \
        else RT_LAUNCH(LDS, 256, M);                                   \
    } while (0)
    if (carryMode == 2) RT_LAUNCH_K(true, 1024, 1, true, true, true);
    else if (gridLds && useStash && tp.grid_in_lds) RT_LAUNCH_K(true, 1024, 3, true, true, false, true);
    else if (gridLds && tp.grid_in_lds && tp.ray_cache_off16) RT_LAUNCH_K(true, 1024, 3, true, true);
    else if (gridLds && tp.grid_in_lds) RT_LAUNCH_K(true, 1024, 3, false, true);
    else if (grid && gridSgLds && useStash && tp.grid_in_lds) RT_LAUNCH_K(false, 1024, 3, true, true, false, true, false, true);
    else if (grid && gridSgLds) return Fail(RT_ERR_HIP, "internal: the grid variant with the shadow index in LDS needs the hit stash");
    else if (grid && ctx->blockThreads == 1024 && useStash && tp.grid_in_lds && gridQBytes != 0) RT_LAUNCH_K(false, 1024, 3, true, true, false, true, false, false, true);
    else if (grid && gridQBytes != 0) return Fail(RT_ERR_HIP, "internal: quantised grid bounds without the stash variant");
    else if (grid && ctx->blockThreads == 1024 && useStash && tp.grid_in_lds) RT_LAUNCH_K(false, 1024, 3, true, true, false, true);
    else if (grid && ctx->blockThreads == 1024 && tp.grid_in_lds && tp.ray_cache_off16) RT_LAUNCH_K(false, 1024, 3, true, true);
    else if (grid && ctx->blockThreads == 1024 && tp.grid_in_lds) RT_LAUNCH_K(false, 1024, 3, false, true);
    else if (grid && ctx->blockThreads == 1024) RT_LAUNCH_K(false, 1024, 3, false, false);
    else if (grid && ctx->blockThreads == 512 && !tp.ray_cache_off16) RT_LAUNCH_K(false, 512, 3, false, false);
    else if (grid && ctx->blockThreads == 512) RT_LAUNCH_K(false, 512, 3, true, false);
    else if (grid && !tp.ray_cache_off16) RT_LAUNCH_K(false, 256, 3, false, false);
    else if (grid) RT_LAUNCH_K(false, 256, 3, true, false);
    else if (useStash && tree && tp.tree_in_lds) RT_LAUNCH_K(false, 1024, 2, true, true, false, true);
    else if (useStash && tree) RT_LAUNCH_K(false, 1024, 2, true, false, false, true);
    else if (useStash && hitLds && matsL2) RT_LAUNCH_K(true, 1024, 1, true, true, false, true, true);
    else if (useStash && hitLds) RT_LAUNCH_K(true, 1024, 1, true, true, false, true);
    else if (useStash) RT_LAUNCH_K(true, 1024, 1, true, false, false, true);
    else if (tree) RT_LAUNCH_T(false, 2);
    else if (flat) RT_LAUNCH_T(true, 1);
    else if (ldsTables) RT_LAUNCH_T(true, 0);
    else RT_LAUNCH_T(false, 0);
#undef RT_LAUNCH_T
#undef RT_LAUNCH
#undef RT_LAUNCH_K
#undef RT_LAUNCH_FN
    RT_HIP(hipGetLastError());
    return RT_OK;
}


// rt_unit_closest_hit's device part: the scan variant rt_render would launch for this scene, 64 rays per wave (LDS image
// for four waves, hit-processing tables left out).  Also traces the pilot rays of the tile order.
static int LaunchClosest(rt_ctx* ctx, const float* dRays, uint32_t n, float* dOut) {
        // the scan variant rt_render would launch for this scene; LDS image for four waves, hit-processing tables left out
        rtd::TraceParams tp = ctx->base;
        const TraceVariant V = ChooseVariant(ctx, tp);
        tp.mats_in_lds = 0;
        tp.sg_in_lds = 0;
        const uint32_t topCnt = tp.level_cnt[tp.n_levels - 1];
        const size_t waves = 256 / 64;
        size_t ldsBytes = waves * (V.grid ? rtd::kWaveGridBytes : (V.tree ? rtd::kWaveCandBytes : rtd::kWaveListBytes));
        if (V.grid) {
            tp.grid_in_lds = 0;
        } else if (V.tree) {
            ldsBytes += MfmaOpsBytesFor(topCnt);
            const size_t treeBytes = (size_t)(tp.level_off[tp.n_levels - 1] + topCnt) * 16;
            tp.tree_in_lds = (ctx->treeInLds && ldsBytes + treeBytes <= 160 * 1024) ? 1u : 0u;
            if (tp.tree_in_lds) ldsBytes += treeBytes;
        } else if (V.flat || V.ldsTables) {
            ldsBytes += LdsBytesFor(tp.n, tp.n_padded, false);
            if (V.flat) ldsBytes += V.leafBytes + MfmaOpsBytesFor(topCnt);
        }
#define RT_UNIT_CLOSEST(LDS, M)                                                                                        \
    do {                                                                                                               \
        const void* fn_ = reinterpret_cast<const void*>(&rtd::k_unit_closest<LDS, M>);                                 \
        if (ldsBytes > 48 * 1024) RT_HIP(hipFuncSetAttribute(fn_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes)); \
        hipLaunchKernelGGL((rtd::k_unit_closest<LDS, M>), dim3((n + 255) / 256), dim3(256), ldsBytes, ctx->stream, tp, dRays, n, dOut); \
    } while (0)
        if (V.grid) RT_UNIT_CLOSEST(false, 3);
        else if (V.tree) RT_UNIT_CLOSEST(false, 2);
        else if (V.flat) RT_UNIT_CLOSEST(true, 1);
        else if (V.ldsTables) RT_UNIT_CLOSEST(true, 0);
        else RT_UNIT_CLOSEST(false, 0);
#undef RT_UNIT_CLOSEST
        RT_HIP(hipGetLastError());
    return RT_OK;
}

// Work order of the full tiles for the accumulation that is starting (rt_kernels.h, rt_tile_order_kernel): three pilot rays per
// tile through the production scan, then a stable sort by the most expensive first-hit material.  All on the stream, no host wait.
static int BuildTileOrder(rt_ctx* ctx, uint32_t W, uint32_t H, rt_rowset rs, uint32_t npix) {
    // the order is a function of the scene (camera included), the image size and the strip: a new accumulation of the same
    // picture -- a progressive restart, the next frame of a turntable with an unchanged scene -- keeps the one it has
    if (ctx->tileOrderValid && ctx->tileW == W && ctx->tileH == H && std::memcmp(&ctx->tileRs, &rs, sizeof(rs)) == 0) return RT_OK;
    ctx->tileOrderValid = false;
    const uint32_t nFull = npix >> 6;
    if (!ctx->useTileOrder || nFull < 2u * (uint32_t)ctx->cuCount) return RT_OK;  // too little work for the order to matter
    int rc;
    const uint32_t nPilot = nFull * rtd::kPilotsPerTile;
    if ((rc = ctx->pilotRays.Reserve((size_t)nPilot * 6)) != RT_OK) return rc;
    if ((rc = ctx->pilotHits.Reserve((size_t)nPilot * 10)) != RT_OK) return rc;
    if ((rc = ctx->tileClass.Reserve(nFull)) != RT_OK) return rc;
    if ((rc = ctx->tileOrder.Reserve(nFull)) != RT_OK) return rc;
    rtd::TraceParams tp = ctx->base;
    tp.W = W;
    tp.H = H;
    tp.rs = rs;
    tp.s0 = 1;
    tp.sampler = ctx->sampler;
    tp.jitter_tab = nullptr;
    tp.lens_tab = nullptr;
    hipLaunchKernelGGL(rtd::rt_pilot_rays_kernel, dim3((nPilot + 255) / 256), dim3(256), 0, ctx->stream, tp, nFull, ctx->pilotRays.ptr);
    RT_HIP(hipGetLastError());
    if ((rc = LaunchClosest(ctx, ctx->pilotRays.ptr, nPilot, ctx->pilotHits.ptr)) != RT_OK) return rc;
    hipLaunchKernelGGL(rtd::rt_tile_class_kernel, dim3((nFull + 255) / 256), dim3(256), 0, ctx->stream, ctx->pilotHits.ptr, nFull, W,
                       ctx->matType.ptr, ctx->tileClass.ptr);
    hipLaunchKernelGGL(rtd::rt_tile_order_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->tileClass.ptr, nFull, ctx->tileOrder.ptr);
    RT_HIP(hipGetLastError());
    ctx->tileOrderValid = true;
    ctx->tileW = W;
    ctx->tileH = H;
    ctx->tileRs = rs;
    return RT_OK;
}

// ------------------------------------------------------------------------------ frame pipelining
// (rt_params.h "frame pipelining")  Host side: one region of the sample ring per rt_render call, two
// continuation buffers used alternately, and three launches per call: preparation + ray-generation tables, the carrying
// trace kernel, the commit kernel.  Nothing here waits for the device.
static void PipelineDrop(rt_ctx* ctx) { ctx->pipeOpen = false; }  // carried paths and uncommitted regions are abandoned
static uint32_t PipelineWaves(const rt_ctx* ctx) { return (uint32_t)ctx->cuCount * ctx->blocksPerCu * (ctx->blockThreads / 64); }
static void PipelineShards(const rt_ctx* ctx, rtd::TraceParams& tp) {
    // measured (tools/progressive_frames.py, 1-spp frames): 128-path blocks x 1 static beat 64 x 2, 128 x 0, 192 x 1, 256 x 0
    uint32_t qb = EnvU32("RT_PIPE_QUEUE_BLOCK", rtd::kCarryQueueBlock) / 64u * 64u;
    if (qb == 0) qb = 64;
    QueueShards(tp, PipelineWaves(ctx), qb, EnvU32("RT_PIPE_STATIC_BLOCKS", 1));
}

static int PipelineTraceAndCommit(rt_ctx* ctx, rtd::TraceParams& tp, uint32_t npix, bool carry) {
    tp.ctl = ctx->ctl.ptr;
    tp.cont_in = ctx->cont[ctx->pipeInSel].ptr;
    tp.cont_in_n = ctx->contN[ctx->pipeInSel].ptr;
    tp.cont_out = ctx->cont[ctx->pipeInSel ^ 1u].ptr;
    tp.cont_out_n = ctx->contN[ctx->pipeInSel ^ 1u].ptr;
    tp.carry = carry ? 1u : 0u;
    tp.region_seq = ctx->pipeSeq;
    tp.max_carry_age = ctx->pipeDepth;
    tp.min_iters = EnvU32("RT_PIPE_MIN_ITERS", 8);
    tp.counters = ctx->counters.ptr;
    int rc = LaunchTrace(ctx, tp, 2);
    if (rc != RT_OK) return rc;
    hipLaunchKernelGGL(rtd::rt_commit_kernel, dim3((npix + 255) / 256), dim3(256), 0, ctx->stream, ctx->ring.ptr, ctx->hdr.ptr, npix,
                       npix * ctx->pipeSppCap, ctx->pipeRing, ctx->pipeRegions, ctx->ctl.ptr, ctx->pipeSeq, ctx->pipeCommits);
    RT_HIP(hipGetLastError());
    ++ctx->pipeCommits;  // the commit point is now entry pipeCommits & 1 of the control block
    ctx->pipeInSel ^= 1u;
    return RT_OK;
}

// Run every carried path to its end and commit every region: afterwards the HDR strip holds ctx->accumulated samples.
static int PipelineFlush(rt_ctx* ctx) {
    if (!ctx->pipeOpen) return RT_OK;
    const uint32_t npix = ctx->pipeNpix;
    rtd::TraceParams tp = ctx->base;
    tp.total_paths = 0;  // nothing fresh: only the carried paths
    PipelineShards(ctx, tp);
    hipLaunchKernelGGL(rtd::rt_raygen_tables_kernel, dim3(1), dim3(256), 0, ctx->stream, (float2*)nullptr, 0u, 0u, (float2*)nullptr, 0u, 0u,
                       ctx->sampler, ctx->queue.ptr, ctx->ctl.ptr);
    RT_HIP(hipGetLastError());
    tp.W = ctx->W;
    tp.H = ctx->H;
    tp.rs = ctx->rs;
    tp.s0 = 1;
    tp.spp_pass = 1;
    tp.npix_local = npix;
    tp.max_depth = ctx->pipeMaxDepth;
    tp.sampler = ctx->sampler;
    tp.seed = ctx->pipeSeed;
    tp.samples = ctx->ring.ptr;
    int rc = PipelineTraceAndCommit(ctx, tp, npix, false);
    ctx->pipeOpen = false;
    if (rc != RT_OK) ctx->accumulated = 0;
    return rc;
}

// One pipelined call: samples [s0, s1) of the strip become region pipeSeq + 1 of the ring.
static int PipelineRender(rt_ctx* ctx, uint32_t W, uint32_t H, rt_rowset rs, uint32_t npix, uint32_t s0, uint32_t s1, uint32_t max_depth,
                          uint64_t seed) {
    const uint32_t spp = s1 - s0;
    const uint32_t nRing = ctx->pipeDepth + 2u;
    int rc;
    if (ctx->pipeOpen && (ctx->pipeNpix != npix || spp > ctx->pipeSppCap || ctx->pipeRing != nRing || ctx->pipeMaxDepth != max_depth ||
                          ctx->pipeSeed != seed)) {
        if ((rc = PipelineFlush(ctx)) != RT_OK) return rc;  // geometry or parameters changed: start a new pipeline
    }
    if (!ctx->pipeOpen) {
        ctx->pipeNpix = npix;
        ctx->pipeSppCap = spp;
        ctx->pipeRing = nRing;
        ctx->pipeMaxDepth = max_depth;
        ctx->pipeSeed = seed;
        ctx->pipeInSel = 0;
        ctx->pipeRegions = rtd::RegionTable{};
        const size_t waves = PipelineWaves(ctx);
        if ((rc = ctx->ring.Reserve((size_t)nRing * npix * spp * 3)) != RT_OK) return rc;
        for (int k = 0; k < 2; ++k) {
            if ((rc = ctx->cont[k].Reserve(waves * 64)) != RT_OK) return rc;
            if ((rc = ctx->contN[k].Reserve(waves)) != RT_OK) return rc;
            RT_HIP(hipMemsetAsync(ctx->contN[k].ptr, 0, waves * sizeof(uint32_t), ctx->stream));  // nothing carried yet
        }
        if ((rc = ctx->ctl.Reserve(1)) != RT_OK) return rc;
        rtd::FrameCtl init{};
        init.oldest_open = 0xffffffffu;
        init.committed_seq[0] = ctx->pipeSeq;              // everything up to here is in the strip already
        init.committed_samples[0] = ctx->accumulated;
        ctx->pipeCommits = 0;
        RT_HIP(hipMemcpyAsync(ctx->ctl.ptr, &init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(hipStreamSynchronize(ctx->stream));       // `init` is a stack object; once per pipeline start
        ctx->pipeOpen = true;
    }
    const uint32_t seq = ++ctx->pipeSeq;
    const uint32_t slot = seq % nRing;
    ctx->pipeRegions.seq[slot] = seq;
    ctx->pipeRegions.spp[slot] = spp;
    rtd::TraceParams tp = ctx->base;
    tp.W = W;
    tp.H = H;
    tp.rs = rs;
    tp.s0 = s0;
    tp.spp_pass = spp;
    tp.total_paths = npix * spp;
    tp.npix_local = npix;
    tp.max_depth = max_depth;
    tp.sampler = ctx->sampler;
    tp.seed = seed;
    tp.path_list = nullptr;
    // natural tile order: a frame's long paths are carried into the next kernel anyway, and starting every wave on the most
    // expensive tiles only lengthens the stretch before the first wave runs dry (measured 0.285 vs 0.253 ms per frame)
    tp.tile_order = nullptr;
    tp.samples = ctx->ring.ptr;
    tp.sample_base = slot * npix * ctx->pipeSppCap;
    tp.trav_out = nullptr;
    const uint32_t k0 = s0 + rs.first_row;
    const uint32_t nLens = spp + W + rs.num_rows;
    if ((rc = ctx->jitterTab.Reserve(spp)) != RT_OK) return rc;
    if ((rc = ctx->lensTab.Reserve(nLens)) != RT_OK) return rc;
    tp.jitter_tab = ctx->jitterTab.ptr;
    tp.lens_tab = ctx->lensTab.ptr;
    tp.lens_k0 = k0;
    const uint32_t nmax = spp > nLens ? spp : nLens;
    PipelineShards(ctx, tp);
    hipLaunchKernelGGL(rtd::rt_raygen_tables_kernel, dim3((nmax + 255) / 256), dim3(256), 0, ctx->stream, ctx->jitterTab.ptr, s0, spp,
                       ctx->lensTab.ptr, k0, nLens, ctx->sampler, ctx->queue.ptr, ctx->ctl.ptr);
    RT_HIP(hipGetLastError());
    return PipelineTraceAndCommit(ctx, tp, npix, true);
}

namespace {
template <typename T>
struct TmpDev {
    T* p = nullptr;
    ~TmpDev() {
        if (p) (void)hipFree(p);
    }
    hipError_t Alloc(size_t n) { return hipMalloc(reinterpret_cast<void**>(&p), (n ? n : 1) * sizeof(T)); }
};
}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }
int rt_api_version(void) { return RT_API_VERSION; }

int rt_create(int device_ordinal, rt_ctx** out) {
    if (!out) return Fail(RT_ERR_INVALID_ARG, "rt_create: null out");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return Fail(RT_ERR_NO_DEVICE, std::string("rt_create: no HIP device (") + hipGetErrorString(e) +
                                          "); this library has no CPU fallback");
    if (device_ordinal < 0 || device_ordinal >= count) return Fail(RT_ERR_NO_DEVICE, "rt_create: device ordinal out of range");
    RT_HIP(hipSetDevice(device_ordinal));
    // owned until the end: every early return below releases what has been created so far (rt_destroy copes with
    // half-initialised contexts)
    std::unique_ptr<rt_ctx, void (*)(rt_ctx*)> guard(new rt_ctx(), rt_destroy);
    rt_ctx* ctx = guard.get();
    ctx->device = device_ordinal;
    hipDeviceProp_t prop;
    RT_HIP(hipGetDeviceProperties(&prop, device_ordinal));
    ctx->cuCount = prop.multiProcessorCount;
    ctx->ldsPerBlockMax = prop.sharedMemPerBlock;
    RT_HIP(hipStreamCreateWithFlags(&ctx->ownStream, hipStreamNonBlocking));
    ctx->stream = ctx->ownStream;
    for (auto& ev : ctx->ev) RT_HIP(hipEventCreate(&ev));
    {
        const char* scan = std::getenv("RT_SCAN");
        ctx->useMfma = !(scan && std::strcmp(scan, "valu") == 0);
        ctx->matsInLds = EnvU32("RT_MATS_LDS", 1) != 0;
        ctx->useShadowGrid = EnvU32("RT_SHADOW_GRID", 1) != 0;
        ctx->useRayCache = EnvU32("RT_RAY_CACHE", 1) != 0;
        ctx->useStash = EnvU32("RT_STASH", 1) != 0;
        ctx->treeInLds = EnvU32("RT_TREE_LDS", 1) != 0;
        ctx->useTileOrder = EnvU32("RT_TILE_ORDER", 1) != 0;
        ctx->treeTop = TreeTopFromEnv();
    }
    // launch geometry (sweeps: profiles/r01_sweep_*.jsonl): the matrix-core scan wants 16 waves per CU in ONE
    // 1024-thread workgroup (one LDS image, 128 VGPRs); the pure-VALU scan runs 4 x 256 threads
    ctx->blocksPerCu = EnvU32("RT_BLOCKS_PER_CU", ctx->useMfma ? 1 : 4);
    if (ctx->blocksPerCu == 0) ctx->blocksPerCu = 1;
    ctx->forceGlobal = EnvU32("RT_FORCE_GLOBAL_TABLES", 0) != 0;
    ctx->blockThreads = EnvU32("RT_BLOCK_THREADS", ctx->useMfma ? 1024 : 256);
    if (ctx->blockThreads != 256 && ctx->blockThreads != 512 && ctx->blockThreads != 1024) ctx->blockThreads = 256;
    int rc = ctx->queue.Reserve(rtd::kQueueShards * rtd::kShardStrideWords);  // eight queue cursors, 128 bytes apart
    if (rc == RT_OK) rc = ctx->counters.Reserve(2);
    if (rc != RT_OK) return rc;
    {
        // sample-buffer workspace: sized for this GPU's HBM (288 GB on MI355X), so that BASELINE configs 3 (on one GPU:
        // 11.8 GB) and 5 (12.9 GB) run as ONE pass; RT_WORKSPACE_GIB or rt_set_workspace_limit override
        size_t freeB = 0, totalB = 0;
        RT_HIP(hipMemGetInfo(&freeB, &totalB));
        uint64_t lim = 64ull << 30;
        if (lim > freeB / 2) lim = freeB / 2;
        const uint32_t envGiB = EnvU32("RT_WORKSPACE_GIB", 0);
        if (envGiB) lim = (uint64_t)envGiB << 30;
        if (lim < (1ull << 20)) lim = 1ull << 20;
        ctx->workspaceLimit = lim;
    }
    *out = guard.release();
    return RT_OK;
}

void rt_destroy(rt_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->scan.Release();
    ctx->orig.Release();
    ctx->tree.Release();
    ctx->sgCells.Release();
    ctx->gridCells.Release();
    ctx->gridQ.Release();
    ctx->sgEntries.Release();
    ctx->sgGlobal.Release();
    ctx->sgSph.Release();
    for (auto& x : ctx->extraIdx) {
        x.cells.Release();
        x.entries.Release();
        x.global.Release();
    }
    ctx->lightRecs.Release();
    ctx->radius.Release();
    ctx->mats.Release();
    ctx->mats16.Release();
    ctx->hdr.Release();
    ctx->ldr.Release();
    ctx->samples.Release();
    ctx->queue.Release();
    ctx->counters.Release();
    ctx->jitterTab.Release();
    ctx->lensTab.Release();
    ctx->leaf.Release();
    ctx->pilotRays.Release();
    ctx->pilotHits.Release();
    ctx->tileOrder.Release();
    ctx->tileClass.Release();
    ctx->matType.Release();
    ctx->ring.Release();
    ctx->cont[0].Release();
    ctx->cont[1].Release();
    ctx->contN[0].Release();
    ctx->contN[1].Release();
    ctx->ctl.Release();
    for (auto& ev : ctx->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : ctx->passEv)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->ownStream) (void)hipStreamDestroy(ctx->ownStream);
    delete ctx;
}

static int BatchFlush(rt_ctx* ctx);  // frame batching (rt_set_frame_batch), next to rt_render

int rt_set_stream(rt_ctx* ctx, void* hip_stream) {
    if (!ctx) return Fail(RT_ERR_INVALID_ARG, "rt_set_stream: null ctx");
    {   // pending frames and frames in flight belong to the old stream: settle them there first
        RT_HIP(hipSetDevice(ctx->device));
        const int rcb = BatchFlush(ctx);
        if (rcb != RT_OK) return rcb;
        const int rcf = PipelineFlush(ctx);
        if (rcf != RT_OK) return rcf;
    }
    ctx->aheadValid = false;  // (planes traced ahead were produced on the old stream)
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->ownStream;
    return RT_OK;
}

int rt_set_workspace_limit(rt_ctx* ctx, uint64_t bytes) {
    if (!ctx || bytes < (1u << 20)) return Fail(RT_ERR_INVALID_ARG, "rt_set_workspace_limit: need >= 1 MiB");
    ctx->workspaceLimit = bytes;
    return RT_OK;
}

// The material table packed into 16 bytes per scan entry (rt_shade.h load_material16), or false when some material of the scene
// does not fit the form: a colour that is read (not a glass sphere's; rgb1 only under a checker texture) must be byte * (1 / 255)
// exactly -- what XMLoadColor of an XMCOLOR gives, i.e. every colour the reference can hold (texture.cpp:5,16-17).
static bool PackMaterials(const std::vector<rt_material>& matc, std::vector<uint4>& out) {
    auto byteOf = [](float c, uint32_t& b) {
        const float r = std::nearbyint(c * 255.0f);
        if (!(r >= 0.f && r <= 255.f)) return false;
        b = (uint32_t)r;
        return (float)b * (1.0f / 255.0f) == c;
    };
    out.assign(matc.size(), make_uint4(0u, 0u, 0u, 0u));
    for (size_t e = 0; e < matc.size(); ++e) {
        const rt_material& m = matc[e];
        if (m.type > 3u || m.tex_type > 1u) return false;
        uint32_t c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0};
        const bool glass = m.type == RT_MAT_DIELECTRIC_TRANSPARENT;
        for (int k = 0; k < 3; ++k) {
            if (!glass && !byteOf(m.rgb0[k], c0[k])) return false;
            if (!glass && m.tex_type == RT_TEX_CHECKER && !byteOf(m.rgb1[k], c1[k])) return false;
        }
        const float slotA = m.type == RT_MAT_EMISSIVE ? m.luminance : m.smoothness;
        const float slotB = glass ? m.ior : m.tiling;
        uint32_t a, b;
        std::memcpy(&a, &slotA, 4);
        std::memcpy(&b, &slotB, 4);
        out[e] = make_uint4(m.type | m.tex_type << 2 | c0[0] << 8 | c0[1] << 16 | c0[2] << 24, c1[0] | c1[1] << 8 | c1[2] << 16, a, b);
    }
    return true;
}

int rt_scene_upload(rt_ctx* ctx, const rt_sphere* spheres, const rt_material* materials, uint32_t n, const rt_camera* camera,
                    const rt_light* lights, uint32_t n_lights, const rt_material* sky, float exposure_scale) {
    if (!ctx || !spheres || !materials || !camera || (!lights && n_lights != 0) || !sky || n == 0)
        return Fail(RT_ERR_INVALID_ARG, "rt_scene_upload: null pointer or empty scene");
    if (n_lights > RT_MAX_LIGHTS) return Fail(RT_ERR_INVALID_ARG, "rt_scene_upload: more than RT_MAX_LIGHTS lights");
    // light 0 keeps the single-light kernel path's slots; an empty list is uploaded as one dark light that is never consulted
    const rt_light noLight{{0.f, 1.f, 0.f}, {0.f, 0.f, 0.f}, 0.f};
    const rt_light* sun = n_lights ? lights : &noLight;
    RT_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = BatchFlush(ctx)) != RT_OK) return rc;  // pending frames were asked of the scene that is being replaced
    SceneLayout L;
    BuildLayout(spheres, n, ctx->treeTop, L);
    // work lists, the shadow index and the closest-hit keys carry scan-entry ids in 16 bits
    if (L.scan.size() >= 65536) return Fail(RT_ERR_INVALID_ARG, "rt_scene_upload: scenes beyond 65,535 scan entries (about 65,000 spheres) are not supported");
    const uint32_t nPad = (uint32_t)L.scan.size();
    if ((rc = ctx->scan.Reserve(nPad)) != RT_OK) return rc;
    if ((rc = ctx->orig.Reserve(nPad)) != RT_OK) return rc;
    if ((rc = ctx->tree.Reserve(L.tree.size())) != RT_OK) return rc;
    if ((rc = ctx->leaf.Reserve(nPad)) != RT_OK) return rc;
    if ((rc = ctx->radius.Reserve(nPad)) != RT_OK) return rc;
    if ((rc = ctx->mats.Reserve(nPad)) != RT_OK) return rc;
    // radius and material of every scan entry, in clustered order: the hit processing indexes them with the entry it
    // found, with no detour through the original index (one dependent load less; material i still belongs to sphere i)
    std::vector<float> rad(nPad, 0.f);
    std::vector<rt_material> matc(nPad);
    std::memset(matc.data(), 0, nPad * sizeof(rt_material));
    for (uint32_t e = 0; e < nPad; ++e) {
        if (L.orig[e] == 0xffffffffu) continue;
        rad[e] = spheres[L.orig[e]].r;
        matc[e] = materials[L.orig[e]];
    }
    RT_HIP(hipStreamSynchronize(ctx->stream));
    RT_HIP(hipMemcpy(ctx->scan.ptr, L.scan.data(), nPad * sizeof(float4), hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(ctx->orig.ptr, L.orig.data(), nPad * sizeof(uint32_t), hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(ctx->tree.ptr, L.tree.data(), L.tree.size() * sizeof(float4), hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(ctx->leaf.ptr, L.leaf.data(), nPad * sizeof(float4), hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(ctx->radius.ptr, rad.data(), nPad * sizeof(float), hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(ctx->mats.ptr, matc.data(), nPad * sizeof(rt_material), hipMemcpyHostToDevice));
    {
        std::vector<uint4> m16;
        ctx->mats16Ok = PackMaterials(matc, m16);
        if (ctx->mats16Ok) {
            if ((rc = ctx->mats16.Reserve(nPad)) != RT_OK) return rc;
            RT_HIP(hipMemcpy(ctx->mats16.ptr, m16.data(), nPad * sizeof(uint4), hipMemcpyHostToDevice));
        }
    }
    {   // material type by ORIGINAL sphere index: what rt_tile_order_kernel classifies the pilot rays' first hits by
        std::vector<uint32_t> types(n);
        for (uint32_t k = 0; k < n; ++k) types[k] = materials[k].type;
        if ((rc = ctx->matType.Reserve(n)) != RT_OK) return rc;
        RT_HIP(hipMemcpy(ctx->matType.ptr, types.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
        ctx->tileOrderValid = false;
    }

    ShadowGrid SG;
    if (ctx->useShadowGrid && n_lights != 0) BuildShadowGrid(spheres, L, sun->direction, (L.gridOn || L.nLevels > 1) ? EnvU32("RT_SHADOW_CELLS", 256u) : 64u, SG);
    // lights 1 .. : one index each, in global memory (256 x 256 cells at most), and their records
    std::vector<rtd::LightRec> recs;
    for (uint32_t k = 1; k < n_lights; ++k) {
        ShadowGrid G2;
        if (ctx->useShadowGrid) BuildShadowGrid(spheres, L, lights[k].direction, EnvU32("RT_SHADOW_CELLS", 256u), G2);
        rt_ctx::ExtraLight& X = ctx->extraIdx[k];
        rtd::LightRec R{};
        for (int c = 0; c < 3; ++c) {
            R.sun_dir[c] = lights[k].direction[c];
            R.sun_rad[c] = lights[k].luminance * lights[k].color[c];  // m_luminance * m_color, light.cpp:27
            R.cam_o[c] = camera->origin[c];
        }
        R.sg_enabled = G2.enabled ? 1u : 0u;
        if (G2.enabled) {
            if ((rc = X.cells.Reserve(G2.cellStart.size())) != RT_OK) return rc;
            if ((rc = X.entries.Reserve(G2.entries.size() + 1)) != RT_OK) return rc;
            if ((rc = X.global.Reserve(G2.global.size() + 1)) != RT_OK) return rc;
            RT_HIP(hipMemcpy(X.cells.ptr, G2.cellStart.data(), G2.cellStart.size() * 2, hipMemcpyHostToDevice));
            if (!G2.entries.empty()) RT_HIP(hipMemcpy(X.entries.ptr, G2.entries.data(), G2.entries.size() * 2, hipMemcpyHostToDevice));
            if (!G2.global.empty()) RT_HIP(hipMemcpy(X.global.ptr, G2.global.data(), G2.global.size() * 2, hipMemcpyHostToDevice));
            for (int c = 0; c < 3; ++c) {
                R.sg_e1[c] = G2.e1[c];
                R.sg_e2[c] = G2.e2[c];
            }
            R.sg_u0 = G2.u0; R.sg_v0 = G2.v0; R.sg_inv_cell = G2.invCell; R.sg_p0sq = G2.p0sq;
            R.sg_nx = G2.nx; R.sg_ny = G2.ny; R.sg_nglobal = (uint32_t)G2.global.size();
            R.cell_start = X.cells.ptr; R.entries = X.entries.ptr; R.global = X.global.ptr;
        }
        recs.push_back(R);
    }
    if (!recs.empty()) {
        if ((rc = ctx->lightRecs.Reserve(recs.size())) != RT_OK) return rc;
        RT_HIP(hipMemcpy(ctx->lightRecs.ptr, recs.data(), recs.size() * sizeof(rtd::LightRec), hipMemcpyHostToDevice));
    }
    if (SG.enabled) {
        if ((rc = ctx->sgCells.Reserve(SG.cellStart.size())) != RT_OK) return rc;
        if ((rc = ctx->sgEntries.Reserve(SG.entries.size() + 1)) != RT_OK) return rc;
        if ((rc = ctx->sgGlobal.Reserve(SG.global.size() + 1)) != RT_OK) return rc;
        RT_HIP(hipMemcpy(ctx->sgCells.ptr, SG.cellStart.data(), SG.cellStart.size() * 2, hipMemcpyHostToDevice));
        if (!SG.entries.empty()) RT_HIP(hipMemcpy(ctx->sgEntries.ptr, SG.entries.data(), SG.entries.size() * 2, hipMemcpyHostToDevice));
        if (!SG.global.empty()) RT_HIP(hipMemcpy(ctx->sgGlobal.ptr, SG.global.data(), SG.global.size() * 2, hipMemcpyHostToDevice));
        // RT_SG_SPH=1 (experiments): the index stays in global memory -- the sphere record of every entry side by side with the ids, so
        // that a walk round is one round trip instead of two.  Measured on grid10k: 7.79 -> 7.66 Gsamples/s (700 KB of duplicated
        // spheres hit the L1 less often than the 160 KB table they are shared from).  Default off.
        if ((L.gridOn || L.nLevels > 1) && EnvU32("RT_SG_SPH", 0u) != 0u) {
            std::vector<float4> sph(SG.entries.size() + 1, make_float4(0.f, 0.f, 0.f, -1e30f));
            for (size_t k = 0; k < SG.entries.size(); ++k) sph[k] = L.scan[SG.entries[k]];
            if ((rc = ctx->sgSph.Reserve(sph.size())) != RT_OK) return rc;
            RT_HIP(hipMemcpy(ctx->sgSph.ptr, sph.data(), sph.size() * sizeof(float4), hipMemcpyHostToDevice));
        }
    }

    if (L.gridOn) {
        if ((rc = ctx->gridCells.Reserve(L.gridCellStart.size())) != RT_OK) return rc;
        RT_HIP(hipMemcpy(ctx->gridCells.ptr, L.gridCellStart.data(), L.gridCellStart.size() * 2, hipMemcpyHostToDevice));
        if (!L.gridQ.empty()) {
            if ((rc = ctx->gridQ.Reserve(L.gridQ.size())) != RT_OK) return rc;
            RT_HIP(hipMemcpy(ctx->gridQ.ptr, L.gridQ.data(), L.gridQ.size() * 4, hipMemcpyHostToDevice));
        }
    }

    rtd::TraceParams& b = ctx->base;
    b = rtd::TraceParams{};
    if (L.gridOn) {
        b.grid_cell_start = ctx->gridCells.ptr;
        b.grid_nu = L.gridNu;
        b.grid_nv = L.gridNv;
        b.grid_ax_u = L.gridAxU;
        b.grid_ax_v = L.gridAxV;
        b.grid_g0u = L.gridG0u;
        b.grid_g0v = L.gridG0v;
        b.grid_inv_h = L.gridInvH;
        b.grid_rmax_over_h = L.gridRmaxOverH;
        b.grid_big_norm = L.gridBigNorm;
        b.grid_qrec = L.gridQ.empty() ? nullptr : ctx->gridQ.ptr;
        for (int c = 0; c < 8; ++c) b.grid_q[c] = L.gridQc[c];
    }
    b.n_lights = n_lights;
    b.extra_lights = recs.empty() ? nullptr : ctx->lightRecs.ptr;
    b.sg_enabled = SG.enabled ? 1u : 0u;
    if (SG.enabled) {
        b.sg_cell_start = ctx->sgCells.ptr;
        b.sg_entries = ctx->sgEntries.ptr;
        b.sg_global = ctx->sgGlobal.ptr;
        b.sg_sph = ((L.gridOn || L.nLevels > 1) && EnvU32("RT_SG_SPH", 0u) != 0u) ? ctx->sgSph.ptr : nullptr;
        b.sg_nx = SG.nx;
        b.sg_ny = SG.ny;
        b.sg_nglobal = (uint32_t)SG.global.size();
        b.sg_nentries = (uint32_t)SG.entries.size();
        for (int k = 0; k < 3; ++k) {
            b.sg_e1[k] = SG.e1[k];
            b.sg_e2[k] = SG.e2[k];
        }
        b.sg_u0 = SG.u0;
        b.sg_v0 = SG.v0;
        b.sg_inv_cell = SG.invCell;
        b.sg_p0sq = SG.p0sq;
    }
    b.scan = ctx->scan.ptr;
    b.orig = ctx->orig.ptr;
    b.leaf = ctx->leaf.ptr;
    b.tree = ctx->tree.ptr;
    b.n_groups = L.nGroups;
    b.n_levels = L.nLevels;
    for (uint32_t k = 0; k < rtd::kMaxLevels; ++k) {
        b.level_off[k] = L.levelOff[k];
        b.level_cnt[k] = L.levelCnt[k];
    }
    b.bound_norm = L.boundNorm;
    b.n_always = L.nAlways;
    for (int k = 0; k < 9; ++k) b.tree_box[k] = L.treeBox[k];
    b.tree_box_on = L.treeBoxOn ? 1u : 0u;
    b.single_mask[0] = std::getenv("RT_SINGLE_DIRECT") && std::atoi(std::getenv("RT_SINGLE_DIRECT")) == 0 ? 0ull : L.singleMask[0];
    b.single_mask[1] = std::getenv("RT_SINGLE_DIRECT") && std::atoi(std::getenv("RT_SINGLE_DIRECT")) == 0 ? 0ull : L.singleMask[1];
    b.radius = ctx->radius.ptr;
    b.mats = ctx->mats.ptr;
    b.mats16 = ctx->mats16Ok ? ctx->mats16.ptr : nullptr;
    b.n = n;
    b.n_padded = nPad;
    for (int k = 0; k < 3; ++k) {
        b.cam_o[k] = camera->origin[k];
        b.cam_x[k] = camera->x[k];
        b.cam_y[k] = camera->y[k];
        b.cam_oip[k] = camera->origin_image_plane[k];
        b.sun_dir[k] = sun->direction[k];
        b.sun_rad[k] = sun->luminance * sun->color[k];  // m_luminance * m_color, light.cpp:27
        b.sky_emit[k] = sky->luminance * sky->rgb0[k];  // Emissive::Emit, material.cpp:172-175
    }
    b.aperture = camera->aperture;
    b.focal = camera->focal_length;
    b.exposure = exposure_scale;
    ctx->n = n;
    ctx->hasScene = true;
    ctx->aheadValid = false;
    ctx->accumulated = 0;
    PipelineDrop(ctx);
    return RT_OK;
}

int rt_set_frame_pipelining(rt_ctx* ctx, uint32_t depth) {
    if (!ctx || depth > rtd::kMaxFramesInFlight - 2u) return Fail(RT_ERR_INVALID_ARG, "rt_set_frame_pipelining: depth must be 0..14");
    RT_HIP(hipSetDevice(ctx->device));
    int rc = BatchFlush(ctx);
    if (rc != RT_OK) return rc;
    rc = PipelineFlush(ctx);
    ctx->pipeDepth = depth;
    return rc;
}

int rt_committed_samples(rt_ctx* ctx, uint32_t* out) {
    if (!ctx || !out) return Fail(RT_ERR_INVALID_ARG, "rt_committed_samples: null argument");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    *out = ctx->accumulated;
    if (ctx->pipeOpen) {
        rtd::FrameCtl c{};
        RT_HIP(hipMemcpy(&c, ctx->ctl.ptr, sizeof(c), hipMemcpyDeviceToHost));
        *out = c.committed_samples[ctx->pipeCommits & 1u];
    }
    return RT_OK;
}

int rt_set_sampler(rt_ctx* ctx, uint32_t flags) {
    if (!ctx || (flags & ~(RT_SAMPLER_COSINE_HEMISPHERE | RT_SAMPLER_SQRT_DISK)) != 0u) return Fail(RT_ERR_INVALID_ARG, "rt_set_sampler: unknown flag");
    if (flags != ctx->sampler) {
        ctx->pendOn = false;   // (pending frames of the old mapping are dropped with the accumulation they belonged to)
        ctx->accumulated = 0;  // samples of two mappings do not mix: the next rt_render starts over
        PipelineDrop(ctx);
        ctx->tileOrderValid = false;  // the pilot rays use the lens mapping
    }
    ctx->sampler = flags;
    return RT_OK;
}

int rt_clear(rt_ctx* ctx) {
    if (!ctx) return Fail(RT_ERR_INVALID_ARG, "rt_clear: null ctx");
    ctx->pendOn = false;
    ctx->aheadValid = false;
    PipelineDrop(ctx);
    ctx->accumulated = 0;
    return RT_OK;
}

uint32_t rt_rowset_local_rows(rt_rowset rs) { return RowsetLocalRows(rs); }
uint32_t rt_rowset_global_row(rt_rowset rs, uint32_t lr) {
    const uint32_t lb = lr / rs.block_rows;
    const uint32_t k = lr % rs.block_rows;
    return rs.first_row + (lb * rs.nshards + rs.shard) * rs.block_rows + k;
}

static int RenderNow(rt_ctx* ctx, uint32_t W, uint32_t H, rt_rowset rs, uint32_t s0, uint32_t s1, uint32_t max_depth, uint64_t seed,
                     rt_stats* out_stats, uint32_t aheadEnd = 0);

// Frame batching: render the pending sample planes with ONE launch (no statistics, nothing waits for the device).
static int BatchFlush(rt_ctx* ctx) {
    if (!ctx->pendOn) return RT_OK;
    ctx->pendOn = false;
    return RenderNow(ctx, ctx->pendW, ctx->pendH, ctx->pendRs, ctx->pendS0, ctx->pendS1, ctx->pendDepth, ctx->pendSeed, nullptr);
}

// The readers (rt_resolve, rt_download, rt_copy_to_device) hand out the COMMITTED strip while a batch that continues it is
// pending.  A pending batch that does not continue it -- nothing is committed yet, or its first call had s0 == 1 and so starts a
// new accumulation, possibly of another size: the caller's picture is the new one, and ctx->W / ctx->rows still describe the old
// strip -- is rendered first.
static bool PendingMustRender(const rt_ctx* ctx) { return ctx->pendOn && (ctx->accumulated == 0 || ctx->pendS0 == 1u); }

int rt_set_frame_batch(rt_ctx* ctx, uint32_t frames) {
    if (!ctx || frames == 0 || frames > 4096) return Fail(RT_ERR_INVALID_ARG, "rt_set_frame_batch: frames must be 1..4096");
    RT_HIP(hipSetDevice(ctx->device));
    const int rc = BatchFlush(ctx);
    ctx->batchFrames = frames;
    return rc;
}

int rt_set_frame_lookahead(rt_ctx* ctx, uint32_t frames) {
    if (!ctx || frames == 0 || frames > 4096) return Fail(RT_ERR_INVALID_ARG, "rt_set_frame_lookahead: frames must be 1..4096");
    ctx->lookahead = frames;
    ctx->aheadValid = false;
    return RT_OK;
}

int rt_render(rt_ctx* ctx, uint32_t W, uint32_t H, rt_rowset rs, uint32_t s0, uint32_t s1, uint32_t max_depth, uint64_t seed,
              rt_stats* out_stats) {
    if (!ctx) return Fail(RT_ERR_INVALID_ARG, "rt_render: null ctx");
    if (ctx->lookahead > 1 && ctx->batchFrames <= 1 && ctx->pipeDepth == 0 && out_stats == nullptr && ctx->hasScene && W != 0 && H != 0 && s0 != 0 && s1 > s0) {
        // render-ahead: the planes of this call may have been traced by an earlier call's launch -- then they are only ADDED
        // (in sample order, as always); else this call's launch traces its own planes and the next `lookahead` ones with them
        if (ctx->aheadValid && ctx->aheadW == W && ctx->aheadH == H && std::memcmp(&ctx->aheadRs, &rs, sizeof(rs)) == 0 && ctx->aheadDepth == max_depth &&
            ctx->aheadSeed == seed && s0 == ctx->aheadNext && s1 <= ctx->aheadBase + ctx->aheadSpp && ctx->accumulated + 1 == s0) {
            RT_HIP(hipSetDevice(ctx->device));
            const uint32_t npix = ctx->W * ctx->rows;
            hipLaunchKernelGGL(rtd::rt_accumulate_kernel, dim3((npix + 255) / 256), dim3(256), 0, ctx->stream, ctx->samples.ptr, ctx->hdr.ptr, npix,
                               ctx->aheadSpp, s0 - ctx->aheadBase, s1 - s0);
            RT_HIP(hipGetLastError());
            ctx->aheadNext = s1;
            ctx->accumulated += s1 - s0;
            return RT_OK;
        }
        const uint32_t want = s1 - s0 > ctx->lookahead ? s1 - s0 : ctx->lookahead;
        return RenderNow(ctx, W, H, rs, s0, s1, max_depth, seed, nullptr, s0 + want);
    }
    if (ctx->batchFrames > 1 && out_stats == nullptr && ctx->pipeDepth == 0 && ctx->hasScene && W != 0 && H != 0 && s0 != 0 && s1 > s0) {
        // a pending batch this call does not continue is rendered first, as the calls were made
        if (ctx->pendOn && !(ctx->pendW == W && ctx->pendH == H && std::memcmp(&ctx->pendRs, &rs, sizeof(rs)) == 0 && ctx->pendDepth == max_depth &&
                             ctx->pendSeed == seed && ctx->pendS1 == s0)) {
            const int rcf = BatchFlush(ctx);
            if (rcf != RT_OK) return rcf;
        }
        if (!ctx->pendOn) {
            // only a call that RenderNow would accept as the start or the continuation of an accumulation is deferred
            const uint32_t rows = RowsetLocalRows(rs);
            const bool sameStrip = ctx->W == W && ctx->H == H && ctx->rows == rows && std::memcmp(&ctx->rs, &rs, sizeof(rs)) == 0;
            const bool starts = s0 == 1, continues = ctx->accumulated != 0 && sameStrip && s0 == ctx->accumulated + 1;
            if (rows != 0 && (uint64_t)rs.first_row + rs.num_rows <= H && (starts || continues)) {
                ctx->pendOn = true;
                ctx->pendW = W; ctx->pendH = H; ctx->pendRs = rs; ctx->pendS0 = s0; ctx->pendDepth = max_depth; ctx->pendSeed = seed;
            }
        }
        if (ctx->pendOn) {
            ctx->pendS1 = s1;
            if (ctx->pendS1 - ctx->pendS0 >= ctx->batchFrames) return BatchFlush(ctx);
            return RT_OK;
        }
    }
    const int rcb = BatchFlush(ctx);  // a call with statistics (or one that cannot be deferred) settles the pending batch first
    if (rcb != RT_OK) return rcb;
    return RenderNow(ctx, W, H, rs, s0, s1, max_depth, seed, out_stats);
}

// aheadEnd (render-ahead, rt_set_frame_lookahead): trace the planes [s0, aheadEnd) with this call's ONE launch, add only [s0, s1)
// to the strip and keep the rest in the sample buffer for the calls that continue (0: trace [s0, s1) only).
static int RenderNow(rt_ctx* ctx, uint32_t W, uint32_t H, rt_rowset rs, uint32_t s0, uint32_t s1, uint32_t max_depth, uint64_t seed,
                     rt_stats* out_stats, uint32_t aheadEnd) {
    ctx->aheadValid = false;  // whatever was traced ahead belongs to the calls before this one
    if (!ctx->hasScene) return Fail(RT_ERR_NO_SCENE, "rt_render: no scene uploaded");
    if (W == 0 || H == 0 || s0 == 0 || s1 <= s0) return Fail(RT_ERR_INVALID_ARG, "rt_render: empty image or sample range");
    const uint32_t rows = RowsetLocalRows(rs);
    if (rows == 0 || (uint64_t)rs.first_row + rs.num_rows > H) return Fail(RT_ERR_INVALID_ARG, "rt_render: bad row set");
    const uint64_t npix64 = (uint64_t)W * rows;
    if (npix64 > (1ull << 31) || (uint64_t)W * H > 0xffffffffull) return Fail(RT_ERR_INVALID_ARG, "rt_render: image too large");
    RT_HIP(hipSetDevice(ctx->device));
    const uint32_t npix = (uint32_t)npix64;

    const bool sameStrip = ctx->W == W && ctx->H == H && ctx->rows == rows && std::memcmp(&ctx->rs, &rs, sizeof(rs)) == 0;
    // frame pipelining: only calls that ask for no statistics may leave work in flight; anything else first settles it
    bool pipelined = ctx->pipeDepth > 0 && out_stats == nullptr;
    if (pipelined) {
        rtd::TraceParams probe = ctx->base;
        probe.total_paths = npix;
        pipelined = LaunchTrace(ctx, probe, 1) == RT_OK && (uint64_t)npix * (s1 - s0) * (ctx->pipeDepth + 2u) < (1ull << 31);
    }
    if (s0 == 1 || ctx->accumulated == 0) PipelineDrop(ctx);  // a new accumulation abandons whatever was in flight
    else if (!pipelined) {
        int rcf = PipelineFlush(ctx);
        if (rcf != RT_OK) return rcf;
    }
    if (s0 == 1 || ctx->accumulated == 0) {
        if (s0 != 1) return Fail(RT_ERR_SEQUENCE, "rt_render: first call of an accumulation must start at s0 == 1");
        int rc;
        if ((rc = ctx->hdr.Reserve((size_t)npix * 3)) != RT_OK) return rc;
        if ((rc = ctx->ldr.Reserve((size_t)npix * 3)) != RT_OK) return rc;
        RT_HIP(hipMemsetAsync(ctx->hdr.ptr, 0, (size_t)npix * 3 * sizeof(float), ctx->stream));  // app.cpp:112-119
        ctx->W = W;
        ctx->H = H;
        ctx->rows = rows;
        ctx->rs = rs;
        ctx->accumulated = 0;
        if ((rc = BuildTileOrder(ctx, W, H, rs, npix)) != RT_OK) return rc;
    } else if (!sameStrip || s0 != ctx->accumulated + 1) {
        return Fail(RT_ERR_SEQUENCE, "rt_render: sample range or row set does not continue the accumulation");
    }

    if (pipelined) {
        int rcp = PipelineRender(ctx, W, H, rs, npix, s0, s1, max_depth, seed);
        if (rcp != RT_OK) {
            ctx->accumulated = 0;
            PipelineDrop(ctx);
            return rcp;
        }
        ctx->accumulated += s1 - s0;
        return RT_OK;
    }

    // split [s0, s1) into passes whose sample buffer fits the workspace limit
    const uint64_t bytesPerSpp = (uint64_t)npix * 12;
    uint64_t sppMax = ctx->workspaceLimit / bytesPerSpp;
    const uint64_t sppPathCap = ((1ull << 31) - 1) / npix;  // total_paths stays below 2^31
    if (sppMax > sppPathCap) sppMax = sppPathCap;
    if (sppMax == 0) return Fail(RT_ERR_OUT_OF_MEMORY, "rt_render: workspace limit below one sample per pixel");
    const uint32_t sppTotal = s1 - s0;
    if (aheadEnd <= s1 || (uint64_t)(aheadEnd - s0) > sppMax || pipelined) aheadEnd = 0;  // the planes traced ahead must share one pass
    const uint32_t sppTrace = aheadEnd ? aheadEnd - s0 : sppTotal;
    uint32_t sppPass = (uint32_t)(sppMax < sppTrace ? sppMax : sppTrace);
    int rc;
    // The default limit is a snapshot of the free memory at rt_create; another context or the caller's allocator may have
    // taken memory since.  A smaller sample buffer only means more passes (bit-identical: the accumulation stays sequential
    // in s), so halve the pass until the buffer fits before giving up.
    while ((rc = ctx->samples.Reserve((size_t)npix * sppPass * 3)) != RT_OK) {
        if (rc != RT_ERR_OUT_OF_MEMORY || sppPass == 1u) return rc;
        sppPass = (sppPass + 1u) / 2u;
    }

    // Passes run back to back on the stream: trace(k) -> accumulate(k) -> trace(k+1) are ordered by the stream itself
    // (they share the sample buffer), so the host never waits between passes; each pass has its own three timing
    // events, read once at the end when the caller asked for statistics.
    uint32_t passes = 0;
    auto runPasses = [&]() -> int {
        int rc;
        RT_HIP(hipMemsetAsync(ctx->counters.ptr, 0, 2 * sizeof(unsigned long long), ctx->stream));
        const uint32_t sEnd = aheadEnd ? aheadEnd : s1;
        for (uint32_t s = s0; s < sEnd; s += sppPass) {
            const uint32_t spp = (sEnd - s) < sppPass ? (sEnd - s) : sppPass;
            rtd::TraceParams tp = ctx->base;
            tp.W = W;
            tp.H = H;
            tp.rs = rs;
            tp.s0 = s;
            tp.spp_pass = spp;
            tp.total_paths = npix * spp;
            tp.npix_local = npix;
            tp.max_depth = max_depth;
            tp.sampler = ctx->sampler;
            tp.seed = seed;
            tp.path_list = nullptr;
            tp.tile_order = ctx->tileOrderValid ? ctx->tileOrder.ptr : nullptr;
            tp.samples = ctx->samples.ptr;
            tp.trav_out = nullptr;
            tp.counters = ctx->counters.ptr;
            // ray-generation tables for this pass: s in [s, s+spp), k = s+i+j over the strip's rows
            const uint32_t k0 = s + rs.first_row;
            const uint32_t nLens = spp + W + rs.num_rows;
            if ((rc = ctx->jitterTab.Reserve(spp)) != RT_OK) return rc;
            if ((rc = ctx->lensTab.Reserve(nLens)) != RT_OK) return rc;
            tp.jitter_tab = ctx->jitterTab.ptr;
            tp.lens_tab = ctx->lensTab.ptr;
            tp.lens_k0 = k0;
            while (ctx->passEv.size() < 3 * (size_t)(passes + 1)) {
                hipEvent_t e = nullptr;
                RT_HIP(hipEventCreate(&e));
                ctx->passEv.push_back(e);
            }
            hipEvent_t* ev = ctx->passEv.data() + 3 * (size_t)passes;
            RT_HIP(hipEventRecord(ev[0], ctx->stream));
            {
                const uint32_t nmax = spp > nLens ? spp : nLens;
                hipLaunchKernelGGL(rtd::rt_raygen_tables_kernel, dim3((nmax + 255) / 256), dim3(256), 0, ctx->stream, ctx->jitterTab.ptr, s,
                                   spp, ctx->lensTab.ptr, k0, nLens, ctx->sampler);
                RT_HIP(hipGetLastError());
            }
            if ((rc = LaunchTrace(ctx, tp)) != RT_OK) return rc;
            RT_HIP(hipEventRecord(ev[1], ctx->stream));
            hipLaunchKernelGGL(rtd::rt_accumulate_kernel, dim3((npix + 255) / 256), dim3(256), 0, ctx->stream, ctx->samples.ptr,
                               ctx->hdr.ptr, npix, spp, 0u, aheadEnd ? sppTotal : spp);
            RT_HIP(hipGetLastError());
            RT_HIP(hipEventRecord(ev[2], ctx->stream));
            ++passes;
        }
        return RT_OK;
    };
    if ((rc = runPasses()) != RT_OK) {
        // some passes may already have been added to hdr: the accumulation is void, the next call must restart at s0 == 1
        ctx->accumulated = 0;
        return rc;
    }
    ctx->accumulated += sppTotal;
    if (aheadEnd && ctx->samples.ptr && passes == 1) {  // the planes [s1, aheadEnd) wait in the sample buffer
        ctx->aheadValid = true;
        ctx->aheadW = W; ctx->aheadH = H; ctx->aheadRs = rs; ctx->aheadDepth = max_depth; ctx->aheadSeed = seed;
        ctx->aheadBase = s0; ctx->aheadSpp = sppTrace; ctx->aheadNext = s1;
    }

    if (out_stats) {
        // Without a stats request the call stays asynchronous on the stream (progressive 1-spp frames are launch bound:
        // ~0.2 ms of GPU work each); with one it waits for the last pass and reads the event timers.
        float msTrace = 0.f, msAcc = 0.f;
        RT_HIP(hipEventSynchronize(ctx->passEv[3 * (size_t)(passes - 1) + 2]));
        for (uint32_t k = 0; k < passes; ++k) {
            float a = 0.f, b = 0.f;
            RT_HIP(hipEventElapsedTime(&a, ctx->passEv[3 * (size_t)k], ctx->passEv[3 * (size_t)k + 1]));
            RT_HIP(hipEventElapsedTime(&b, ctx->passEv[3 * (size_t)k + 1], ctx->passEv[3 * (size_t)k + 2]));
            msTrace += a;
            msAcc += b;
        }
        unsigned long long c[2] = {0, 0};
        RT_HIP(hipMemcpy(c, ctx->counters.ptr, sizeof(c), hipMemcpyDeviceToHost));
        std::memset(out_stats, 0, sizeof(*out_stats));
        out_stats->samples = (uint64_t)npix * sppTotal;
        out_stats->traversals = c[0];
        out_stats->segments = c[1];
        out_stats->ms_render = msTrace;
        out_stats->ms_accumulate = msAcc;
        out_stats->ms_resolve = 0.0;
        out_stats->local_rows = rows;
        out_stats->passes = passes;
    }
    return RT_OK;
}

int rt_resolve(rt_ctx* ctx, uint32_t n_samples) {
    if (!ctx) return Fail(RT_ERR_INVALID_ARG, "rt_resolve: null ctx");
    if (PendingMustRender(ctx)) {  // nothing committed yet, or the pending frames START a new accumulation: they are what there is to show
        const int rcb = BatchFlush(ctx);
        if (rcb != RT_OK) return rcb;
    }
    if (ctx->accumulated == 0) return Fail(RT_ERR_SEQUENCE, "rt_resolve: nothing accumulated");
    RT_HIP(hipSetDevice(ctx->device));
    const uint32_t n = n_samples ? n_samples : ctx->accumulated;
    const uint32_t npix = ctx->W * ctx->rows;
    RT_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    // with frames in flight the strip holds FrameCtl::committed_samples samples, a number only the device knows
    const uint32_t* devCount = (ctx->pipeOpen && n_samples == 0) ? &ctx->ctl.ptr->committed_samples[ctx->pipeCommits & 1u] : nullptr;
    hipLaunchKernelGGL(rtd::rt_resolve_kernel, dim3((npix + 255) / 256), dim3(256), 0, ctx->stream, ctx->hdr.ptr, ctx->ldr.ptr, npix, n,
                       devCount);
    RT_HIP(hipGetLastError());
    RT_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
    RT_HIP(hipEventSynchronize(ctx->ev[1]));
    float ms = 0.f;
    RT_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->lastResolveMs = ms;
    return RT_OK;
}

double rt_last_resolve_ms(rt_ctx* ctx) { return ctx ? ctx->lastResolveMs : 0.0; }

int rt_download(rt_ctx* ctx, float* hdr_rgb, uint8_t* ldr_rgb) {
    if (!ctx) return Fail(RT_ERR_INVALID_ARG, "rt_download: null ctx");
    if (PendingMustRender(ctx)) {
        const int rcb = BatchFlush(ctx);
        if (rcb != RT_OK) return rcb;
    }
    if (ctx->accumulated == 0) return Fail(RT_ERR_SEQUENCE, "rt_download: nothing rendered");
    RT_HIP(hipSetDevice(ctx->device));
    const size_t npix = (size_t)ctx->W * ctx->rows;
    RT_HIP(hipStreamSynchronize(ctx->stream));
    if (hdr_rgb) RT_HIP(hipMemcpy(hdr_rgb, ctx->hdr.ptr, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (ldr_rgb) RT_HIP(hipMemcpy(ldr_rgb, ctx->ldr.ptr, npix * 3, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_copy_to_device(rt_ctx* ctx, void* dev_hdr_rgb, void* dev_ldr_rgb) {
    if (!ctx) return Fail(RT_ERR_INVALID_ARG, "rt_copy_to_device: null ctx");
    if (PendingMustRender(ctx)) {
        const int rcb = BatchFlush(ctx);
        if (rcb != RT_OK) return rcb;
    }
    if (ctx->accumulated == 0) return Fail(RT_ERR_SEQUENCE, "rt_copy_to_device: nothing rendered");
    RT_HIP(hipSetDevice(ctx->device));
    const size_t npix = (size_t)ctx->W * ctx->rows;
    if (dev_hdr_rgb) RT_HIP(hipMemcpyAsync(dev_hdr_rgb, ctx->hdr.ptr, npix * 3 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    if (dev_ldr_rgb) RT_HIP(hipMemcpyAsync(dev_ldr_rgb, ctx->ldr.ptr, npix * 3, hipMemcpyDeviceToDevice, ctx->stream));
    return RT_OK;
}

int rt_synchronize(rt_ctx* ctx) {
    if (!ctx) return Fail(RT_ERR_INVALID_ARG, "rt_synchronize: null ctx");
    RT_HIP(hipSetDevice(ctx->device));
    int rcf = BatchFlush(ctx);  // pending frames are rendered,
    if (rcf != RT_OK) return rcf;
    rcf = PipelineFlush(ctx);  // frames in flight are finished and committed first
    if (rcf != RT_OK) return rcf;
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

// ------------------------------------------------------------------------ unit entries
int rt_unit_halton(rt_ctx* ctx, const uint32_t* index, uint32_t base, uint32_t n, float* out) {
    if (!ctx || !index || !out || base < 2) return Fail(RT_ERR_INVALID_ARG, "rt_unit_halton: invalid argument");
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    TmpDev<uint32_t> dIdx;
    TmpDev<float> dOut;
    RT_HIP(dIdx.Alloc(n));
    RT_HIP(dOut.Alloc(n));
    RT_HIP(hipMemcpy(dIdx.p, index, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rtd::k_unit_halton, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, dIdx.p, base, n, dOut.p);
    RT_HIP(hipGetLastError());
    RT_HIP(hipStreamSynchronize(ctx->stream));
    RT_HIP(hipMemcpy(out, dOut.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_unit_math(rt_ctx* ctx, uint32_t op, const float* x, const float* y, uint32_t n, float* out) {
    if (!ctx || !x || !out || op > 9) return Fail(RT_ERR_INVALID_ARG, "rt_unit_math: invalid argument");
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    TmpDev<float> dX, dY, dOut;
    RT_HIP(dX.Alloc(n));
    RT_HIP(dY.Alloc(n));
    RT_HIP(dOut.Alloc(n));
    RT_HIP(hipMemcpy(dX.p, x, n * sizeof(float), hipMemcpyHostToDevice));
    if (y) RT_HIP(hipMemcpy(dY.p, y, n * sizeof(float), hipMemcpyHostToDevice));
    else RT_HIP(hipMemset(dY.p, 0, n * sizeof(float)));
    hipLaunchKernelGGL(rtd::k_unit_math, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, op, dX.p, dY.p, n, dOut.p);
    RT_HIP(hipGetLastError());
    RT_HIP(hipStreamSynchronize(ctx->stream));
    RT_HIP(hipMemcpy(out, dOut.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_unit_primary_rays(rt_ctx* ctx, uint32_t W, uint32_t H, const uint32_t* ijs, uint32_t n, float* out_rays) {
    if (!ctx || !ijs || !out_rays || W == 0 || H == 0) return Fail(RT_ERR_INVALID_ARG, "rt_unit_primary_rays: invalid argument");
    if (!ctx->hasScene) return Fail(RT_ERR_NO_SCENE, "rt_unit_primary_rays: no scene uploaded");
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    TmpDev<uint32_t> dIjs;
    TmpDev<float> dOut;
    RT_HIP(dIjs.Alloc((size_t)n * 3));
    RT_HIP(dOut.Alloc((size_t)n * 6));
    RT_HIP(hipMemcpy(dIjs.p, ijs, (size_t)n * 3 * sizeof(uint32_t), hipMemcpyHostToDevice));
    rtd::TraceParams tp = ctx->base;
    tp.W = W;
    tp.H = H;
    tp.sampler = ctx->sampler;
    hipLaunchKernelGGL(rtd::k_unit_primary, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, tp, dIjs.p, n, dOut.p);
    RT_HIP(hipGetLastError());
    RT_HIP(hipStreamSynchronize(ctx->stream));
    RT_HIP(hipMemcpy(out_rays, dOut.p, (size_t)n * 6 * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_unit_closest_hit(rt_ctx* ctx, const float* rays, uint32_t n, float* out_hits) {
    if (!ctx || !rays || !out_hits) return Fail(RT_ERR_INVALID_ARG, "rt_unit_closest_hit: invalid argument");
    if (!ctx->hasScene) return Fail(RT_ERR_NO_SCENE, "rt_unit_closest_hit: no scene uploaded");
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    TmpDev<float> dRays, dOut;
    RT_HIP(dRays.Alloc((size_t)n * 6));
    RT_HIP(dOut.Alloc((size_t)n * 10));
    RT_HIP(hipMemcpy(dRays.p, rays, (size_t)n * 6 * sizeof(float), hipMemcpyHostToDevice));
    {
        const int rcl = LaunchClosest(ctx, dRays.p, n, dOut.p);
        if (rcl != RT_OK) return rcl;
    }
    RT_HIP(hipGetLastError());
    RT_HIP(hipStreamSynchronize(ctx->stream));
    RT_HIP(hipMemcpy(out_hits, dOut.p, (size_t)n * 10 * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_unit_trace(rt_ctx* ctx, uint32_t W, uint32_t H, const uint32_t* ijs, uint32_t n, uint32_t max_depth, uint64_t seed,
                  float* out_rgb, uint32_t* out_traversals) {
    if (!ctx || !ijs || !out_rgb || W == 0 || H == 0) return Fail(RT_ERR_INVALID_ARG, "rt_unit_trace: invalid argument");
    if (!ctx->hasScene) return Fail(RT_ERR_NO_SCENE, "rt_unit_trace: no scene uploaded");
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    TmpDev<uint32_t> dIjs, dTrav;
    TmpDev<float> dOut;
    RT_HIP(dIjs.Alloc((size_t)n * 3));
    RT_HIP(dTrav.Alloc(n));
    RT_HIP(dOut.Alloc((size_t)n * 3));
    RT_HIP(hipMemcpy(dIjs.p, ijs, (size_t)n * 3 * sizeof(uint32_t), hipMemcpyHostToDevice));
    RT_HIP(hipMemsetAsync(ctx->counters.ptr, 0, 2 * sizeof(unsigned long long), ctx->stream));
    rtd::TraceParams tp = ctx->base;
    tp.W = W;
    tp.H = H;
    tp.rs = rt_rowset{0, H, H, 0, 1};
    tp.s0 = 1;
    tp.spp_pass = 1;
    tp.total_paths = n;
    tp.npix_local = n;
    tp.max_depth = max_depth;
    tp.sampler = ctx->sampler;
    tp.seed = seed;
    tp.path_list = dIjs.p;
    tp.jitter_tab = nullptr;
    tp.lens_tab = nullptr;
    tp.samples = dOut.p;
    tp.trav_out = dTrav.p;
    tp.counters = ctx->counters.ptr;
    int rc = LaunchTrace(ctx, tp);
    if (rc != RT_OK) return rc;
    RT_HIP(hipStreamSynchronize(ctx->stream));
    RT_HIP(hipMemcpy(out_rgb, dOut.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (out_traversals) RT_HIP(hipMemcpy(out_traversals, dTrav.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_unit_camera_rays(rt_ctx* ctx, const rt_camera* camera, const float* uv_offset, uint32_t n, float* out_rays) {
    if (!ctx || !camera || !uv_offset || !out_rays) return Fail(RT_ERR_INVALID_ARG, "rt_unit_camera_rays: invalid argument");
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    TmpDev<float> dIn, dOut;
    RT_HIP(dIn.Alloc((size_t)n * 4));
    RT_HIP(dOut.Alloc((size_t)n * 6));
    RT_HIP(hipMemcpy(dIn.p, uv_offset, (size_t)n * 4 * sizeof(float), hipMemcpyHostToDevice));
    rtd::TraceParams tp{};
    for (int k = 0; k < 3; ++k) {
        tp.cam_o[k] = camera->origin[k];
        tp.cam_x[k] = camera->x[k];
        tp.cam_y[k] = camera->y[k];
        tp.cam_oip[k] = camera->origin_image_plane[k];
    }
    tp.aperture = camera->aperture;
    tp.focal = camera->focal_length;
    hipLaunchKernelGGL(rtd::k_unit_camera, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, tp, dIn.p, n, dOut.p);
    RT_HIP(hipGetLastError());
    RT_HIP(hipStreamSynchronize(ctx->stream));
    RT_HIP(hipMemcpy(out_rays, dOut.p, (size_t)n * 6 * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_unit_scatter(rt_ctx* ctx, const rt_material* material, const rt_light* sun, const float view_origin[3], const float* in,
                    uint32_t n, float* out) {
    if (!ctx || !material || !sun || !view_origin || !in || !out) return Fail(RT_ERR_INVALID_ARG, "rt_unit_scatter: invalid argument");
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    TmpDev<float> dIn, dOut;
    TmpDev<rt_material> dMat;
    RT_HIP(dIn.Alloc((size_t)n * 12));
    RT_HIP(dOut.Alloc((size_t)n * 11));
    RT_HIP(dMat.Alloc(1));
    RT_HIP(hipMemcpy(dIn.p, in, (size_t)n * 12 * sizeof(float), hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(dMat.p, material, sizeof(rt_material), hipMemcpyHostToDevice));
    rtd::TraceParams tp{};
    for (int k = 0; k < 3; ++k) {
        tp.cam_o[k] = view_origin[k];
        tp.sun_dir[k] = sun->direction[k];
        tp.sun_rad[k] = sun->luminance * sun->color[k];
    }
    tp.sampler = ctx->sampler;
    hipLaunchKernelGGL(rtd::k_unit_scatter, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, tp, dMat.p, dIn.p, n, dOut.p);
    RT_HIP(hipGetLastError());
    RT_HIP(hipStreamSynchronize(ctx->stream));
    RT_HIP(hipMemcpy(out, dOut.p, (size_t)n * 11 * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_unit_tonemap(rt_ctx* ctx, const float* hdr_rgb, uint32_t n, uint32_t n_samples, uint8_t* out_rgb) {
    if (!ctx || !hdr_rgb || !out_rgb || n_samples == 0) return Fail(RT_ERR_INVALID_ARG, "rt_unit_tonemap: invalid argument");
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    TmpDev<float> dIn;
    TmpDev<uint8_t> dOut;
    RT_HIP(dIn.Alloc((size_t)n * 3));
    RT_HIP(dOut.Alloc((size_t)n * 3));
    RT_HIP(hipMemcpy(dIn.p, hdr_rgb, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rtd::k_unit_tonemap, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, dIn.p, n, n_samples, dOut.p);
    RT_HIP(hipGetLastError());
    RT_HIP(hipStreamSynchronize(ctx->stream));
    RT_HIP(hipMemcpy(out_rgb, dOut.p, (size_t)n * 3, hipMemcpyDeviceToHost));
    return RT_OK;
}

// Host-only: the clustered layout rt_scene_upload builds (no device needed).  orig: 4 entries per group
// (0xffffffff = padding), bounds: Cx, Cy, Cz, |C|^2 - Rf^2 per group.  Pass cap_groups = 0 to query the count.
int rt_unit_layout(const rt_sphere* spheres, uint32_t n, uint32_t cap_groups, uint32_t* n_groups, uint32_t* orig, float* bounds) {
    if (!spheres || n == 0 || !n_groups) return Fail(RT_ERR_INVALID_ARG, "rt_unit_layout: invalid argument");
    SceneLayout L;
    BuildLayout(spheres, n, TreeTopFromEnv(), L);
    *n_groups = L.nGroups;
    if (cap_groups == 0) return RT_OK;
    if (cap_groups < L.nGroups || !orig || !bounds) return Fail(RT_ERR_INVALID_ARG, "rt_unit_layout: capacity too small");
    std::memcpy(orig, L.orig.data(), (size_t)L.nGroups * 4 * sizeof(uint32_t));
    std::memcpy(bounds, L.tree.data(), (size_t)L.nGroups * sizeof(float4));  // level 0 comes first
    return RT_OK;
}

// Host-only: the grid walk's row computation, the SAME functions the kernels call (rt_scan.h; RT_DEV is __host__ __device__).
int rt_unit_grid_rows(const float* segments, const int32_t* iu, uint32_t n, int32_t nv, int32_t* out_rows, float* out_s_enter) {
    if (!segments || !iu || !out_rows || nv <= 0) return Fail(RT_ERR_INVALID_ARG, "rt_unit_grid_rows: invalid argument");
    for (uint32_t k = 0; k < n; ++k) {
        const float* q = segments + 5 * (size_t)k;
        float slope, invAbsDu, sEnter;
        rtd::grid_segment_slope(q[0], q[1], q[2], q[3], slope, invAbsDu);
        int r0, r1;
        rtd::grid_slab_rows(q[0], q[1], q[2], q[3], q[4], slope, invAbsDu, (int)iu[k], (int)nv, r0, r1, sEnter);
        out_rows[2 * k] = r0;
        out_rows[2 * k + 1] = r1;
        if (out_s_enter) out_s_enter[k] = sEnter;
    }
    return RT_OK;
}

int rt_unit_layout_info(const rt_sphere* spheres, uint32_t n, uint32_t out[5]) {
    if (!spheres || n == 0 || !out) return Fail(RT_ERR_INVALID_ARG, "rt_unit_layout_info: invalid argument");
    SceneLayout L;
    BuildLayout(spheres, n, TreeTopFromEnv(), L);
    out[0] = L.gridOn ? 1u : (L.nLevels > 1 ? 2u : 0u);
    out[1] = L.gridOn ? L.gridNu : 0u;
    out[2] = L.gridOn ? L.gridNv : 0u;
    out[3] = L.nAlways;
    out[4] = L.nLevels;
    return RT_OK;
}

#ifdef RT_TIMELINE
// Diagnostic build only: wall-clock landmarks (100 MHz ticks) of the trace kernels since the last call; then reset.
int rt_debug_timeline(rt_ctx* ctx, unsigned long long out[16]) {
    if (!ctx || !out) return Fail(RT_ERR_INVALID_ARG, "rt_debug_timeline: invalid argument");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    RT_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(rtd::g_tl), 16 * sizeof(unsigned long long)));
    unsigned long long z[16] = {~0ull, 0, 0, 0, 0, 0, ~0ull, ~0ull, 0, 0, 0, 0, 0, 0, ~0ull, 0};
    RT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rtd::g_tl), z, sizeof(z)));
    return RT_OK;
}
int rt_debug_timeline_waves(rt_ctx* ctx, unsigned long long out[4096 * 8]) {
    if (!ctx || !out) return Fail(RT_ERR_INVALID_ARG, "rt_debug_timeline_waves: invalid argument");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    RT_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(rtd::g_tlWave), 4096 * 8 * sizeof(unsigned long long)));
    return RT_OK;
}
int rt_debug_timeline_ring(rt_ctx* ctx, unsigned long long out[4096 * 16]) {
    if (!ctx || !out) return Fail(RT_ERR_INVALID_ARG, "rt_debug_timeline_ring: invalid argument");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    RT_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(rtd::g_tlRing), 4096 * 16 * sizeof(unsigned long long)));
    return RT_OK;
}
int rt_debug_timeline_last(rt_ctx* ctx, unsigned int out[4096 * 4]) {
    if (!ctx || !out) return Fail(RT_ERR_INVALID_ARG, "rt_debug_timeline_last: invalid argument");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    RT_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(rtd::g_tlLast), 4096 * 4 * sizeof(unsigned int)));
    return RT_OK;
}
int rt_debug_timeline_hist(rt_ctx* ctx, unsigned int out[1024]) {
    if (!ctx || !out) return Fail(RT_ERR_INVALID_ARG, "rt_debug_timeline_hist: invalid argument");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    RT_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(rtd::g_tlHist), 1024 * sizeof(unsigned int)));
    static unsigned int zero[1024];
    RT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rtd::g_tlHist), zero, sizeof(zero)));
    return RT_OK;
}
#endif

#ifdef RT_STAMPS
// Diagnostic build only: read and clear the section clocks (see rt_params.h g_dbg).
int rt_debug_stamps(rt_ctx* ctx, unsigned long long out[20]) {
    if (!ctx || !out) return Fail(RT_ERR_INVALID_ARG, "rt_debug_stamps: invalid argument");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    RT_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(rtd::g_dbg), 20 * sizeof(unsigned long long)));
    unsigned long long z[20] = {0};
    RT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rtd::g_dbg), z, sizeof(z)));
    return RT_OK;
}
#endif

#ifdef RT_PHASES
// Diagnostic build only (tools/phase_budget.py): read and clear the region counters (rt_params.h g_sites); out[2 * site] = wave
// visits, out[2 * site + 1] = active lanes.  names: the sites' names, comma separated, in id order.
int rt_debug_sites(rt_ctx* ctx, unsigned long long* out, uint32_t cap, const char** names) {
    static const char* kNames =
#define RT_SITE_NAME(n) #n ","
        RT_SITE_LIST(RT_SITE_NAME)
#undef RT_SITE_NAME
        ;
    if (names) *names = kNames;
    if (!ctx || !out || cap < 2u * rtd::SITE_COUNT) return Fail(RT_ERR_INVALID_ARG, "rt_debug_sites: invalid argument");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    static unsigned long long raw[rtd::SITE_COUNT * 16];
    RT_HIP(hipMemcpyFromSymbol(raw, HIP_SYMBOL(rtd::g_sites), sizeof(raw)));
    for (uint32_t k = 0; k < rtd::SITE_COUNT; ++k) {
        out[2 * k] = raw[16 * k];
        out[2 * k + 1] = raw[16 * k + 1];
    }
    std::memset(raw, 0, sizeof(raw));
    RT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rtd::g_sites), raw, sizeof(raw)));
    return RT_OK;
}
#endif

}  // extern "C"

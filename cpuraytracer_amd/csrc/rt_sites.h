// rt_sites.h -- RT_SITE(NAME): markers of the source regions tools/phase_budget.py attributes instructions and trip counts to.
#pragma once
#include <stdint.h>

namespace rtd {
// Diagnostic build only (-DRT_PHASES; tools/phase_budget.py): how often each marked source region runs, per wave (visits) and
// per lane (active lanes summed over the visits).  RT_SITE(NAME) is the first statement of the region's { } block; in the shipped
// library it expands to nothing.  The tool multiplies these trip counts with the static instruction counts of the SHIPPED kernel
// (every instruction attributed, through its inline stack, to the innermost marked region its source lines lie in).
#define RT_SITE_LIST(X) \
    X(K_WAVE) X(K_ITER) X(K_PATHLIST) X(K_PARTIAL) X(P_HALTON) X(M_DIV_SLOW) X(M_SQRT_SLOW) X(M_ROOT_SLOW) X(K_POP) X(K_POP_LANE) X(K_PUSH) \
    X(K_PUSH_LANE) X(K_PROCESS) X(K_GEN) X(K_GEN_LANE) X(K_NEXTBLOCK) X(K_CLAIM) X(K_TRANS_MISS) X(K_TRANS_HIT) \
    X(K_TRANS_SHADOW) X(K_FINISH) X(H_PROCESS) X(H_MAT16) X(H_MULTI) X(H_SCATTER) X(H_SHADEV) X(H_TRANSPARENT) X(H_METAL) X(H_OPAQUE) \
    X(H_OPAQUE_DIFFUSE) X(H_SHADOWQ) X(H_SQ_WALK) X(H_SQ_CONSIDER) X(H_SQ_GROUND) X(H_SQ_GTAIL) X(H_SQ_ROUND) X(H_SQ_TAIL1) X(H_SQ_CELL) \
    X(H_SQ_ROOTS) X(H_SQ_FULL) X(H_SHADE) X(H_INDEXED) X(H_FARHIT) X(S_SCAN) X(S_TILEPAIR) X(S_SINGLE) \
    X(S_SINGLE_PUSH0) X(S_SINGLE_PUSH1) X(S_PASS) X(S_TAKE) X(S_PUSH_WORD) X(S_TAKE_PUSH0) X(S_TAKE_PUSH1) X(S_ASTEP) X(S_ASTEP2) X(S_APUSH) X(S_DRAIN) X(S_BSTEP) X(S_BSTEP2) X(S_BMIN) \
    X(G_SCAN) X(G_DRAIN) X(G_BSTEP) X(G_BMIN) X(G_BIG) X(G_BIGPUSH) X(G_FEED) X(G_FEED_LANE) X(G_ROUND) X(G_STEP) X(G_PUSH)
enum RtSite : uint32_t {
#define RT_SITE_ENUM(n) SITE_##n,
    RT_SITE_LIST(RT_SITE_ENUM)
#undef RT_SITE_ENUM
    SITE_COUNT
};
#if defined(RT_PHASES) && defined(__HIPCC__)
static __device__ unsigned long long g_sites[SITE_COUNT * 16];  // [site][0] = wave visits, [site][1] = active lanes; 128 bytes apart
__device__ __forceinline__ void rt_site(uint32_t id) {
    const unsigned long long m = __ballot(1);
    if ((threadIdx.x & 63u) == (uint32_t)__ffsll(m) - 1u) {
        atomicAdd(&g_sites[id * 16u], 1ull);
        atomicAdd(&g_sites[id * 16u + 1u], (unsigned long long)__popcll(m));
    }
}
#define RT_SITE(n) rtd::rt_site(rtd::SITE_##n)
#else
#define RT_SITE(n)
#endif

}  // namespace rtd

// <optix.h> stand-in for the device side of render/optixutils/c_src/envsampling/kernel.cu (TEST INFRASTRUCTURE).
//
// Only what the raygen / miss programs touch: the launch index, the payload register and optixTrace.  The any-hit query
// (kernel.cu:101-117: tmin 0, tmax 1e16, TERMINATE_ON_FIRST_HIT, closest-hit disabled, the miss program sets payload 0 to 1)
// is answered with the Moeller-Trumbore predicate in fp32, compiled without contraction (oracle/anyhit_grid.h: ah_tri_hit) -- the
// SAME predicate the HIP traversal applies at its leaves (gshell_amd/csrc/bvh.hpp: tri_hit), so that a visibility difference
// between this build and the product is a traversal bug, not a predicate choice -- over EVERY triangle (use_grid = 0: the
// definition) or over the candidates of a conservative uniform grid (use_grid = 1: what makes a 512 x 512 launch against a
// 2 x 10^5-triangle mesh take seconds; asserted identical to the definition by tests/test_oracle_anyhit_cpu.py and by the golden
// re-mint in both modes).  OptiX's own hardware predicate is not specified anywhere; rays that graze a triangle edge may differ
// from a real OptiX run.
#pragma once
#include "cuda_host_shim.h"
#include "../anyhit_grid.h"   // the predicate (ah_tri_hit) and its two evaluations: every triangle / grid-filtered candidates

typedef unsigned long long OptixTraversableHandle;
typedef unsigned int OptixVisibilityMask;
enum {
    OPTIX_RAY_FLAG_NONE = 0,
    OPTIX_RAY_FLAG_DISABLE_ANYHIT = 1u << 0,
    OPTIX_RAY_FLAG_TERMINATE_ON_FIRST_HIT = 1u << 2,
    OPTIX_RAY_FLAG_DISABLE_CLOSESTHIT = 1u << 3,
};

namespace optixhost {
struct Scene {                // occluder mesh of the current launch: v0, e1, e2 per triangle (fp32)
    const float* v0e1e2 = nullptr;
    long long T = 0;
    AhGrid grid = {};         // candidate filter over the same records (oracle/anyhit_grid.h); grid.built == 0: brute force
    int use_grid = 1;         // 0: every triangle is tested (the definition; what the goldens were first minted with)
};
inline Scene g_scene;
inline thread_local uint3 g_idx = {0, 0, 0};
inline uint3 g_dim = {0, 0, 0};
inline thread_local unsigned int g_payload0 = 0;
}  // namespace optixhost

extern "C" void __miss__ms();   // the reference's miss program (kernel.cu:543-546)

static inline uint3 optixGetLaunchIndex() { return optixhost::g_idx; }
static inline uint3 optixGetLaunchDimensions() { return optixhost::g_dim; }
static inline void optixSetPayload_0(unsigned int v) { optixhost::g_payload0 = v; }

static inline void optixTrace(OptixTraversableHandle, float3 o, float3 d, float tmin, float tmax, float /*time*/, OptixVisibilityMask,
                              unsigned int /*flags*/, unsigned int, unsigned int, unsigned int, unsigned int& p0) {
    (void)tmin; (void)tmax;     // 0 and 1e16 at the only call site; tri_hit has them built in
    optixhost::g_payload0 = p0;
    const optixhost::Scene& s = optixhost::g_scene;
    const bool hit = (s.use_grid && s.grid.built) ? ah_grid_query(&s.grid, o.x, o.y, o.z, d.x, d.y, d.z, nullptr) != 0
                                                  : ah_brute(s.v0e1e2, s.T, o.x, o.y, o.z, d.x, d.y, d.z) != 0;
    if (!hit) __miss__ms();
    p0 = optixhost::g_payload0;
}

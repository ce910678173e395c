// <optix.h> stand-in for the device side of render/optixutils/c_src/envsampling/kernel.cu (TEST INFRASTRUCTURE).
//
// Only what the raygen / miss programs touch: the launch index, the payload register and optixTrace.  The any-hit query
// (kernel.cu:101-117: tmin 0, tmax 1e16, TERMINATE_ON_FIRST_HIT, closest-hit disabled, the miss program sets payload 0 to 1)
// is answered by brute force over the occluder triangles with the Moeller-Trumbore predicate in fp32, compiled without
// contraction -- the SAME predicate the HIP traversal applies at its leaves (gshell_amd/csrc/bvh.hpp: tri_hit), so that a
// visibility difference between this build and the product is a traversal bug, not a predicate choice.  OptiX's own
// hardware predicate is not specified anywhere; rays that graze a triangle edge may differ from a real OptiX run.
#pragma once
#include "cuda_host_shim.h"

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
    long long n_rays = 0, n_tests = 0;
};
inline Scene g_scene;
inline thread_local uint3 g_idx = {0, 0, 0};
inline uint3 g_dim = {0, 0, 0};
inline thread_local unsigned int g_payload0 = 0;

inline bool tri_hit(const float* r, float ox, float oy, float oz, float dx, float dy, float dz) {
    const float v0x = r[0], v0y = r[1], v0z = r[2], e1x = r[3], e1y = r[4], e1z = r[5], e2x = r[6], e2y = r[7], e2z = r[8];
    float px = dy * e2z - dz * e2y, py = dz * e2x - dx * e2z, pz = dx * e2y - dy * e2x;
    float det = e1x * px + e1y * py + e1z * pz;
    if (!(fabsf(det) > 1e-20f)) return false;
    float inv = 1.0f / det;
    float tx = ox - v0x, ty = oy - v0y, tz = oz - v0z;
    float u = (tx * px + ty * py + tz * pz) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    float qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
    float v = (dx * qx + dy * qy + dz * qz) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    float t = (e2x * qx + e2y * qy + e2z * qz) * inv;
    return t > 0.0f && t < 1e16f;
}
}  // namespace optixhost

extern "C" void __miss__ms();   // the reference's miss program (kernel.cu:543-546)

static inline uint3 optixGetLaunchIndex() { return optixhost::g_idx; }
static inline uint3 optixGetLaunchDimensions() { return optixhost::g_dim; }
static inline void optixSetPayload_0(unsigned int v) { optixhost::g_payload0 = v; }

static inline void optixTrace(OptixTraversableHandle, float3 o, float3 d, float tmin, float tmax, float /*time*/, OptixVisibilityMask,
                              unsigned int /*flags*/, unsigned int, unsigned int, unsigned int, unsigned int& p0) {
    (void)tmin; (void)tmax;     // 0 and 1e16 at the only call site; tri_hit has them built in
    optixhost::g_payload0 = p0;
    bool hit = false;
    const bool valid = (d.x == d.x && d.y == d.y && d.z == d.z) && !(d.x == 0.f && d.y == 0.f && d.z == 0.f);
    if (valid) {
        const optixhost::Scene& s = optixhost::g_scene;
        for (long long t = 0; t < s.T && !hit; ++t) hit = optixhost::tri_hit(s.v0e1e2 + 9 * t, o.x, o.y, o.z, d.x, d.y, d.z);
    }
    if (!hit) __miss__ms();
    p0 = optixhost::g_payload0;
}

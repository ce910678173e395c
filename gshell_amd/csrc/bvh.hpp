// Implicit 4-ary BVH over Morton-sorted triangles (device side: any-hit shadow-ray traversal).
//
// Replaces the OptiX geometry-acceleration structure the reference rebuilds every iteration
// (render/optixutils/c_src/torch_bindings.cpp:37-116, called from geometry/gshell_tets_geometry.py:211)
// and the hardware any-hit query `optixTrace(... TERMINATE_ON_FIRST_HIT ...)`
// (render/optixutils/c_src/envsampling/kernel.cu:101-117).  CDNA4 has no ray-tracing units, so:
//   * build  = one 30-bit Morton radix sort of the centroids + a pointer-free complete 4-ary heap:
//              leaf i owns the sorted triangles [i*leaf, (i+1)*leaf), node n has children 4n+1..4n+4,
//              so "refit" is D tiny launches of min/max over 4 slots -- no atomics, no fences, no
//              parent pointers; the whole rebuild is a handful of launches per iteration.
//   * layout = per internal node ONE 96-byte record holding its 4 child boxes in SoA form
//              (6 x float4: lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4]) -> 6 dwordx4 loads per visit;
//              triangles are stored pre-gathered in sorted order as (v0, e1, e2) = 3 x float4.
//   * traverse = per-lane short stack in LDS (stack[entry][lane], conflict-free), children tested
//              4 at a time, leaves intersected immediately (Moeller-Trumbore, t > 0).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

struct gs_bvh {
    int64_t T = 0;           // triangles in the current build
    int depth = 0;           // leaves live at heap level `depth` (>= 1)
    int leaf = 1;            // triangles per leaf (1..GS_BVH_LEAF)
    int64_t n_internal = 0;  // (4^depth - 1) / 3
    int64_t n_leaf = 0;      // ceil(T / leaf)
    float4* groups = nullptr;   // [n_internal * 6]
    float4* tris = nullptr;     // [T * 3]  v0, e1, e2 in Morton order
    int32_t* tri_id = nullptr;  // [T] original triangle id of each sorted slot
    // build scratch
    uint32_t *keys = nullptr, *keys2 = nullptr, *vals = nullptr, *vals2 = nullptr;
    uint32_t* bounds = nullptr;  // [6] ordered-uint encoded scene bounds of the centroids
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    int64_t cap_T = 0, cap_internal = 0;
};

struct BvhView {  // passed by value to kernels
    const float4* groups;
    const float4* tris;
    int64_t T, n_internal, n_leaf;
    int leaf;
};

static inline BvhView bvh_view(const gs_bvh* b) { return {b->groups, b->tris, b->T, b->n_internal, b->n_leaf, b->leaf}; }

constexpr int BVH_STACK = 32;  // entries per lane: a 4-ary heap pushes <= 3 siblings per level, depth <= 10 for T <= 4M

__device__ __forceinline__ bool tri_hit(const float4* __restrict__ tp, float ox, float oy, float oz, float dx, float dy, float dz) {
    float4 v0 = tp[0], e1 = tp[1], e2 = tp[2];
    float px = dy * e2.z - dz * e2.y, py = dz * e2.x - dx * e2.z, pz = dx * e2.y - dy * e2.x;
    float det = e1.x * px + e1.y * py + e1.z * pz;
    if (!(fabsf(det) > 1e-20f)) return false;
    float inv = 1.0f / det;
    float tx = ox - v0.x, ty = oy - v0.y, tz = oz - v0.z;
    float u = (tx * px + ty * py + tz * pz) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    float qx = ty * e1.z - tz * e1.y, qy = tz * e1.x - tx * e1.z, qz = tx * e1.y - ty * e1.x;
    float v = (dx * qx + dy * qy + dz * qz) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    float t = (e2.x * qx + e2.y * qy + e2.z * qz) * inv;
    return t > 0.0f && t < 1e16f;
}

// true if the ray (o, d), t in (0, 1e16), hits any triangle.  `stack` = this block's LDS stack base,
// entry e of lane `tid` lives at stack[e * nthreads + tid].
template <bool STATS = false>
__device__ __forceinline__ bool bvh_any_hit(const BvhView& bv, float ox, float oy, float oz, float dx, float dy, float dz, int32_t* stack,
                                            int tid, int nthreads, int* n_nodes = nullptr, int* n_tris = nullptr) {
    if (bv.T <= 0) return false;
    if (!(dx == dx && dy == dy && dz == dz) || (dx == 0.f && dy == 0.f && dz == 0.f)) return false;
    float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    int sp = 0;
    stack[tid] = 0;
    sp = 1;
    while (sp > 0) {
        --sp;
        int32_t n = stack[sp * nthreads + tid];
        if (STATS) ++*n_nodes;
        const float4* g = bv.groups + (int64_t)n * 6;
        float4 lox = g[0], loy = g[1], loz = g[2], hix = g[3], hiy = g[4], hiz = g[5];
        const float* plx = &lox.x; const float* ply = &loy.x; const float* plz = &loz.x;
        const float* phx = &hix.x; const float* phy = &hiy.x; const float* phz = &hiz.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float lx = plx[k], hx = phx[k];
            if (!(lx <= hx)) continue;  // empty slot
            float t0 = (lx - ox) * ix, t1 = (hx - ox) * ix;
            float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
            t0 = (ply[k] - oy) * iy;
            t1 = (phy[k] - oy) * iy;
            tn = fmaxf(tn, fminf(t0, t1));
            tf = fminf(tf, fmaxf(t0, t1));
            t0 = (plz[k] - oz) * iz;
            t1 = (phz[k] - oz) * iz;
            tn = fmaxf(tn, fminf(t0, t1));
            tf = fminf(tf, fmaxf(t0, t1));
            if (!(tf >= fmaxf(tn, 0.0f))) continue;
            int64_t c = 4 * (int64_t)n + 1 + k;
            if (c >= bv.n_internal) {
                int64_t li = c - bv.n_internal;
                int64_t t_begin = li * bv.leaf, t_end = min(t_begin + bv.leaf, bv.T);
                for (int64_t t = t_begin; t < t_end; ++t) {
                    if (STATS) ++*n_tris;
                    if (tri_hit(bv.tris + 3 * t, ox, oy, oz, dx, dy, dz)) return true;
                }
            } else if (sp < BVH_STACK) {
                stack[sp * nthreads + tid] = (int32_t)c;
                ++sp;
            }
        }
    }
    return false;
}

// BVH build (see bvh.hpp for the design).  Replaces ou.optix_build_bvh
// (render/optixutils/ops.py:133-139 -> c_src/torch_bindings.cpp:37-116), rebuilt every iteration.
#include "bvh.hpp"

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cmath>

#include "../../include/gshell_hip.h"
#include "common.hpp"

#ifndef GS_BVH_HILBERT
#define GS_BVH_HILBERT 1
#endif
// one triangle per leaf (round 1 measured leaf <= 4 -> 25 triangle tests / ray, leaf <= 2 -> 2.8, against 20 % more node visits)

namespace {

__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__global__ void k_bounds_init(uint32_t* b) {
    if (threadIdx.x < 3) b[threadIdx.x] = 0xffffffffu;
    else if (threadIdx.x < 6) b[threadIdx.x] = 0u;
}

__device__ __forceinline__ bool load_tri(const float* __restrict__ v, const int32_t* __restrict__ tri, int64_t t, int64_t V, float3& a, float3& b,
                                         float3& c) {
    int32_t i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
    if ((uint32_t)i0 >= (uint32_t)V || (uint32_t)i1 >= (uint32_t)V || (uint32_t)i2 >= (uint32_t)V) return false;
    a = make_float3(v[3 * (int64_t)i0], v[3 * (int64_t)i0 + 1], v[3 * (int64_t)i0 + 2]);
    b = make_float3(v[3 * (int64_t)i1], v[3 * (int64_t)i1 + 1], v[3 * (int64_t)i1 + 2]);
    c = make_float3(v[3 * (int64_t)i2], v[3 * (int64_t)i2 + 1], v[3 * (int64_t)i2 + 2]);
    bool fin = isfinite(a.x + a.y + a.z + b.x + b.y + b.z + c.x + c.y + c.z);
    return fin;
}

// Grid-stride over the triangles, one atomic per bound and WORKGROUP (the first version issued six atomics per wave onto the
// same six words: 21 k serialised atomics = 0.25 ms for a 0.23 M-triangle mesh).
__global__ void __launch_bounds__(256) k_centroid_bounds(const float* __restrict__ v, const int32_t* __restrict__ tri, int64_t T, int64_t V,
                                                         uint32_t* __restrict__ bounds) {
    __shared__ float s_lo[4][3], s_hi[4][3];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (int64_t)gridDim.x * blockDim.x) {
        float3 a, b, c;
        if (load_tri(v, tri, t, V, a, b, c)) {
            const float cx = (a.x + b.x + c.x) * (1.0f / 3.0f), cy = (a.y + b.y + c.y) * (1.0f / 3.0f), cz = (a.z + b.z + c.z) * (1.0f / 3.0f);
            lo[0] = fminf(lo[0], cx); hi[0] = fmaxf(hi[0], cx);
            lo[1] = fminf(lo[1], cy); hi[1] = fmaxf(hi[1], cy);
            lo[2] = fminf(lo[2], cz); hi[2] = fmaxf(hi[2], cz);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], o, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o, 64));
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int k = 0; k < 3; ++k) {
            s_lo[wave][k] = lo[k];
            s_hi[wave][k] = hi[k];
        }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        const float l = fminf(fminf(s_lo[0][k], s_lo[1][k]), fminf(s_lo[2][k], s_lo[3][k]));
        const float h = fmaxf(fmaxf(s_hi[0][k], s_hi[1][k]), fmaxf(s_hi[2][k], s_hi[3][k]));
        if (l <= h) {
            atomicMin(&bounds[k], f2ord(l));
            atomicMax(&bounds[3 + k], f2ord(h));
        }
    }
}

__device__ __forceinline__ uint32_t expand10(uint32_t x) {
    x &= 0x3ffu;
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ void __launch_bounds__(256) k_morton(const float* __restrict__ v, const int32_t* __restrict__ tri, int64_t T, int64_t V,
                                                const uint32_t* __restrict__ bounds, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    float3 a, b, c;
    uint32_t key = 0x3fffffffu;  // degenerate / invalid triangles sort last
    if (load_tri(v, tri, t, V, a, b, c)) {
        float cen[3] = {(a.x + b.x + c.x) * (1.0f / 3.0f), (a.y + b.y + c.y) * (1.0f / 3.0f), (a.z + b.z + c.z) * (1.0f / 3.0f)};
        uint32_t q[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float lo = ord2f(bounds[k]), hi = ord2f(bounds[3 + k]);
            float ext = fmaxf(hi - lo, 1e-30f);
            q[k] = (uint32_t)fminf(fmaxf((cen[k] - lo) / ext * 1024.0f, 0.0f), 1023.0f);
        }
#if GS_BVH_HILBERT
        // Hilbert index instead of the Morton code (Skilling 2004, "Programming the Hilbert curve", axes -> transpose): consecutive
        // keys are always spatially adjacent, so the equal-count ranges of the implicit heap are compact patches of the surface;
        // the Z curve's jumps gave ranges with large, overlapping boxes (measured: node visits per shadow ray in profiles/r03_bvh_stats.json)
        {
            const uint32_t M = 1u << 9;
            for (uint32_t Q = M; Q > 1u; Q >>= 1) {
                const uint32_t P = Q - 1u;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (q[i] & Q) {
                        q[0] ^= P;
                    } else {
                        const uint32_t t = (q[0] ^ q[i]) & P;
                        q[0] ^= t;
                        q[i] ^= t;
                    }
                }
            }
            q[1] ^= q[0];
            q[2] ^= q[1];
            uint32_t t = 0u;
            for (uint32_t Q = M; Q > 1u; Q >>= 1)
                if (q[2] & Q) t ^= Q - 1u;
            q[0] ^= t; q[1] ^= t; q[2] ^= t;
        }
#endif
        key = (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
    }
    keys[t] = key;
    vals[t] = (uint32_t)t;
}

// gather sorted triangles as (v0, e1, e2) and write each leaf's box into its parent's group record
__global__ void __launch_bounds__(256) k_leaves(const float* __restrict__ v, const int32_t* __restrict__ tri, int64_t T, int64_t V,
                                                const uint32_t* __restrict__ sorted_ids, int leaf, int64_t n_leaf_slots, int64_t n_internal,
                                                float4* __restrict__ tris_out, int32_t* __restrict__ tri_id, float* __restrict__ groups) {
    int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n_leaf_slots) return;
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    int64_t t0 = li * leaf, t1 = min(t0 + leaf, T);
    for (int64_t s = t0; s < t1; ++s) {
        uint32_t t = sorted_ids[s];
        tri_id[s] = (int32_t)t;
        float3 a, b, c;
        if (!load_tri(v, tri, t, V, a, b, c)) {  // never hit: NaN edges fail every comparison
            float nanv = __uint_as_float(0x7fc00000u);
            tris_out[3 * s] = tris_out[3 * s + 1] = tris_out[3 * s + 2] = make_float4(nanv, nanv, nanv, 0.f);
            continue;
        }
        tris_out[3 * s] = make_float4(a.x, a.y, a.z, 0.f);
        tris_out[3 * s + 1] = make_float4(b.x - a.x, b.y - a.y, b.z - a.z, 0.f);
        tris_out[3 * s + 2] = make_float4(c.x - a.x, c.y - a.y, c.z - a.z, 0.f);
        lo[0] = fminf(lo[0], fminf(a.x, fminf(b.x, c.x)));
        lo[1] = fminf(lo[1], fminf(a.y, fminf(b.y, c.y)));
        lo[2] = fminf(lo[2], fminf(a.z, fminf(b.z, c.z)));
        hi[0] = fmaxf(hi[0], fmaxf(a.x, fmaxf(b.x, c.x)));
        hi[1] = fmaxf(hi[1], fmaxf(a.y, fmaxf(b.y, c.y)));
        hi[2] = fmaxf(hi[2], fmaxf(a.z, fmaxf(b.z, c.z)));
    }
    if (lo[0] <= hi[0]) {  // conservative padding: the slab test must never cull a triangle the exact test accepts
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float pad = 1e-5f * fmaxf(fmaxf(fabsf(lo[k]), fabsf(hi[k])), 1.0f);
            lo[k] -= pad;
            hi[k] += pad;
        }
    }
    int64_t heap = n_internal + li, parent = (heap - 1) >> 3;
    int slot = (int)((heap - 1) & 7);
    float* g = groups + parent * 48;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        g[8 * k + slot] = lo[k];
        g[24 + 8 * k + slot] = hi[k];
    }
}

// box of every node at heap level `lvl` (first heap index `first`, `count` nodes) = union of its 8 child slots
__global__ void __launch_bounds__(256) k_level_up(int64_t first, int64_t count, float* __restrict__ groups) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    int64_t n = first + i;
    const float* g = groups + n * 48;
    float lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = g[8 * k];
        hi[k] = g[24 + 8 * k];
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            lo[k] = fminf(lo[k], g[8 * k + j]);
            hi[k] = fmaxf(hi[k], g[24 + 8 * k + j]);
        }
    }
    int64_t parent = (n - 1) >> 3;
    int slot = (int)((n - 1) & 7);
    float* pg = groups + parent * 48;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        pg[8 * k + slot] = lo[k];
        pg[24 + 8 * k + slot] = hi[k];
    }
}

// largest half <= x / smallest half >= x, as bit patterns (boxes may only grow when they are narrowed to 16 bits)
__device__ __forceinline__ uint16_t half_floor(float x) {
    union { _Float16 h; uint16_t b; } c;
    c.h = (_Float16)x;
    if ((float)c.h > x) c.b = (c.b & 0x7fff) == 0 ? 0x8001 : ((c.b & 0x8000) ? c.b + 1 : c.b - 1);
    return c.b;
}
__device__ __forceinline__ uint16_t half_ceil(float x) {
    union { _Float16 h; uint16_t b; } c;
    c.h = (_Float16)x;
    if ((float)c.h < x) c.b = (c.b & 0x7fff) == 0 ? 0x0001 : ((c.b & 0x8000) ? c.b - 1 : c.b + 1);
    return c.b;
}

// fp32 build records (48 floats) -> traversal records (48 halves = 6 groups of 16 bytes), one thread per (node, group)
__global__ void __launch_bounds__(256) k_pack_nodes(int64_t n_internal, const float* __restrict__ groups, uint4* __restrict__ nodes) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_internal * 6) return;
    const int grp = (int)(i % 6);                       // 0..2 lo (round down), 3..5 hi (round up)
    const float* g = groups + (i / 6) * 48 + grp * 8;
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint16_t a = grp < 3 ? half_floor(g[2 * q]) : half_ceil(g[2 * q]);
        const uint16_t b = grp < 3 ? half_floor(g[2 * q + 1]) : half_ceil(g[2 * q + 1]);
        w[q] = (uint32_t)a | ((uint32_t)b << 16);
    }
    nodes[i] = make_uint4(w[0], w[1], w[2], w[3]);
}

}  // namespace

extern "C" int gs_bvh_create(gs_bvh** out) {
    GS_REQUIRE(out != nullptr, "gs_bvh_create: out is null");
    gs_bvh* b = new gs_bvh();      // no device memory until the first build (the handle can be created before a device is chosen)
    *out = b;
    return 0;
}

extern "C" int gs_bvh_destroy(gs_bvh* b) {
    if (!b) return 0;
    for (void* p : {(void*)b->groups, (void*)b->nodes, (void*)b->tris, (void*)b->tri_id, (void*)b->keys, (void*)b->keys2, (void*)b->vals, (void*)b->vals2,
                    (void*)b->bounds, b->sort_tmp})
        (void)hipFree(p);
    delete b;
    return 0;
}

extern "C" int gs_bvh_info(const gs_bvh* b, int64_t* T, int64_t* depth, int64_t* leaf_size, int64_t* bytes) {
    GS_REQUIRE(b != nullptr, "gs_bvh_info: bvh is null");
    if (T) *T = b->T;
    if (depth) *depth = b->depth;
    if (leaf_size) *leaf_size = b->leaf;
    if (bytes) *bytes = b->n_internal * 96 + b->T * 48;   // what traversal reads
    return 0;
}

extern "C" int gs_bvh_build(gs_bvh* b, const float* verts, int64_t V, const int32_t* tris, int64_t T, gs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GS_REQUIRE(b != nullptr, "gs_bvh_build: bvh is null");
    GS_REQUIRE(T >= 0 && T < (1ll << 30), "gs_bvh_build: too many triangles");
    b->T = T;
    if (T == 0) return 0;  // empty meshes are legal (reference: ops.py:134-139)
    if (!b->bounds) GS_HIP_CHECK(hipMalloc(&b->bounds, 6 * sizeof(uint32_t)));
    GS_REQUIRE(verts && tris && V > 0, "gs_bvh_build: null mesh pointer");
    // shape of the implicit heap: 8^depth leaf slots >= T, one triangle per leaf, depth >= 1
    int depth = 1;
    while ((1ll << (3 * depth)) < T) ++depth;
    int64_t slots = 1ll << (3 * depth);
    GS_REQUIRE(depth <= BVH_STACK, "gs_bvh_build: mesh too large for the traversal stack");
    b->depth = depth;
    b->leaf = 1;
    b->n_leaf = T;
    b->n_internal = (slots - 1) / 7;
    GS_REQUIRE(b->n_internal < (1ll << 24), "gs_bvh_build: too many nodes for the 24-bit node index of a stack entry");
    if (T > b->cap_T) {
        GS_HIP_CHECK(hipStreamSynchronize(stream));
        for (void* p : {(void*)b->tris, (void*)b->tri_id, (void*)b->keys, (void*)b->keys2, (void*)b->vals, (void*)b->vals2, b->sort_tmp}) (void)hipFree(p);
        int64_t cap = T + T / 4 + 1024;
        GS_HIP_CHECK(hipMalloc(&b->tris, (size_t)cap * 48));
        GS_HIP_CHECK(hipMalloc(&b->tri_id, (size_t)cap * 4));
        GS_HIP_CHECK(hipMalloc(&b->keys, (size_t)cap * 4));
        GS_HIP_CHECK(hipMalloc(&b->keys2, (size_t)cap * 4));
        GS_HIP_CHECK(hipMalloc(&b->vals, (size_t)cap * 4));
        GS_HIP_CHECK(hipMalloc(&b->vals2, (size_t)cap * 4));
        size_t tmp = 0;
        GS_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp, b->keys, b->keys2, b->vals, b->vals2, (size_t)cap, 0, 30, stream));
        b->sort_tmp_bytes = tmp + 1024;
        GS_HIP_CHECK(hipMalloc(&b->sort_tmp, b->sort_tmp_bytes));
        b->cap_T = cap;
    }
    if (b->n_internal > b->cap_internal) {
        GS_HIP_CHECK(hipStreamSynchronize(stream));
        (void)hipFree(b->groups);
        (void)hipFree(b->nodes);
        GS_HIP_CHECK(hipMalloc(&b->groups, (size_t)b->n_internal * 192));
        GS_HIP_CHECK(hipMalloc(&b->nodes, (size_t)b->n_internal * 96));
        b->cap_internal = b->n_internal;
    }
    hipLaunchKernelGGL(k_bounds_init, dim3(1), dim3(64), 0, stream, b->bounds);
    hipLaunchKernelGGL(k_centroid_bounds, dim3((unsigned)std::min<int64_t>(gs::cdiv(T, 256), 256)), dim3(256), 0, stream, verts, tris, T, V, b->bounds);
    hipLaunchKernelGGL(k_morton, dim3((unsigned)gs::cdiv(T, 256)), dim3(256), 0, stream, verts, tris, T, V, b->bounds, b->keys, b->vals);
    size_t tmp = b->sort_tmp_bytes;
    GS_HIP_CHECK(rocprim::radix_sort_pairs(b->sort_tmp, tmp, b->keys, b->keys2, b->vals, b->vals2, (size_t)T, 0, 30, stream));
    hipLaunchKernelGGL(k_leaves, dim3((unsigned)gs::cdiv(slots, 256)), dim3(256), 0, stream, verts, tris, T, V, b->vals2, 1, slots, b->n_internal,
                       b->tris, b->tri_id, (float*)b->groups);
    for (int lvl = depth - 1; lvl >= 1; --lvl) {
        int64_t count = 1ll << (3 * lvl), first = (count - 1) / 7;
        hipLaunchKernelGGL(k_level_up, dim3((unsigned)gs::cdiv(count, 256)), dim3(256), 0, stream, first, count, (float*)b->groups);
    }
    hipLaunchKernelGGL(k_pack_nodes, dim3((unsigned)gs::cdiv(b->n_internal * 6, 256)), dim3(256), 0, stream, b->n_internal, (const float*)b->groups,
                       b->nodes);
    GS_LAUNCH_CHECK();
    return 0;
}

// ---- stand-alone any-hit query (tests, tools): hit[i] = 1 if ray i is occluded --------------------
namespace {
__global__ void __launch_bounds__(256) k_any_hit(BvhView bv, const float* __restrict__ org, const float* __restrict__ dir, int64_t n,
                                                 uint8_t* __restrict__ hit) {
    __shared__ int32_t stack[BVH_STACK * 256];
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    hit[i] = bvh_any_hit(bv, org[3 * i], org[3 * i + 1], org[3 * i + 2], dir[3 * i], dir[3 * i + 1], dir[3 * i + 2], stack, threadIdx.x, 256) ? 1 : 0;
}

__global__ void __launch_bounds__(256) k_any_hit_stats(BvhView bv, const float* __restrict__ org, const float* __restrict__ dir, int64_t n,
                                                       uint8_t* __restrict__ hit, int32_t* __restrict__ stats) {
    __shared__ int32_t stack[BVH_STACK * 256];
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int nn = 0, nt = 0;
    hit[i] = bvh_any_hit<true>(bv, org[3 * i], org[3 * i + 1], org[3 * i + 2], dir[3 * i], dir[3 * i + 1], dir[3 * i + 2], stack, threadIdx.x, 256, &nn,
                               &nt) ? 1 : 0;
    stats[2 * i] = nn;
    stats[2 * i + 1] = nt;
}
}  // namespace

extern "C" int gs_bvh_any_hit_stats(const gs_bvh* b, const float* origins, const float* dirs, int64_t n, uint8_t* hit, int32_t* stats,
                                    gs_stream_t stream) {
    GS_REQUIRE(b != nullptr, "gs_bvh_any_hit_stats: bvh is null");
    if (n == 0) return 0;
    GS_REQUIRE(origins && dirs && hit && stats, "gs_bvh_any_hit_stats: null pointer");
    hipLaunchKernelGGL(k_any_hit_stats, dim3((unsigned)gs::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, bvh_view(b), origins, dirs, n, hit, stats);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_bvh_any_hit(const gs_bvh* b, const float* origins, const float* dirs, int64_t n, uint8_t* hit, gs_stream_t stream) {
    GS_REQUIRE(b != nullptr, "gs_bvh_any_hit: bvh is null");
    if (n == 0) return 0;
    GS_REQUIRE(origins && dirs && hit, "gs_bvh_any_hit: null pointer");
    hipLaunchKernelGGL(k_any_hit, dim3((unsigned)gs::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, bvh_view(b), origins, dirs, n, hit);
    GS_LAUNCH_CHECK();
    return 0;
}

// ---- compile-time variants of this file (common.hpp): non-default values announce themselves through gs_build_flags(); switches that give
// wrong results (timing-only ablations) compile only under -DGS_EXPERIMENT
GS_TUNABLE(GS_BVH_HILBERT, 1)

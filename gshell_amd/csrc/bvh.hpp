// Implicit 8-ary BVH over Morton-sorted triangles (device side: any-hit shadow-ray traversal).
//
// Replaces the OptiX geometry-acceleration structure the reference rebuilds every iteration
// (render/optixutils/c_src/torch_bindings.cpp:37-116, called from geometry/gshell_tets_geometry.py:211)
// and the hardware any-hit query `optixTrace(... TERMINATE_ON_FIRST_HIT ...)`
// (render/optixutils/c_src/envsampling/kernel.cu:101-117).  CDNA4 has no ray-tracing units, so:
//   * build  = one 30-bit Morton radix sort of the centroids + a pointer-free complete 8-ary heap:
//              leaf i owns sorted triangle i, node n has children 8n+1..8n+8 (one octree level of the Morton
//              order per tree level), so "refit" is D tiny launches of min/max over 8 slots -- no atomics, no
//              fences, no parent pointers; the whole rebuild is a handful of launches per iteration.
//   * layout = per internal node ONE 96-byte record holding its 8 child boxes in SoA form as 48 half floats:
//              six 16-byte groups lo.x[8] lo.y[8] lo.z[8] hi.x[8] hi.y[8] hi.z[8] (lo rounded down / hi rounded
//              up, so the boxes only grow).  A ray picks the NEAR and FAR plane group of each axis by the sign
//              of its direction once (a byte offset), so the slab test of a child is 6 fma + max3 + min3 with no
//              per-plane min / max.  Triangles are stored pre-gathered in sorted order as (v0, e1, e2) = 3 x
//              float4 and tested in full fp32: hits are unchanged by the rounding.
//   * why 8 = the traversal kernel is VALU-issue bound (rocprofv3: VALU busy 96 %, profiles/r03_pmc_trace.json):
//              its cost is (steps per ray) x (instructions per step), and shadow rays from a surface almost all MISS
//              (0.5 % hits on the bench mesh): a miss must visit every internal node whose box the ray pierces --
//              its own ancestors plus their pierced siblings.  The 4-ary heap (depth 9 on 2.3 10^5 triangles) cost
//              32 node + 2 triangle visits per ray at ~220 instructions per step; the 8-ary one has 3/7 of the
//              internal nodes and a fixed per-step overhead (stack, addressing, the divergent triangle block) paid
//              half as often.
//   * traverse = see BvhRay below: one record per step, branch-free slab test of the 8 children, pending work
//              kept as (node, child mask) with one 32-bit word per tree level in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

struct gs_bvh {
    int64_t T = 0;           // triangles in the current build
    int depth = 0;           // leaves live at heap level `depth` (>= 1)
    int leaf = 1;            // triangles per leaf (always 1)
    int64_t n_internal = 0;  // (8^depth - 1) / 7
    int64_t n_leaf = 0;      // T
    float4* groups = nullptr;   // [n_internal * 12]  fp32 child boxes (build buffer)
    uint4* nodes = nullptr;     // [n_internal * 6]  half-float child boxes, 96-byte records (traversal)
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
    const uint4* nodes;
    const float4* tris;
    int64_t T, n_internal, n_leaf;
    int leaf;
};

static inline BvhView bvh_view(const gs_bvh* b) { return {b->nodes, b->tris, b->T, b->n_internal, b->n_leaf, b->leaf}; }

constexpr int BVH_STACK = 12;  // entries per lane: one (node, pending-children mask) word per tree level (checked at build time)

// Moeller-Trumbore, t in (0, 1e16); the record is (v0, e1, e2) as 3 x float4
__device__ __forceinline__ bool tri_hit(float4 v0, float4 e1, float4 e2, float ox, float oy, float oz, float dx, float dy, float dz) {
    float px = dy * e2.z - dz * e2.y, py = dz * e2.x - dx * e2.z, pz = dx * e2.y - dy * e2.x;
    float det = e1.x * px + e1.y * py + e1.z * pz;
    if (!(fabsf(det) > 1e-20f)) return false;
    float inv = __fdiv_rn(1.0f, det);          // IEEE whatever the file's divide mode: hits are compared bit for bit with the host oracle
    float tx = ox - v0.x, ty = oy - v0.y, tz = oz - v0.z;
    float u = (tx * px + ty * py + tz * pz) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    float qx = ty * e1.z - tz * e1.y, qy = tz * e1.x - tx * e1.z, qz = tx * e1.y - ty * e1.x;
    float v = (dx * qx + dy * qy + dz * qz) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    float t = (e2.x * qx + e2.y * qy + e2.z * qz) * inv;
    return t > 0.0f && t < 1e16f;
}
__device__ __forceinline__ float4 bvh_as_float4(uint4 q) {
    return make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w));
}

// Traversal state of one shadow ray (registers).  The loop is instruction-issue bound (a wave64 VALU op takes
// 4 cycles and the lanes of a wave follow unrelated paths), so a step is written short and nearly branch-free:
//   * every visit is ONE record fetched from a lane-dependent base (96-byte node or 48-byte triangle): one memory
//     latency per step however the wave is split;
//   * the eight child slabs are tested with fma (origin pre-multiplied by 1/d) on the near / far plane groups the
//     ray selected at start, max3 / min3, no early outs;
//   * pending work is (node, bit mask of its children still to visit): the current pair lives in registers and
//     ONE 32-bit word (node << 8 | mask) per tree level is spilled to LDS, so the stack is `depth` entries deep
//     and occupancy is not limited by LDS.
// For a node whose children are leaves the mask bits index the 8 triangles below it.
// Entry e of a lane lives at st[e * nthreads] (st = stack base + the lane's thread index).
constexpr int BVH_W = 8;
struct BvhRay {
    float ox, oy, oz, dx, dy, dz, ix, iy, iz, nox, noy, noz;
    int32_t node;    // -1 = virtual parent of the root
    uint32_t mask;   // children of `node` not visited yet (never 0 between steps)
    int sp;
    uint32_t sel;    // six 3-bit group indices of a node record: near x, y, z, far x, y, z (near plane of an axis = the child's hi plane when d < 0)
};

// false = the direction is degenerate (zero / NaN): the ray hits nothing
__device__ __forceinline__ bool bvh_ray_init(BvhRay& r, float ox, float oy, float oz, float dx, float dy, float dz) {
    if (!(dx == dx && dy == dy && dz == dz) || (dx == 0.f && dy == 0.f && dz == 0.f)) return false;
    r.ox = ox; r.oy = oy; r.oz = oz; r.dx = dx; r.dy = dy; r.dz = dz;
    // finite reciprocals: a zero component would turn the fma slab test into inf - inf
    r.ix = fabsf(dx) > 1e-18f ? __fdiv_rn(1.0f, dx) : copysignf(1e18f, dx);
    r.iy = fabsf(dy) > 1e-18f ? __fdiv_rn(1.0f, dy) : copysignf(1e18f, dy);
    r.iz = fabsf(dz) > 1e-18f ? __fdiv_rn(1.0f, dz) : copysignf(1e18f, dz);
    r.nox = -ox * r.ix; r.noy = -oy * r.iy; r.noz = -oz * r.iz;
    // the six 16-byte groups of a node record this ray reads as near x, y, z and far x, y, z planes (3 bits each): group a + 3 [d_a < 0] / a + 3 [d_a >= 0]
    const uint32_t nx = r.ix < 0.f ? 3u : 0u, ny = r.iy < 0.f ? 4u : 1u, nz = r.iz < 0.f ? 5u : 2u;
    r.sel = nx | (ny << 3) | (nz << 6) | ((3u - nx) << 9) | ((5u - ny) << 12) | ((7u - nz) << 15);
    r.node = -1;
    r.mask = 1u;
    r.sp = 0;
    return true;
}

constexpr int BVH_CONTINUE = 0, BVH_MISS = 1, BVH_HIT = 2;

// half number `i` (0..7) of a 16-byte group
__device__ __forceinline__ float bvh_half8(const uint4& g, int i) {
    union { uint32_t u; _Float16 h[2]; } c;
    c.u = i < 2 ? g.x : (i < 4 ? g.y : (i < 6 ? g.z : g.w));
    return (float)c.h[i & 1];
}

// 8-bit mask of the children of internal node c whose (half-float, outward-rounded) boxes the ray pierces at some t >= 0
__device__ __forceinline__ uint32_t bvh_children_hit(const BvhView& bv, const BvhRay& r, int32_t c) {
    // groups of the record: 0..2 = lo.x lo.y lo.z, 3..5 = hi.x hi.y hi.z; near plane of axis a = group a + 3 [d_a < 0].
    // Addresses as UNIFORM base + 32-bit lane offset (record c at byte 96 c; the near / far group of an axis 48 bytes apart, chosen by
    // the direction's signs once per ray, bvh_ray_init): a multiply + six (bit-field extract, shift-add) pairs -- 64-bit pointer arithmetic per group was ~36 of the ~140
    // instructions of a node visit (the kernel is bound by instruction issue).
    const char* base = reinterpret_cast<const char*>(bv.nodes);
    const uint32_t o = (uint32_t)c * 96u;
    auto group = [&](int k) { return *reinterpret_cast<const uint4*>(base + (size_t)(o + (((r.sel >> (3 * k)) & 7u) << 4))); };
    const uint4 nx = group(0), ny = group(1), nz = group(2), fx = group(3), fy = group(4), fz = group(5);
    uint32_t m2 = 0u;
#pragma unroll
    for (int j = 0; j < BVH_W; ++j) {
        const float tnx = __builtin_fmaf(bvh_half8(nx, j), r.ix, r.nox), tfx = __builtin_fmaf(bvh_half8(fx, j), r.ix, r.nox);
        const float tny = __builtin_fmaf(bvh_half8(ny, j), r.iy, r.noy), tfy = __builtin_fmaf(bvh_half8(fy, j), r.iy, r.noy);
        const float tnz = __builtin_fmaf(bvh_half8(nz, j), r.iz, r.noz), tfz = __builtin_fmaf(bvh_half8(fz, j), r.iz, r.noz);
        // an empty slot has lo > hi on every axis: its near plane lies behind its far plane for either sign of d -> tf < tn
        const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
        const float tf = fminf(fminf(tfx, tfy), tfz);
        m2 |= (tf >= tn) ? (1u << j) : 0u;
    }
    return m2;
}

// does the box of child k (0..7) of internal node c contain the point?  (an empty slot, lo > hi, contains nothing)
__device__ __forceinline__ bool bvh_child_contains(const BvhView& bv, int32_t c, int k, float x, float y, float z) {
    const _Float16* h = reinterpret_cast<const _Float16*>(bv.nodes + (int64_t)c * 6);
    return (float)h[k] <= x && x <= (float)h[24 + k] && (float)h[8 + k] <= y && y <= (float)h[32 + k] && (float)h[16 + k] <= z && z <= (float)h[40 + k];
}

// visits the lowest pending child of r.node.  Returns BVH_CONTINUE / BVH_MISS (nothing pending) / BVH_HIT.
template <bool STATS = false>
__device__ __forceinline__ int bvh_step(const BvhView& bv, BvhRay& r, int32_t* st, int nthreads, int* n_nodes = nullptr, int* n_tris = nullptr) {
    const uint4* __restrict__ tri_rec = reinterpret_cast<const uint4*>(bv.tris);
    const int32_t n_internal = (int32_t)bv.n_internal;
    const int k = __builtin_ctz(r.mask);
    uint32_t mask = r.mask & (r.mask - 1u);
    const int32_t c0 = BVH_W * r.node + 1;                              // first child of r.node (-7 for the virtual parent)
    const bool at_tris = c0 >= n_internal;                              // r.node's children are leaves: bit k = k-th triangle below it
    const int32_t c = r.node < 0 ? 0 : c0 + k;                          // node to visit (when !at_tris)
    const int32_t t = (c0 - n_internal) + k;                            // triangle to test (when at_tris)
    if (at_tris) {
        if (STATS) ++*n_tris;
        const uint4* rec = tri_rec + (int64_t)t * 3;
        const uint4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
        if (tri_hit(bvh_as_float4(q0), bvh_as_float4(q1), bvh_as_float4(q2), r.ox, r.oy, r.oz, r.dx, r.dy, r.dz)) return BVH_HIT;
    } else {
        if (STATS) ++*n_nodes;
        const uint32_t m2 = bvh_children_hit(bv, r, c);
        if (m2 != 0u) {
            if (mask != 0u) {
                st[min(r.sp, BVH_STACK - 1) * nthreads] = (int32_t)(((uint32_t)r.node << 8) | mask);
                ++r.sp;
            }
            r.node = c;
            mask = m2;
        }
    }
    if (mask == 0u) {
        if (r.sp == 0) return BVH_MISS;
        --r.sp;
        const uint32_t e = (uint32_t)st[r.sp * nthreads];
        r.node = (int32_t)(e >> 8);
        mask = e & 0xffu;
    }
    r.mask = mask;
    return BVH_CONTINUE;
}

// true if the ray (o, d), t in (0, 1e16), hits any triangle.  `stack` = this block's LDS stack base.
template <bool STATS = false>
__device__ __forceinline__ bool bvh_any_hit(const BvhView& bv, float ox, float oy, float oz, float dx, float dy, float dz, int32_t* stack,
                                            int tid, int nthreads, int* n_nodes = nullptr, int* n_tris = nullptr) {
    if (bv.T <= 0) return false;
    BvhRay r;
    if (!bvh_ray_init(r, ox, oy, oz, dx, dy, dz)) return false;
    int32_t* const st = stack + tid;
    int state;
    do {
        state = bvh_step<STATS>(bv, r, st, nthreads, n_nodes, n_tris);
    } while (state == BVH_CONTINUE);
    return state == BVH_HIT;
}

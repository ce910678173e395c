// Triangle rasteriser + barycentric interpolation on gfx950 (fwd + bwd).
//
// Replaces, for the G-Shell training path, the third-party nvdiffrast calls made by the
// reference at render/render.py:377-379 (DepthPeeler first layer == z-buffered rasterize),
// render/render.py:25-26 (dr.interpolate) and the vertex stage ru.xfm_points
// (render/renderutils/ops.py:518, c_src/mesh.cu:22-94).  nvdiffrast is not in the reference tree,
// so the semantics are restated from its public description (SURVEY.md 8c [3P-memory]):
//   * pixel (x,y) centre <-> NDC ((x+.5)/W*2-1, (y+.5)/H*2-1), row 0 = NDC y -1
//   * output rast = (u, v, z/w, triangle_id+1), u/v = perspective-correct barycentrics of
//     vertices 0/1, 0 in all channels = empty; rast_db = (du/dX, du/dY, dv/dX, dv/dY)
//   * nearest z/w wins; here ties are broken by the LOWER triangle id (deterministic)
//
// MI355X design: G-Shell meshes at tet-res 128/256 are ~1e5-1e6 triangles of a few pixels each,
// so this is a micro-triangle rasteriser, not a tile binner: one lane per (view, triangle) walks
// the triangle's pixel bounding box with exact fixed-point edge functions (8 sub-pixel bits,
// 64-bit integers -> watertight, top-left rule exact) and resolves visibility with a single
// 64-bit atomicMin per covered sample into an HBM z-buffer key (ordered depth bits << 32 | id).
// Triangles with a large bounding box are deferred to a queue and rasterised by a whole
// workgroup each.  A per-pixel resolve pass recomputes barycentrics, z/w and their pixel
// derivatives in fp32 from the clip-space vertices (2-D homogeneous form), which is also the
// function whose analytic adjoint the backward kernel evaluates.
//
// GS_CXXFLAGS: -fno-slp-vectorize
// ^ per-file compiler flag (csrc/Makefile): no packed-fp32 instructions in this file.  Round 6, MI355X: with hipcc's SLP vectoriser the per-sample
// barycentric / depth arithmetic of k_rast_small becomes v_pk_mul_f32 / v_pk_add_f32 (op_sel / neg modifiers).  Stand-alone that code is deterministic
// (thousands of frames, bit for bit).  While k_h2_fwd / k_h2_bwd of csrc/mlp_h2.hip run on ANOTHER HIP stream, single samples of a frame come out with a
// wrong depth -- right pixel, right triangle, one packed product wrong (a0 of bary_eval in the self-checking build, GS_RAST_DEBUG=2) -- in 13 % / 71 % of
// the frames (tools/raster_race_probe6.py), never under k_h1_fwd, k_h2_wgrad16, hipBLASLt GEMMs, a device copy or a synthetic scratch kernel, never with
// agent- / system-scope z-buffer accesses making a difference (GS_RAST_COHERENT), never after a register-file poison (no uninitialised read), and never
// (0 of 1000 frames, 0 of 25 chain-test runs with the eikonal side stream on) once this file is compiled without packed-fp32 instructions.  Same IEEE
// results either way (-ffp-contract=off); the rasteriser is 0.06 ms of a 14.5 ms iteration.  The record: profiles/r06_two_queue_probes.txt; DESIGN.md 5.4.
#include <hip/hip_runtime.h>

#include "../../include/gshell_hip.h"
#include "common.hpp"

namespace {

constexpr int SUBPIX_BITS = 8;
constexpr int SUBPIX = 1 << SUBPIX_BITS;
constexpr int LARGE_BBOX = 1024;  // pixels; larger bounding boxes go to the workgroup-per-triangle path
constexpr float W_EPS = 1e-6f;

// Diagnostic switch (tools/raster_race_probe.py): how the z-buffer words are written / read around the atomicMin pass.
//   bit 0: k_rast_resolve reads the keys with agent-scope loads;  bit 1: k_rast_clear writes them with agent-scope stores;  bit 2: system-scope atomicMin
#ifndef GS_RAST_COHERENT
#define GS_RAST_COHERENT 0
#endif
GS_TUNABLE(GS_RAST_COHERENT, 0)

// Diagnostic build (tools/raster_race_probe3.py): every sample k_rast_small issues is also logged with the operands its depth came from.
#ifndef GS_RAST_DEBUG
#define GS_RAST_DEBUG 0
#endif
GS_TUNABLE(GS_RAST_DEBUG, 0)
constexpr int64_t RAST_DEBUG_CAP = 1 << 20;          // records of 16 words

__device__ __forceinline__ void zkey_min(uint64_t* p, uint64_t key) {
#if GS_RAST_COHERENT & 4
    __hip_atomic_fetch_min((unsigned long long*)p, (unsigned long long)key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
    atomicMin((unsigned long long*)p, (unsigned long long)key);
#endif
}

struct TriFix {  // fixed-point setup of one projected triangle
    int32_t x[3], y[3];
    int64_t area2;
};

// screen-space fixed-point position of a clip-space vertex; false if not representable
__device__ __forceinline__ bool project_fix(const float4 p, int H, int W, int32_t& ix, int32_t& iy) {
    if (!(p.w > W_EPS)) return false;
    float xn = p.x / p.w, yn = p.y / p.w;
    float sx = (xn * 0.5f + 0.5f) * (float)W;
    float sy = (yn * 0.5f + 0.5f) * (float)H;
    float fx = floorf(sx * (float)SUBPIX + 0.5f), fy = floorf(sy * (float)SUBPIX + 0.5f);
    if (!(fabsf(fx) < 16777216.0f) || !(fabsf(fy) < 16777216.0f)) return false;
    ix = (int32_t)fx;
    iy = (int32_t)fy;
    return true;
}

__device__ __forceinline__ uint32_t depth_key(float zw) {  // order-preserving float -> uint
    uint32_t u = __float_as_uint(zw);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// perspective-correct barycentrics / depth of the pixel centre (fx,fy in NDC), 2-D homogeneous form
struct Bary {
    float b0, b1, zw, s;  // s = a0+a1+a2
    float a0, a1, a2;
};
__device__ __forceinline__ Bary bary_eval(const float4 p0, const float4 p1, const float4 p2, float fx, float fy) {
    float p0x = p0.x - fx * p0.w, p0y = p0.y - fy * p0.w;
    float p1x = p1.x - fx * p1.w, p1y = p1.y - fy * p1.w;
    float p2x = p2.x - fx * p2.w, p2y = p2.y - fy * p2.w;
    Bary r;
    r.a0 = p1x * p2y - p1y * p2x;
    r.a1 = p2x * p0y - p2y * p0x;
    r.a2 = p0x * p1y - p0y * p1x;
    r.s = r.a0 + r.a1 + r.a2;
    float iw = 1.0f / r.s;
    r.b0 = r.a0 * iw;
    r.b1 = r.a1 * iw;
    float z = p0.z * r.a0 + p1.z * r.a1 + p2.z * r.a2;
    float w = p0.w * r.a0 + p1.w * r.a1 + p2.w * r.a2;
    r.zw = z / w;
    return r;
}

__device__ __forceinline__ float pix_ndc(int p, int n) { return ((float)p + 0.5f) * (2.0f / (float)n) - 1.0f; }

// edge i runs from vertex (i+1)%3 to (i+2)%3; E_i(v_i) == area2
struct EdgeEq {
    int64_t A, B, C;  // E(cx,cy) = A*cx + B*cy + C   (sub-pixel units)
    bool own;         // owns samples lying exactly on the edge (top-left rule)
};
__device__ __forceinline__ EdgeEq edge_setup(int32_t ax, int32_t ay, int32_t bx, int32_t by, int sgn) {
    EdgeEq e;
    int64_t dx = (int64_t)(bx - ax) * sgn, dy = (int64_t)(by - ay) * sgn;
    e.A = -dy;
    e.B = dx;
    e.C = dy * ax - dx * ay;
    e.own = (dy > 0) || (dy == 0 && dx > 0);
    return e;
}
__device__ __forceinline__ bool edge_in(const EdgeEq& e, int64_t v) { return v > 0 || (v == 0 && e.own); }

__device__ __forceinline__ void raster_sample(uint64_t* __restrict__ zrow, int px, int py, int H, int W, const float4 p0,
                                              const float4 p1, const float4 p2, uint32_t tri_id, int32_t* dbg = nullptr, int view = 0) {
    Bary r = bary_eval(p0, p1, p2, pix_ndc(px, W), pix_ndc(py, H));
#if GS_RAST_DEBUG == 1
    if (dbg) {
        int32_t slot = atomicAdd(dbg, 1);
        if (slot < RAST_DEBUG_CAP) {
            int32_t* q = dbg + 2 + 16 * (int64_t)slot;
            q[0] = px; q[1] = py; q[2] = view; q[3] = (int32_t)tri_id; q[4] = __float_as_int(r.zw); q[5] = __float_as_int(pix_ndc(px, W)); q[6] = __float_as_int(pix_ndc(py, H));
            q[7] = __float_as_int(p0.z); q[8] = __float_as_int(p1.z); q[9] = __float_as_int(p2.z); q[10] = __float_as_int(r.a0); q[11] = __float_as_int(r.a1);
            q[12] = __float_as_int(r.a2); q[13] = W; q[14] = H; q[15] = __float_as_int(p0.w);
        }
    }
#elif GS_RAST_DEBUG == 2
    // self-check, logged only when it fails: the same depth from operands the compiler must treat as new (asm barriers), bit for bit
    if (dbg) {
        float4 q0 = p0, q1 = p1, q2 = p2;
        asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w));
        asm volatile("" : "+v"(q2.x), "+v"(q2.y), "+v"(q2.z), "+v"(q2.w));
        Bary r2 = bary_eval(q0, q1, q2, pix_ndc(px, W), pix_ndc(py, H));
        float z1 = p0.z * r.a0 + p1.z * r.a1 + p2.z * r.a2, w1 = p0.w * r.a0 + p1.w * r.a1 + p2.w * r.a2;
        asm volatile("" : "+v"(z1), "+v"(w1));
        const float zw3 = z1 / w1;
        if (__float_as_int(r2.zw) != __float_as_int(r.zw) || __float_as_int(zw3) != __float_as_int(r.zw)) {
            int32_t slot = atomicAdd(dbg, 1);
            if (slot < RAST_DEBUG_CAP) {
                int32_t* q = dbg + 2 + 16 * (int64_t)slot;
                q[0] = px; q[1] = py; q[2] = view; q[3] = (int32_t)tri_id; q[4] = __float_as_int(r.zw); q[5] = __float_as_int(r2.zw); q[6] = __float_as_int(zw3);
                q[7] = __float_as_int(z1); q[8] = __float_as_int(w1); q[9] = __float_as_int(r.a0); q[10] = __float_as_int(r2.a0); q[11] = __float_as_int(r.a1);
                q[12] = __float_as_int(r2.a1); q[13] = __float_as_int(r.a2); q[14] = __float_as_int(r2.a2); q[15] = __float_as_int(r.s);
            }
        }
    }
#endif
    if (!(r.zw >= -1.0f && r.zw <= 1.0f)) return;  // near/far clip (also rejects NaN)
    uint64_t key = ((uint64_t)depth_key(r.zw) << 32) | tri_id;
    zkey_min(&zrow[px], key);
}

struct TriJob {
    float4 p0, p1, p2;
    EdgeEq e0, e1, e2;
    int x0, x1, y0, y1;  // inclusive pixel bbox (clamped); empty if x0 > x1
};

enum { TRI_REJECT = 0, TRI_FIXED = 1, TRI_NEAR_CLIPPED = 2 };

// A triangle with a vertex at w <= eps (behind or at the eye) cannot be projected vertex by vertex.  nvdiffrast clips it
// against the view volume; here the part in front of the near plane (z >= -w) is rasterised by evaluating the 2-D
// homogeneous edge functions per pixel (no clipped geometry is ever built): this routine only bounds the pixels to visit
// -- the screen-space box of the polygon "triangle  intersected with  z + w >= 0" (its corners all have w > 0 for a
// perspective projection with a positive near distance; any corner that does not falls back to the whole screen).
__device__ __forceinline__ int near_clip_bbox(const float4 (&p)[3], int H, int W, TriJob& j) {
    float xmin = 3.0e38f, xmax = -3.0e38f, ymin = 3.0e38f, ymax = -3.0e38f;
    bool any = false, whole = false;
    auto add = [&](float x, float y, float w) {
        any = true;
        if (!(w > W_EPS)) { whole = true; return; }
        const float sx = (x / w * 0.5f + 0.5f) * (float)W, sy = (y / w * 0.5f + 0.5f) * (float)H;
        xmin = fminf(xmin, sx); xmax = fmaxf(xmax, sx); ymin = fminf(ymin, sy); ymax = fmaxf(ymax, sy);
    };
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float4 a = p[i], b = p[(i + 1) % 3];
        const float da = a.z + a.w, db = b.z + b.w;        // signed distance to the near plane in clip space
        if (da >= 0.0f) add(a.x, a.y, a.w);
        if ((da >= 0.0f) != (db >= 0.0f)) {
            const float t = da / (da - db);
            add(a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.w + t * (b.w - a.w));
        }
    }
    if (!any) return TRI_REJECT;                             // entirely behind the near plane
    if (whole || !(xmin == xmin) || !(ymin == ymin)) { xmin = 0.f; ymin = 0.f; xmax = (float)W; ymax = (float)H; }
    j.x0 = max(0, (int)floorf(fminf(fmaxf(xmin, -1.0f), (float)W + 1.0f)) - 1);
    j.x1 = min(W - 1, (int)ceilf(fminf(fmaxf(xmax, -1.0f), (float)W + 1.0f)) + 1);
    j.y0 = max(0, (int)floorf(fminf(fmaxf(ymin, -1.0f), (float)H + 1.0f)) - 1);
    j.y1 = min(H - 1, (int)ceilf(fminf(fmaxf(ymax, -1.0f), (float)H + 1.0f)) + 1);
    return (j.x0 <= j.x1 && j.y0 <= j.y1) ? TRI_NEAR_CLIPPED : TRI_REJECT;
}

// coverage of a near-clipped triangle at one pixel centre: inside the triangle's plane polygon (all homogeneous barycentrics
// >= 0), in front of the eye, and inside the depth range (raster_sample's test)
__device__ __forceinline__ void raster_sample_homogeneous(uint64_t* __restrict__ zrow, int px, int py, int H, int W, const float4 p0,
                                                          const float4 p1, const float4 p2, uint32_t tri_id) {
    Bary r = bary_eval(p0, p1, p2, pix_ndc(px, W), pix_ndc(py, H));
    if (!(r.s != 0.0f)) return;
    const float b2 = 1.0f - r.b0 - r.b1;
    if (!(r.b0 >= 0.0f && r.b1 >= 0.0f && b2 >= 0.0f)) return;
    const float w = p0.w * r.b0 + p1.w * r.b1 + p2.w * b2;    // clip-space w of the point hit by the pixel's ray
    if (!(w > 0.0f) || !(r.zw >= -1.0f && r.zw <= 1.0f)) return;
    uint64_t key = ((uint64_t)depth_key(r.zw) << 32) | tri_id;
    zkey_min(&zrow[px], key);
}

__device__ __forceinline__ int tri_setup(const float4* __restrict__ pos, const int32_t* __restrict__ tri, int64_t t, int64_t V, int H,
                                         int W, TriJob& j) {
    int32_t i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
    if ((uint32_t)i0 >= (uint32_t)V || (uint32_t)i1 >= (uint32_t)V || (uint32_t)i2 >= (uint32_t)V) return TRI_REJECT;
    j.p0 = pos[i0];
    j.p1 = pos[i1];
    j.p2 = pos[i2];
    const int n_front = (j.p0.w > W_EPS) + (j.p1.w > W_EPS) + (j.p2.w > W_EPS);
    if (n_front == 0) return TRI_REJECT;
    if (n_front < 3) {
        const float4 p[3] = {j.p0, j.p1, j.p2};
        return near_clip_bbox(p, H, W, j);
    }
    int32_t x[3], y[3];
    if (!project_fix(j.p0, H, W, x[0], y[0]) || !project_fix(j.p1, H, W, x[1], y[1]) || !project_fix(j.p2, H, W, x[2], y[2])) return TRI_REJECT;
    int64_t area2 = (int64_t)(x[1] - x[0]) * (y[2] - y[0]) - (int64_t)(y[1] - y[0]) * (x[2] - x[0]);
    if (area2 == 0) return TRI_REJECT;
    int sgn = area2 > 0 ? 1 : -1;
    j.e0 = edge_setup(x[1], y[1], x[2], y[2], sgn);
    j.e1 = edge_setup(x[2], y[2], x[0], y[0], sgn);
    j.e2 = edge_setup(x[0], y[0], x[1], y[1], sgn);
    int32_t mnx = min(x[0], min(x[1], x[2])), mxx = max(x[0], max(x[1], x[2]));
    int32_t mny = min(y[0], min(y[1], y[2])), mxy = max(y[0], max(y[1], y[2]));
    // pixel p covers sub-pixel centre p*256+128: first centre >= mn, last centre <= mx (arithmetic shift = floor)
    j.x0 = max(0, (mnx - SUBPIX / 2 + SUBPIX - 1) >> SUBPIX_BITS);
    j.x1 = min(W - 1, (mxx - SUBPIX / 2) >> SUBPIX_BITS);
    j.y0 = max(0, (mny - SUBPIX / 2 + SUBPIX - 1) >> SUBPIX_BITS);
    j.y1 = min(H - 1, (mxy - SUBPIX / 2) >> SUBPIX_BITS);
    return (j.x0 <= j.x1 && j.y0 <= j.y1) ? TRI_FIXED : TRI_REJECT;
}

__global__ void __launch_bounds__(256) k_rast_clear(uint64_t* __restrict__ zbuf, int64_t n, int32_t* __restrict__ qcount, int32_t* __restrict__ dbg = nullptr) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *qcount = 0;
#if GS_RAST_DEBUG
    if (i == 0 && dbg) *dbg = 0;
#endif
#if GS_RAST_COHERENT & 2
    for (; i < n; i += (int64_t)gridDim.x * blockDim.x) __hip_atomic_store((unsigned long long*)&zbuf[i], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    for (; i < n; i += (int64_t)gridDim.x * blockDim.x) zbuf[i] = ~0ull;
#endif
}

__global__ void __launch_bounds__(256) k_rast_small(const float4* __restrict__ pos, const int32_t* __restrict__ tri, int64_t B, int64_t V,
                                                    int64_t T, int H, int W, uint64_t* __restrict__ zbuf, int32_t* __restrict__ qcount,
                                                    int64_t* __restrict__ queue) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * T) return;
    int64_t b = idx / T, t = idx - b * T;
    TriJob j;
    const int kind = tri_setup(pos + b * V, tri, t, V, H, W, j);
    if (kind == TRI_REJECT) return;
    int64_t area = (int64_t)(j.x1 - j.x0 + 1) * (j.y1 - j.y0 + 1);
    if (area > LARGE_BBOX || kind == TRI_NEAR_CLIPPED) {      // near-clipped triangles always take the workgroup path
        int32_t slot = atomicAdd(qcount, 1);
        queue[slot] = idx;
        return;
    }
    uint64_t* zview = zbuf + b * (int64_t)H * W;
    const int64_t cx0 = (int64_t)j.x0 * SUBPIX + SUBPIX / 2;
    for (int py = j.y0; py <= j.y1; ++py) {
        int64_t cy = (int64_t)py * SUBPIX + SUBPIX / 2;
        int64_t v0 = j.e0.A * cx0 + j.e0.B * cy + j.e0.C;
        int64_t v1 = j.e1.A * cx0 + j.e1.B * cy + j.e1.C;
        int64_t v2 = j.e2.A * cx0 + j.e2.B * cy + j.e2.C;
        for (int px = j.x0; px <= j.x1; ++px) {
            if (edge_in(j.e0, v0) && edge_in(j.e1, v1) && edge_in(j.e2, v2))
#if GS_RAST_DEBUG
                raster_sample(zview + (int64_t)py * W, px, py, H, W, j.p0, j.p1, j.p2, (uint32_t)t, (int32_t*)(queue + B * T), (int)b);
#else
                raster_sample(zview + (int64_t)py * W, px, py, H, W, j.p0, j.p1, j.p2, (uint32_t)t);
#endif
            v0 += j.e0.A * SUBPIX;
            v1 += j.e1.A * SUBPIX;
            v2 += j.e2.A * SUBPIX;
        }
    }
}

__global__ void __launch_bounds__(256) k_rast_large(const float4* __restrict__ pos, const int32_t* __restrict__ tri, int64_t V, int64_t T,
                                                    int H, int W, uint64_t* __restrict__ zbuf, const int32_t* __restrict__ qcount,
                                                    const int64_t* __restrict__ queue) {
    int32_t n = *qcount;
    for (int32_t q = blockIdx.x; q < n; q += gridDim.x) {
        int64_t idx = queue[q];
        int64_t b = idx / T, t = idx - b * T;
        TriJob j;
        const int kind = tri_setup(pos + b * V, tri, t, V, H, W, j);  // uniform across the block
        if (kind == TRI_REJECT) continue;
        uint64_t* zview = zbuf + b * (int64_t)H * W;
        int bw = j.x1 - j.x0 + 1, bh = j.y1 - j.y0 + 1;
        if (kind == TRI_NEAR_CLIPPED) {
            for (int i = threadIdx.x; i < bw * bh; i += blockDim.x) {
                int py = j.y0 + i / bw, px = j.x0 + i % bw;
                raster_sample_homogeneous(zview + (int64_t)py * W, px, py, H, W, j.p0, j.p1, j.p2, (uint32_t)t);
            }
            continue;
        }
        for (int i = threadIdx.x; i < bw * bh; i += blockDim.x) {
            int py = j.y0 + i / bw, px = j.x0 + i % bw;
            int64_t cx = (int64_t)px * SUBPIX + SUBPIX / 2, cy = (int64_t)py * SUBPIX + SUBPIX / 2;
            int64_t v0 = j.e0.A * cx + j.e0.B * cy + j.e0.C;
            int64_t v1 = j.e1.A * cx + j.e1.B * cy + j.e1.C;
            int64_t v2 = j.e2.A * cx + j.e2.B * cy + j.e2.C;
            if (edge_in(j.e0, v0) && edge_in(j.e1, v1) && edge_in(j.e2, v2))
                raster_sample(zview + (int64_t)py * W, px, py, H, W, j.p0, j.p1, j.p2, (uint32_t)t);
        }
    }
}

// per pixel: winner id -> (u, v, z/w, id+1) and (du/dX, du/dY, dv/dX, dv/dY)
__global__ void __launch_bounds__(256) k_rast_resolve(const float4* __restrict__ pos, const int32_t* __restrict__ tri, int64_t B, int64_t V,
                                                      int H, int W, const uint64_t* __restrict__ zbuf, float4* __restrict__ rast,
                                                      float4* __restrict__ rast_db, uint8_t* __restrict__ tri_visible) {
    int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t npix = B * (int64_t)H * W;
    if (pix >= npix) return;
#if GS_RAST_COHERENT & 1
    uint64_t key = __hip_atomic_load((const unsigned long long*)&zbuf[pix], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    uint64_t key = zbuf[pix];
#endif
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f), d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (key != ~0ull) {
        uint32_t t = (uint32_t)(key & 0xffffffffu);
        int64_t b = pix / ((int64_t)H * W);
        int rem = (int)(pix - b * (int64_t)H * W);
        int py = rem / W, px = rem - py * W;
        const float4* pv = pos + b * V;
        float4 p0 = pv[tri[3 * (int64_t)t]], p1 = pv[tri[3 * (int64_t)t + 1]], p2 = pv[tri[3 * (int64_t)t + 2]];
        float fx = pix_ndc(px, W), fy = pix_ndc(py, H);
        Bary q = bary_eval(p0, p1, p2, fx, fy);
        float iw = 1.0f / q.s;
        // d a_i / d fx, d a_i / d fy   (p_ix = x_i - fx w_i, p_iy = y_i - fy w_i)
        float p0x = p0.x - fx * p0.w, p0y = p0.y - fy * p0.w;
        float p1x = p1.x - fx * p1.w, p1y = p1.y - fy * p1.w;
        float p2x = p2.x - fx * p2.w, p2y = p2.y - fy * p2.w;
        float da0x = p1y * p2.w - p1.w * p2y, da0y = p1.w * p2x - p1x * p2.w;
        float da1x = p2y * p0.w - p2.w * p0y, da1y = p2.w * p0x - p2x * p0.w;
        float da2x = p0y * p1.w - p0.w * p1y, da2y = p0.w * p1x - p0x * p1.w;
        float dsx = da0x + da1x + da2x, dsy = da0y + da1y + da2y;
        float sxp = 2.0f / (float)W, syp = 2.0f / (float)H;  // d fx / d X, d fy / d Y
        d.x = (da0x - q.b0 * dsx) * iw * sxp;
        d.y = (da0y - q.b0 * dsy) * iw * syp;
        d.z = (da1x - q.b1 * dsx) * iw * sxp;
        d.w = (da1y - q.b1 * dsy) * iw * syp;
        r.x = fminf(fmaxf(q.b0, 0.0f), 1.0f);
        r.y = fminf(fmaxf(q.b1, 0.0f), 1.0f);
        r.z = fminf(fmaxf(q.zw, -1.0f), 1.0f);
        r.w = (float)(t + 1u);
        if (tri_visible) tri_visible[t] = 1;
    }
    rast[pix] = r;
    if (rast_db) rast_db[pix] = d;
}

// d loss / d clip-space vertices from d loss / d (u, v); z/w and the id carry no gradient
__global__ void __launch_bounds__(256) k_rast_bwd(const float4* __restrict__ pos, const int32_t* __restrict__ tri, int64_t B, int64_t V, int H,
                                                  int W, const float4* __restrict__ rast, const float4* __restrict__ g_rast,
                                                  float* __restrict__ g_pos) {
    int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t npix = B * (int64_t)H * W;
    if (pix >= npix) return;
    float4 r = rast[pix];
    if (!(r.w > 0.0f)) return;
    float4 g = g_rast[pix];
    if (g.x == 0.0f && g.y == 0.0f) return;
    uint32_t t = (uint32_t)r.w - 1u;
    int64_t b = pix / ((int64_t)H * W);
    int rem = (int)(pix - b * (int64_t)H * W);
    int py = rem / W, px = rem - py * W;
    int32_t i0 = tri[3 * (int64_t)t], i1 = tri[3 * (int64_t)t + 1], i2 = tri[3 * (int64_t)t + 2];
    const float4* pv = pos + b * V;
    float4 p0 = pv[i0], p1 = pv[i1], p2 = pv[i2];
    float fx = pix_ndc(px, W), fy = pix_ndc(py, H);
    Bary q = bary_eval(p0, p1, p2, fx, fy);
    float iw = 1.0f / q.s;
    float ga0 = (g.x * (1.0f - q.b0) - g.y * q.b1) * iw;
    float ga1 = (g.y * (1.0f - q.b1) - g.x * q.b0) * iw;
    float ga2 = (-g.x * q.b0 - g.y * q.b1) * iw;
    float p0x = p0.x - fx * p0.w, p0y = p0.y - fy * p0.w;
    float p1x = p1.x - fx * p1.w, p1y = p1.y - fy * p1.w;
    float p2x = p2.x - fx * p2.w, p2y = p2.y - fy * p2.w;
    // a0 = p1x p2y - p1y p2x ; a1 = p2x p0y - p2y p0x ; a2 = p0x p1y - p0y p1x
    float g0x = ga2 * p1y - ga1 * p2y, g0y = ga1 * p2x - ga2 * p1x;
    float g1x = ga0 * p2y - ga2 * p0y, g1y = ga2 * p0x - ga0 * p2x;
    float g2x = ga1 * p0y - ga0 * p1y, g2y = ga0 * p1x - ga1 * p0x;
    float* gp = g_pos + b * V * 4;
    atomicAdd(&gp[4 * (int64_t)i0 + 0], g0x);
    atomicAdd(&gp[4 * (int64_t)i0 + 1], g0y);
    atomicAdd(&gp[4 * (int64_t)i0 + 3], -fx * g0x - fy * g0y);
    atomicAdd(&gp[4 * (int64_t)i1 + 0], g1x);
    atomicAdd(&gp[4 * (int64_t)i1 + 1], g1y);
    atomicAdd(&gp[4 * (int64_t)i1 + 3], -fx * g1x - fy * g1y);
    atomicAdd(&gp[4 * (int64_t)i2 + 0], g2x);
    atomicAdd(&gp[4 * (int64_t)i2 + 1], g2y);
    atomicAdd(&gp[4 * (int64_t)i2 + 3], -fx * g2x - fy * g2y);
}

// ---- xfm_points:  out[b,v,:] = [p,1] . M_b^T ------------------------------------------------------
__global__ void __launch_bounds__(256) k_xfm_fwd(const float* __restrict__ pts, int64_t Bp, const float* __restrict__ mtx, int64_t B,
                                                 int64_t V, float4* __restrict__ out) {
    __shared__ float m[16];
    int64_t b = blockIdx.y;
    if (threadIdx.x < 16) m[threadIdx.x] = mtx[b * 16 + threadIdx.x];
    __syncthreads();
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const float* p = pts + ((Bp == 1 ? 0 : b) * V + v) * 3;
    float x = p[0], y = p[1], z = p[2];
    float4 o;
    o.x = x * m[0] + y * m[1] + z * m[2] + m[3];
    o.y = x * m[4] + y * m[5] + z * m[6] + m[7];
    o.z = x * m[8] + y * m[9] + z * m[10] + m[11];
    o.w = x * m[12] + y * m[13] + z * m[14] + m[15];
    out[b * V + v] = o;
}

__global__ void __launch_bounds__(256) k_xfm_bwd(const float4* __restrict__ g_out, int64_t Bp, const float* __restrict__ mtx, int64_t B,
                                                 int64_t V, float* __restrict__ g_pts) {
    extern __shared__ float ms[];  // [B,16]
    for (int i = threadIdx.x; i < B * 16; i += blockDim.x) ms[i] = mtx[i];
    __syncthreads();
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    if (Bp == 1) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
        for (int64_t b = 0; b < B; ++b) {
            float4 g = g_out[b * V + v];
            const float* m = ms + b * 16;
            gx += g.x * m[0] + g.y * m[4] + g.z * m[8] + g.w * m[12];
            gy += g.x * m[1] + g.y * m[5] + g.z * m[9] + g.w * m[13];
            gz += g.x * m[2] + g.y * m[6] + g.z * m[10] + g.w * m[14];
        }
        g_pts[v * 3 + 0] = gx;
        g_pts[v * 3 + 1] = gy;
        g_pts[v * 3 + 2] = gz;
    } else {
        for (int64_t b = 0; b < B; ++b) {
            float4 g = g_out[b * V + v];
            const float* m = ms + b * 16;
            float* o = g_pts + (b * V + v) * 3;
            o[0] = g.x * m[0] + g.y * m[4] + g.z * m[8] + g.w * m[12];
            o[1] = g.x * m[1] + g.y * m[5] + g.z * m[9] + g.w * m[13];
            o[2] = g.x * m[2] + g.y * m[6] + g.z * m[10] + g.w * m[14];
        }
    }
}

// ---- interpolate ------------------------------------------------------------------------------------
// out[pix, c] = b0 a0[c] + b1 a1[c] + (1-b0-b1) a2[c];  out_da[pix, 2c+{0,1}] = d out / d{X,Y}
template <int A_STATIC>
__global__ void __launch_bounds__(256) k_interp_fwd(const float* __restrict__ attr, int64_t Ba, int64_t V, int A_dyn,
                                                    const float4* __restrict__ rast, const float4* __restrict__ rast_db,
                                                    const int32_t* __restrict__ tri, int64_t T, int64_t B, int64_t HW, float* __restrict__ out,
                                                    float* __restrict__ out_da) {
    const int A = A_STATIC > 0 ? A_STATIC : A_dyn;
    int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= B * HW) return;
    float4 r = rast[pix];
    float* o = out + pix * A;
    int32_t t = (int32_t)r.w - 1;
    if (t < 0 || t >= T) {
        for (int c = 0; c < A; ++c) o[c] = 0.f;
        if (out_da)
            for (int c = 0; c < 2 * A; ++c) out_da[pix * 2 * A + c] = 0.f;
        return;
    }
    int64_t b = Ba == 1 ? 0 : pix / HW;
    const float* base = attr + b * V * A;
    const float* a0 = base + (int64_t)tri[3 * (int64_t)t] * A;
    const float* a1 = base + (int64_t)tri[3 * (int64_t)t + 1] * A;
    const float* a2 = base + (int64_t)tri[3 * (int64_t)t + 2] * A;
    float b0 = r.x, b1 = r.y, b2 = 1.0f - r.x - r.y;
    float4 d = rast_db && out_da ? rast_db[pix] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < A; ++c) {
        float v0 = a0[c], v1 = a1[c], v2 = a2[c];
        o[c] = b0 * v0 + b1 * v1 + b2 * v2;
        if (out_da) {
            float e0 = v0 - v2, e1 = v1 - v2;
            out_da[pix * 2 * A + 2 * c] = d.x * e0 + d.z * e1;
            out_da[pix * 2 * A + 2 * c + 1] = d.y * e0 + d.w * e1;
        }
    }
}

__global__ void __launch_bounds__(256) k_interp_bwd(const float* __restrict__ attr, int64_t Ba, int64_t V, int A,
                                                    const float4* __restrict__ rast, const int32_t* __restrict__ tri, int64_t T, int64_t B,
                                                    int64_t HW, const float* __restrict__ g_out, float* __restrict__ g_attr,
                                                    float4* __restrict__ g_rast) {
    int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= B * HW) return;
    float4 r = rast[pix];
    int32_t t = (int32_t)r.w - 1;
    float gb0 = 0.f, gb1 = 0.f;
    if (t >= 0 && t < T) {
        int64_t b = Ba == 1 ? 0 : pix / HW;
        int64_t i0 = tri[3 * (int64_t)t], i1 = tri[3 * (int64_t)t + 1], i2 = tri[3 * (int64_t)t + 2];
        const float* base = attr + b * V * A;
        float* gbase = g_attr ? g_attr + b * V * A : nullptr;
        float b0 = r.x, b1 = r.y, b2 = 1.0f - r.x - r.y;
        const float* g = g_out + pix * A;
        for (int c = 0; c < A; ++c) {
            float gc = g[c];
            if (gc == 0.0f) continue;
            float v0 = base[i0 * A + c], v1 = base[i1 * A + c], v2 = base[i2 * A + c];
            gb0 += gc * (v0 - v2);
            gb1 += gc * (v1 - v2);
            if (gbase) {
                atomicAdd(&gbase[i0 * A + c], gc * b0);
                atomicAdd(&gbase[i1 * A + c], gc * b1);
                atomicAdd(&gbase[i2 * A + c], gc * b2);
            }
        }
    }
    if (g_rast) g_rast[pix] = make_float4(gb0, gb1, 0.f, 0.f);
}


// ---- interpolate, several per-vertex attribute tensors in one pass ------------------------------------------------------
// The g-buffer of a frame is position, smooth normal (and mSDF) interpolated with the same barycentrics (reference
// render/render.py:240, :263, :306 interpolates them one by one).  One launch, one contiguous output per attribute: no stacked [V,7] copy, no
// channel slices downstream (a slice of a stacked output costs a copy per consumer forward and a zero fill + copy + add backward).
// Per channel the arithmetic is k_interp_fwd / k_interp_bwd's, in the order of the stacked tensor: bit-identical results.
struct InterpGroups {
    const float* attr[4];
    float* out[4];
    const float* g_out[4];
    float* g_attr[4];
    int ch[4];
    int n;
};

__global__ void __launch_bounds__(256) k_interp_groups_fwd(InterpGroups G, const float4* __restrict__ rast, const int32_t* __restrict__ tri, int64_t T,
                                                           int64_t npix) {
    int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    const float4 r = rast[pix];
    const int32_t t = (int32_t)r.w - 1;
    const bool hit = t >= 0 && t < T;
    int64_t i0 = 0, i1 = 0, i2 = 0;
    if (hit) {
        i0 = tri[3 * (int64_t)t];
        i1 = tri[3 * (int64_t)t + 1];
        i2 = tri[3 * (int64_t)t + 2];
    }
    const float b0 = r.x, b1 = r.y, b2 = 1.0f - r.x - r.y;
    for (int k = 0; k < G.n; ++k) {
        const int A = G.ch[k];
        float* o = G.out[k] + pix * A;
        const float* a = G.attr[k];
        for (int c = 0; c < A; ++c) o[c] = hit ? b0 * a[i0 * A + c] + b1 * a[i1 * A + c] + b2 * a[i2 * A + c] : 0.f;
    }
}

__global__ void __launch_bounds__(256) k_interp_groups_bwd(InterpGroups G, const float4* __restrict__ rast, const int32_t* __restrict__ tri, int64_t T,
                                                           int64_t npix, float4* __restrict__ g_rast) {
    int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    const float4 r = rast[pix];
    const int32_t t = (int32_t)r.w - 1;
    float gb0 = 0.f, gb1 = 0.f;
    if (t >= 0 && t < T) {
        const int64_t i0 = tri[3 * (int64_t)t], i1 = tri[3 * (int64_t)t + 1], i2 = tri[3 * (int64_t)t + 2];
        const float b0 = r.x, b1 = r.y, b2 = 1.0f - r.x - r.y;
        for (int k = 0; k < G.n; ++k) {
            if (!G.g_out[k]) continue;
            const int A = G.ch[k];
            const float* g = G.g_out[k] + pix * A;
            const float* a = G.attr[k];
            float* ga = G.g_attr[k];
            for (int c = 0; c < A; ++c) {
                const float gc = g[c];
                if (gc == 0.0f) continue;
                const float v0 = a[i0 * A + c], v1 = a[i1 * A + c], v2 = a[i2 * A + c];
                gb0 += gc * (v0 - v2);
                gb1 += gc * (v1 - v2);
                if (ga) {
                    atomicAdd(&ga[i0 * A + c], gc * b0);
                    atomicAdd(&ga[i1 * A + c], gc * b1);
                    atomicAdd(&ga[i2 * A + c], gc * b2);
                }
            }
        }
    }
    if (g_rast) g_rast[pix] = make_float4(gb0, gb1, 0.f, 0.f);
}

// ---- per-pixel geometric (face) normal ---------------------------------------------------------------
// The reference interpolates a per-face constant with index [[i,i,i]] (render/render.py:243-248), i.e. a gather by
// triangle id.  n = cross(v1-v0, v2-v0) / sqrt(max(|.|^2, 1e-20))  (util.safe_normalize).
__global__ void __launch_bounds__(256) k_face_normal_fwd(const float* __restrict__ v, const int32_t* __restrict__ tri, int64_t T,
                                                         const float4* __restrict__ rast, int64_t npix, float* __restrict__ out) {
    int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    int32_t t = (int32_t)rast[pix].w - 1;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (t >= 0 && t < T) {
        const float* a = v + 3 * (int64_t)tri[3 * (int64_t)t];
        const float* b = v + 3 * (int64_t)tri[3 * (int64_t)t + 1];
        const float* c = v + 3 * (int64_t)tri[3 * (int64_t)t + 2];
        float e1x = b[0] - a[0], e1y = b[1] - a[1], e1z = b[2] - a[2];
        float e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];
        float cx = e1y * e2z - e1z * e2y, cy = e1z * e2x - e1x * e2z, cz = e1x * e2y - e1y * e2x;
        float l = sqrtf(fmaxf(cx * cx + cy * cy + cz * cz, 1e-20f));
        nx = cx / l;
        ny = cy / l;
        nz = cz / l;
    }
    out[3 * pix] = nx;
    out[3 * pix + 1] = ny;
    out[3 * pix + 2] = nz;
}

__global__ void __launch_bounds__(256) k_face_normal_bwd(const float* __restrict__ v, const int32_t* __restrict__ tri, int64_t T,
                                                         const float4* __restrict__ rast, int64_t npix, const float* __restrict__ g_out,
                                                         float* __restrict__ g_v) {
    int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    int32_t t = (int32_t)rast[pix].w - 1;
    if (t < 0 || t >= T) return;
    float gx = g_out[3 * pix], gy = g_out[3 * pix + 1], gz = g_out[3 * pix + 2];
    if (gx == 0.f && gy == 0.f && gz == 0.f) return;
    int64_t i0 = tri[3 * (int64_t)t], i1 = tri[3 * (int64_t)t + 1], i2 = tri[3 * (int64_t)t + 2];
    const float *a = v + 3 * i0, *b = v + 3 * i1, *c = v + 3 * i2;
    float e1x = b[0] - a[0], e1y = b[1] - a[1], e1z = b[2] - a[2];
    float e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];
    float cx = e1y * e2z - e1z * e2y, cy = e1z * e2x - e1x * e2z, cz = e1x * e2y - e1y * e2x;
    float l2 = cx * cx + cy * cy + cz * cz;
    float dcx, dcy, dcz;
    if (l2 > 1e-20f) {  // d (c/|c|) = (g - n (n.g)) / |c|
        float l = sqrtf(l2), il = 1.0f / l;
        float nx = cx * il, ny = cy * il, nz = cz * il;
        float ng = nx * gx + ny * gy + nz * gz;
        dcx = (gx - nx * ng) * il;
        dcy = (gy - ny * ng) * il;
        dcz = (gz - nz * ng) * il;
    } else {  // clamped length: n = c / 1e-10
        dcx = gx * 1e10f;
        dcy = gy * 1e10f;
        dcz = gz * 1e10f;
    }
    // c = e1 x e2:  d e1 = e2 x dc,  d e2 = dc x e1
    float d1x = e2y * dcz - e2z * dcy, d1y = e2z * dcx - e2x * dcz, d1z = e2x * dcy - e2y * dcx;
    float d2x = dcy * e1z - dcz * e1y, d2y = dcz * e1x - dcx * e1z, d2z = dcx * e1y - dcy * e1x;
    atomicAdd(&g_v[3 * i1], d1x); atomicAdd(&g_v[3 * i1 + 1], d1y); atomicAdd(&g_v[3 * i1 + 2], d1z);
    atomicAdd(&g_v[3 * i2], d2x); atomicAdd(&g_v[3 * i2 + 1], d2y); atomicAdd(&g_v[3 * i2 + 2], d2z);
    atomicAdd(&g_v[3 * i0], -(d1x + d2x)); atomicAdd(&g_v[3 * i0 + 1], -(d1y + d2y)); atomicAdd(&g_v[3 * i0 + 2], -(d1z + d2z));
}

}  // namespace

extern "C" int gs_xfm_points_fwd(const float* pts, int64_t Bp, const float* mtx, int64_t B, int64_t V, float* out, gs_stream_t stream) {
    GS_REQUIRE(Bp == 1 || Bp == B, "gs_xfm_points_fwd: points batch must be 1 or B");
    if (B <= 0 || V <= 0) return 0;
    GS_REQUIRE(pts && mtx && out, "gs_xfm_points_fwd: null pointer");
    dim3 grid((unsigned)gs::cdiv(V, 256), (unsigned)B);
    hipLaunchKernelGGL(k_xfm_fwd, grid, dim3(256), 0, (hipStream_t)stream, pts, Bp, mtx, B, V, (float4*)out);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_xfm_points_bwd(const float* g_out, int64_t Bp, const float* mtx, int64_t B, int64_t V, float* g_pts, gs_stream_t stream) {
    GS_REQUIRE(Bp == 1 || Bp == B, "gs_xfm_points_bwd: points batch must be 1 or B");
    if (B <= 0 || V <= 0) return 0;
    GS_REQUIRE(g_out && mtx && g_pts, "gs_xfm_points_bwd: null pointer");
    GS_REQUIRE(B <= 512, "gs_xfm_points_bwd: batch too large");
    hipLaunchKernelGGL(k_xfm_bwd, dim3((unsigned)gs::cdiv(V, 256)), dim3(256), (size_t)B * 16 * sizeof(float), (hipStream_t)stream,
                       (const float4*)g_out, Bp, mtx, B, V, g_pts);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t gs_rasterize_scratch_bytes(int64_t B, int64_t T, int64_t H, int64_t W) {
    return B * H * W * 8 + 16 + B * T * 8 + (GS_RAST_DEBUG ? 8 + RAST_DEBUG_CAP * 64 : 0);
}

extern "C" int gs_rasterize_fwd(const float* pos_clip, int64_t B, int64_t V, const int32_t* tri, int64_t T, int64_t H, int64_t W,
                                void* scratch, float* rast, float* rast_db, uint8_t* tri_visible, gs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GS_REQUIRE(B >= 0 && H > 0 && W > 0 && H <= 16384 && W <= 16384, "gs_rasterize_fwd: bad resolution");
    GS_REQUIRE(T < (1ll << 31) && V < (1ll << 31), "gs_rasterize_fwd: mesh too large for int32 ids");
    int64_t npix = B * H * W;
    if (npix == 0) return 0;
    GS_REQUIRE(rast && scratch, "gs_rasterize_fwd: null pointer");
    uint64_t* zbuf = (uint64_t*)scratch;
    int32_t* qcount = (int32_t*)(zbuf + npix);
    int64_t* queue = (int64_t*)(zbuf + npix + 2);
    hipLaunchKernelGGL(k_rast_clear, dim3((unsigned)std::min<int64_t>(gs::cdiv(npix, 256), 4096)), dim3(256), 0, stream, zbuf, npix, qcount,
                       GS_RAST_DEBUG ? (int32_t*)(queue + B * T) : (int32_t*)nullptr);
    if (T > 0 && V > 0) {
        GS_REQUIRE(pos_clip && tri, "gs_rasterize_fwd: null mesh pointer");
        hipLaunchKernelGGL(k_rast_small, dim3((unsigned)gs::cdiv(B * T, 256)), dim3(256), 0, stream, (const float4*)pos_clip, tri, B, V, T,
                           (int)H, (int)W, zbuf, qcount, queue);
        hipLaunchKernelGGL(k_rast_large, dim3(1024), dim3(256), 0, stream, (const float4*)pos_clip, tri, V, T, (int)H, (int)W, zbuf, qcount,
                           queue);
    }
    hipLaunchKernelGGL(k_rast_resolve, dim3((unsigned)gs::cdiv(npix, 256)), dim3(256), 0, stream, (const float4*)pos_clip, tri, B, V, (int)H,
                       (int)W, zbuf, (float4*)rast, (float4*)rast_db, tri_visible);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_rasterize_bwd(const float* pos_clip, int64_t B, int64_t V, const int32_t* tri, int64_t T, int64_t H, int64_t W,
                                const float* rast, const float* g_rast, float* g_pos, gs_stream_t stream) {
    int64_t npix = B * H * W;
    if (npix == 0 || T == 0 || V == 0) return 0;
    GS_REQUIRE(pos_clip && tri && rast && g_rast && g_pos, "gs_rasterize_bwd: null pointer");
    hipLaunchKernelGGL(k_rast_bwd, dim3((unsigned)gs::cdiv(npix, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)pos_clip, tri, B, V,
                       (int)H, (int)W, (const float4*)rast, (const float4*)g_rast, g_pos);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_interpolate_fwd(const float* attr, int64_t Ba, int64_t V, int64_t A, const float* rast, const float* rast_db,
                                  const int32_t* tri, int64_t T, int64_t B, int64_t H, int64_t W, float* out, float* out_da,
                                  gs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int64_t npix = B * H * W;
    if (npix == 0 || A == 0) return 0;
    GS_REQUIRE(Ba == 1 || Ba == B, "gs_interpolate_fwd: attribute batch must be 1 or B");
    GS_REQUIRE(rast && out, "gs_interpolate_fwd: null pointer");
    GS_REQUIRE(T == 0 || (attr && tri), "gs_interpolate_fwd: null mesh pointer");
    GS_REQUIRE(out_da == nullptr || rast_db != nullptr, "gs_interpolate_fwd: out_da needs rast_db");
    dim3 grid((unsigned)gs::cdiv(npix, 256)), block(256);
#define GS_INTERP(AS) \
    hipLaunchKernelGGL(k_interp_fwd<AS>, grid, block, 0, stream, attr, Ba, V, (int)A, (const float4*)rast, (const float4*)rast_db, tri, T, B, \
                       H * W, out, out_da)
    switch (A) {
        case 1: GS_INTERP(1); break;
        case 3: GS_INTERP(3); break;
        case 4: GS_INTERP(4); break;
        default: GS_INTERP(0); break;
    }
#undef GS_INTERP
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_interpolate_bwd(const float* attr, int64_t Ba, int64_t V, int64_t A, const float* rast, const int32_t* tri, int64_t T,
                                  int64_t B, int64_t H, int64_t W, const float* g_out, float* g_attr, float* g_rast, gs_stream_t stream) {
    int64_t npix = B * H * W;
    if (npix == 0 || A == 0) return 0;
    GS_REQUIRE(Ba == 1 || Ba == B, "gs_interpolate_bwd: attribute batch must be 1 or B");
    GS_REQUIRE(rast && g_out, "gs_interpolate_bwd: null pointer");
    GS_REQUIRE(T == 0 || (attr && tri), "gs_interpolate_bwd: null mesh pointer");
    hipLaunchKernelGGL(k_interp_bwd, dim3((unsigned)gs::cdiv(npix, 256)), dim3(256), 0, (hipStream_t)stream, attr, Ba, V, (int)A,
                       (const float4*)rast, tri, T, B, H * W, g_out, g_attr, (float4*)g_rast);
    GS_LAUNCH_CHECK();
    return 0;
}

static int interp_groups_args(InterpGroups& G, int n_groups, const int32_t* channels, const float* const* attrs, int64_t T, const char* who) {
    GS_REQUIRE(n_groups >= 1 && n_groups <= 4 && channels && attrs, "gs_interpolate_groups: 1..4 attribute tensors");
    G.n = n_groups;
    for (int k = 0; k < n_groups; ++k) {
        GS_REQUIRE(channels[k] >= 1 && channels[k] <= 64 && (attrs[k] || T == 0), "gs_interpolate_groups: bad attribute tensor");   // empty mesh: nothing is read
        G.ch[k] = channels[k];
        G.attr[k] = attrs[k];
    }
    (void)who;
    return 0;
}

extern "C" int gs_interpolate_groups_fwd(int n_groups, const int32_t* channels, const float* const* attrs, const float* rast, const int32_t* tri, int64_t T,
                                         int64_t B, int64_t H, int64_t W, float* const* outs, gs_stream_t stream) {
    const int64_t npix = B * H * W;
    if (npix == 0) return 0;
    InterpGroups G{};
    if (int rc = interp_groups_args(G, n_groups, channels, attrs, T, "gs_interpolate_groups_fwd")) return rc;
    GS_REQUIRE(rast && outs && (T == 0 || tri), "gs_interpolate_groups_fwd: null pointer");
    for (int k = 0; k < n_groups; ++k) {
        GS_REQUIRE(outs[k], "gs_interpolate_groups_fwd: null output");
        G.out[k] = outs[k];
    }
    hipLaunchKernelGGL(k_interp_groups_fwd, dim3((unsigned)gs::cdiv(npix, 256)), dim3(256), 0, (hipStream_t)stream, G, (const float4*)rast, tri, T, npix);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_interpolate_groups_bwd(int n_groups, const int32_t* channels, const float* const* attrs, const float* rast, const int32_t* tri, int64_t T,
                                         int64_t B, int64_t H, int64_t W, const float* const* g_outs, float* const* g_attrs, float* g_rast,
                                         gs_stream_t stream) {
    const int64_t npix = B * H * W;
    if (npix == 0) return 0;
    InterpGroups G{};
    if (int rc = interp_groups_args(G, n_groups, channels, attrs, T, "gs_interpolate_groups_bwd")) return rc;
    GS_REQUIRE(rast && g_outs && g_attrs && (T == 0 || tri), "gs_interpolate_groups_bwd: null pointer");
    for (int k = 0; k < n_groups; ++k) {
        G.g_out[k] = g_outs[k];        // NULL: no gradient flows into this output
        G.g_attr[k] = g_attrs[k];      // NULL: this attribute needs no gradient; otherwise ACCUMULATED (atomics)
    }
    hipLaunchKernelGGL(k_interp_groups_bwd, dim3((unsigned)gs::cdiv(npix, 256)), dim3(256), 0, (hipStream_t)stream, G, (const float4*)rast, tri, T, npix,
                       (float4*)g_rast);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_face_normal_fwd(const float* v_pos, int64_t V, const int32_t* tri, int64_t T, const float* rast, int64_t B, int64_t H, int64_t W,
                                  float* out, gs_stream_t stream) {
    int64_t npix = B * H * W;
    if (npix == 0) return 0;
    GS_REQUIRE(rast && out && (T == 0 || (v_pos && tri)), "gs_face_normal_fwd: null pointer");
    hipLaunchKernelGGL(k_face_normal_fwd, dim3((unsigned)gs::cdiv(npix, 256)), dim3(256), 0, (hipStream_t)stream, v_pos, tri, T, (const float4*)rast, npix,
                       out);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_face_normal_bwd(const float* v_pos, int64_t V, const int32_t* tri, int64_t T, const float* rast, int64_t B, int64_t H, int64_t W,
                                  const float* g_out, float* g_v_pos, gs_stream_t stream) {
    int64_t npix = B * H * W;
    if (npix == 0 || T == 0) return 0;
    GS_REQUIRE(rast && g_out && g_v_pos && v_pos && tri, "gs_face_normal_bwd: null pointer");
    hipLaunchKernelGGL(k_face_normal_bwd, dim3((unsigned)gs::cdiv(npix, 256)), dim3(256), 0, (hipStream_t)stream, v_pos, tri, T, (const float4*)rast, npix,
                       g_out, g_v_pos);
    GS_LAUNCH_CHECK();
    return 0;
}

// Silhouette antialiasing on gfx950 (fwd + bwd) and the triangle-adjacency table it needs.
//
// Replaces the third-party nvdiffrast call `dr.antialias(color, rast, pos, tri)` that the reference
// issues once per output buffer (render/render.py:352-359, :417-433; ~12 calls per iteration, each
// redoing the silhouette analysis and rebuilding a topology hash).  Semantics restated from the
// public description of nvdiffrast's antialias op (SURVEY.md 8c [3P-memory]):
//   for each horizontally / vertically adjacent pixel pair with different triangle ids, take the
//   nearer triangle; if the edge through which the segment between the two pixel centres leaves that
//   triangle is a silhouette edge (it has no neighbour triangle, or the neighbour lies on the same
//   screen-space side) and is steeper than 45 degrees w.r.t. the segment, blend the two pixels by the
//   crossing fraction: alpha = ds (0.5 - dc);  out[alpha > 0 ? p0 : p1] += alpha (c1 - c0).
//   Gradients reach the colours and the clip-space positions of the edge's two vertices.
//
// MI355X design (differs from the per-call structure above on purpose):
//   1. gs_tri_adjacency  : opposite vertex across each triangle edge, by ONE radix sort of the 3T
//                          (min,max) edge keys (deterministic; replaces the per-call hash build).
//   2. gs_aa_analyze     : ONE pass per render producing a dense alpha[B,H,W,2] (right / down pair).
//   3. gs_aa_apply_fwd   : any number of channels (all output buffers concatenated) in one launch, lane per
//                          (pixel, channel), gather form -> no atomics, bit-reproducible, fully coalesced.
//   4. gs_aa_apply_bwd   : gather form for colour grads; d loss / d alpha by sparse atomics (silhouette pixels only).
//   5. gs_aa_analyze_bwd : sparse: only pairs with alpha != 0 touch vertex gradients (atomics).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>

#include "../../include/gshell_hip.h"
#include "common.hpp"

namespace {

__global__ void __launch_bounds__(256) k_edge_keys(const int32_t* __restrict__ tri, int64_t T, uint64_t* __restrict__ keys,
                                                   uint32_t* __restrict__ vals) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * T) return;
    int64_t t = i / 3;
    int e = (int)(i - 3 * t);  // edge e is opposite vertex e
    uint32_t a = (uint32_t)tri[3 * t + (e + 1) % 3], b = (uint32_t)tri[3 * t + (e + 2) % 3];
    keys[i] = ((uint64_t)min(a, b) << 32) | max(a, b);
    vals[i] = (uint32_t)i;
}

// groups of exactly two equal keys are manifold edges: each side learns the other's opposite vertex
__global__ void __launch_bounds__(256) k_edge_pair(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, int64_t n,
                                                   const int32_t* __restrict__ tri, int32_t* __restrict__ opp) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = keys[i];
    bool prev = i > 0 && keys[i - 1] == k, next = i + 1 < n && keys[i + 1] == k;
    int32_t o = -1;
    if (next && !prev && !(i + 2 < n && keys[i + 2] == k)) o = tri[vals[i + 1]];
    if (prev && !next && !(i >= 2 && keys[i - 2] == k)) o = tri[vals[i - 1]];
    opp[vals[i]] = o;
}

__device__ __forceinline__ bool same_sign(float a, float b) { return ((__float_as_uint(a) ^ __float_as_uint(b)) & 0x80000000u) == 0u; }

struct PairGeom {
    int tri;      // chosen triangle (-1: nothing to do)
    int ei;       // exit edge (opposite vertex ei), valid if alpha != 0
    float ds;     // +1: reference pixel is p0, -1: reference pixel is p1
    float dc;     // unclamped crossing distance from the reference pixel centre
    float alpha;  // 0 if no blend
    float ax, ay, bx, by;  // exit edge end points in the pair frame (pair axis = x)
    int32_t ia, ib;        // their vertex ids
    float qx, qy;          // reference pixel
};

// Shared by analysis fwd and bwd so both take identical decisions.
__device__ __forceinline__ PairGeom pair_analyze(const float4* __restrict__ pv, const int32_t* __restrict__ tri, const int32_t* __restrict__ opp,
                                                 float4 r0, float4 r1, int px, int py, int d, int H, int W) {
    PairGeom g;
    g.tri = -1;
    g.alpha = 0.f;
    int t0 = (int)r0.w - 1, t1 = (int)r1.w - 1;
    if (t0 == t1) return g;
    int t = t0 >= 0 ? t0 : t1;
    if (t0 >= 0 && t1 >= 0) t = (r0.z < r1.z) ? t0 : t1;
    g.tri = t;
    g.ds = (t == t0) ? 1.0f : -1.0f;
    int qx = px, qy = py;
    if (t == t1) {
        qx += 1 - d;
        qy += d;
    }
    g.qx = (float)qx;
    g.qy = (float)qy;
    int32_t vi[3] = {tri[3 * (int64_t)t], tri[3 * (int64_t)t + 1], tri[3 * (int64_t)t + 2]};
    float hx = 0.5f * (float)W, hy = 0.5f * (float)H;
    float fx = (float)qx + 0.5f - hx, fy = (float)qy + 0.5f - hy;
    float Px[3], Py[3], Ox[3], Oy[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float4 p = pv[vi[i]];
        float iw = 1.0f / p.w;
        Px[i] = p.x * iw * hx - fx;
        Py[i] = p.y * iw * hy - fy;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int32_t o = opp[3 * (int64_t)t + i];
        if (o < 0) {
            Ox[i] = Px[i];
            Oy[i] = Py[i];
        } else {
            float4 p = pv[o];
            float iw = 1.0f / p.w;
            Ox[i] = p.x * iw * hx - fx;
            Oy[i] = p.y * iw * hy - fy;
        }
    }
    float bb = (Px[1] - Px[0]) * (Py[2] - Py[0]) - (Px[2] - Px[0]) * (Py[1] - Py[0]);
    bool sil[3];
    sil[0] = same_sign((Px[1] - Ox[0]) * (Py[2] - Oy[0]) - (Px[2] - Ox[0]) * (Py[1] - Oy[0]), bb);
    sil[1] = same_sign((Px[2] - Ox[1]) * (Py[0] - Oy[1]) - (Px[0] - Ox[1]) * (Py[2] - Oy[1]), bb);
    sil[2] = same_sign((Px[0] - Ox[2]) * (Py[1] - Oy[2]) - (Px[1] - Ox[2]) * (Py[0] - Oy[2]), bb);
    if (!(sil[0] || sil[1] || sil[2])) return g;
    if (d) {  // vertical pair: swap axes so that the pair axis is x
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float tmp = Px[i];
            Px[i] = Py[i];
            Py[i] = tmp;
        }
    }
    // exit edge = crossing edge with the largest crossing distance along the pair axis
    int best = -1;
    float best_dc = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int a = (i + 1) % 3, b = (i + 2) % 3;
        if (same_sign(Py[a], Py[b])) continue;  // does not cross the centre line
        float num = g.ds * (Px[a] * Py[b] - Px[b] * Py[a]);
        float den = Py[b] - Py[a];
        float dc = num / den;
        if (best < 0 || dc > best_dc) {
            best = i;
            best_dc = dc;
        }
    }
    if (best < 0 || !sil[best]) return g;
    int a = (best + 1) % 3, b = (best + 2) % 3;
    if (!(fabsf(Py[b] - Py[a]) >= fabsf(Px[b] - Px[a]))) return g;
    const float eps = 0.0625f;
    if (!(best_dc > -eps && best_dc < 1.0f + eps)) return g;
    g.ei = best;
    g.dc = best_dc;
    g.alpha = g.ds * (0.5f - fminf(fmaxf(best_dc, 0.0f), 1.0f));
    g.ax = Px[a];
    g.ay = Py[a];
    g.bx = Px[b];
    g.by = Py[b];
    g.ia = vi[a];
    g.ib = vi[b];
    return g;
}

__global__ void __launch_bounds__(256) k_aa_analyze(const float4* __restrict__ pos, const int32_t* __restrict__ tri,
                                                    const int32_t* __restrict__ opp, int64_t B, int64_t V, int H, int W,
                                                    const float4* __restrict__ rast, float2* __restrict__ alpha) {
    int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= B * (int64_t)H * W) return;
    int64_t b = pix / ((int64_t)H * W);
    int rem = (int)(pix - b * (int64_t)H * W);
    int py = rem / W, px = rem - py * W;
    float4 r0 = rast[pix];
    float2 out = make_float2(0.f, 0.f);
    const float4* pv = pos + b * V;
    if (px + 1 < W) {
        float4 r1 = rast[pix + 1];
        if (r0.w != r1.w) out.x = pair_analyze(pv, tri, opp, r0, r1, px, py, 0, H, W).alpha;
    }
    if (py + 1 < H) {
        float4 r1 = rast[pix + W];
        if (r0.w != r1.w) out.y = pair_analyze(pv, tri, opp, r0, r1, px, py, 1, H, W).alpha;
    }
    alpha[pix] = out;
}

// out[p] = c[p] + sum over the (up to) four pairs that target p, in the fixed order right, left, down, up.
// One lane per (pixel, channel) with the channel index fastest: a wave touches 64 consecutive floats of every tensor it
// reads or writes (the first version walked the C channels of one pixel per lane -- 180-byte lane stride at C = 45 -- and
// rocprof showed 7x the algorithmic HBM traffic).
__global__ void __launch_bounds__(256) k_aa_apply_fwd(const float* __restrict__ color, const float2* __restrict__ alpha, int64_t B, int H,
                                                      int W, int C, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n = B * (int64_t)H * W * C;
    if (i >= n) return;
    int64_t pix = i / C;
    int rem = (int)(pix % ((int64_t)H * W));
    int py = rem / W, px = rem - py * W;
    float2 a = alpha[pix];
    float ar = a.x > 0.f ? a.x : 0.f;                                  // pair (p, right), target p
    float ad = a.y > 0.f ? a.y : 0.f;                                  // pair (p, down),  target p
    float al = (px > 0) ? alpha[pix - 1].x : 0.f;                      // pair (left, p),  target p iff alpha < 0
    float au = (py > 0) ? alpha[pix - W].y : 0.f;                      // pair (up, p)
    al = al < 0.f ? al : 0.f;
    au = au < 0.f ? au : 0.f;
    float v = color[i];
    float acc = v;
    if (ar != 0.f) acc += ar * (color[i + C] - v);
    if (al != 0.f) acc += al * (v - color[i - C]);
    if (ad != 0.f) acc += ad * (color[i + (int64_t)W * C] - v);
    if (au != 0.f) acc += au * (v - color[i - (int64_t)W * C]);
    out[i] = acc;
}

// colour gradient (gather form, same lane mapping) and d loss / d alpha of the two pairs owned by this pixel
// (g_alpha must be zero-filled: channels of a silhouette pixel add their share with one atomic each -- sparse).
__global__ void __launch_bounds__(256) k_aa_apply_bwd(const float* __restrict__ color, const float2* __restrict__ alpha, int64_t B, int H,
                                                      int W, int C, const float* __restrict__ g_out, float* __restrict__ g_color,
                                                      float* __restrict__ g_alpha) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n = B * (int64_t)H * W * C;
    if (i >= n) return;
    int64_t pix = i / C;
    int rem = (int)(pix % ((int64_t)H * W));
    int py = rem / W, px = rem - py * W;
    float2 a = alpha[pix];
    float al = (px > 0) ? alpha[pix - 1].x : 0.f;
    float au = (py > 0) ? alpha[pix - W].y : 0.f;
    const int64_t dn = (int64_t)W * C;
    float g = g_out[i];
    float acc = g;
    if (a.x != 0.f) {   // pair (p, right): tau = a.x > 0 ? p : right ; c0 = c[p], c1 = c[right]
        float gt = a.x > 0.f ? g : g_out[i + C];
        acc -= a.x * gt;
        float d = gt * (color[i + C] - color[i]);
        if (g_alpha && d != 0.f) atomicAdd(&g_alpha[2 * pix], d);
    }
    if (a.y != 0.f) {
        float gt = a.y > 0.f ? g : g_out[i + dn];
        acc -= a.y * gt;
        float d = gt * (color[i + dn] - color[i]);
        if (g_alpha && d != 0.f) atomicAdd(&g_alpha[2 * pix + 1], d);
    }
    if (al != 0.f) acc += al * (al > 0.f ? g_out[i - C] : g);       // pair (left, p): tau = al > 0 ? left : p ; here p is c1
    if (au != 0.f) acc += au * (au > 0.f ? g_out[i - dn] : g);
    if (g_color) g_color[i] = acc;
}

// ---- in-place apply: only silhouette pixels are touched ---------------------------------------------------------------------------
// k_aa_apply_fwd / bwd stream the whole [B,H,W,C] frame through (2 x 190 MB each way at 4 x 512^2 x 45) although ~1 % of its pixels change.
// When the caller owns the frame (render_mesh: the composite is consumed by the antialias only) it is updated in place, in two launches so
// that every blend reads UNMODIFIED colours: phase 1 writes the new value of every target pixel to a scratch tensor of the frame's shape (only
// those entries are ever touched), phase 2 swaps them in -- the scratch then holds the ORIGINAL colours of the modified pixels, which is what
// the backward pass needs.  Same expressions in the same order as the streaming kernels: bit-identical values and gradients.
struct AaWeights { float ar, al, ad, au; };
__device__ __forceinline__ bool aa_target(const float2* __restrict__ alpha, int64_t pix, int px, int py, int W, AaWeights& w) {
    const float2 a = alpha[pix];
    w.ar = a.x > 0.f ? a.x : 0.f;
    w.ad = a.y > 0.f ? a.y : 0.f;
    w.al = (px > 0) ? alpha[pix - 1].x : 0.f;
    w.au = (py > 0) ? alpha[pix - W].y : 0.f;
    w.al = w.al < 0.f ? w.al : 0.f;
    w.au = w.au < 0.f ? w.au : 0.f;
    return w.ar != 0.f || w.al != 0.f || w.ad != 0.f || w.au != 0.f;
}

template <int PHASE>
__global__ void __launch_bounds__(256) k_aa_inplace_fwd(float* __restrict__ color, const float2* __restrict__ alpha, int64_t B, int H, int W, int C,
                                                        float* __restrict__ scratch) {
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= B * (int64_t)H * W) return;
    const int rem = (int)(pix % ((int64_t)H * W));
    const int py = rem / W, px = rem - py * W;
    AaWeights w;
    if (!aa_target(alpha, pix, px, py, W, w)) return;
    const int64_t i0 = pix * C, dn = (int64_t)W * C;
    for (int c = 0; c < C; ++c) {
        const int64_t i = i0 + c;
        if (PHASE == 1) {
            const float v = color[i];
            float acc = v;
            if (w.ar != 0.f) acc += w.ar * (color[i + C] - v);
            if (w.al != 0.f) acc += w.al * (v - color[i - C]);
            if (w.ad != 0.f) acc += w.ad * (color[i + dn] - v);
            if (w.au != 0.f) acc += w.au * (v - color[i - dn]);
            scratch[i] = acc;
        } else {
            const float t = color[i];
            color[i] = scratch[i];
            scratch[i] = t;
        }
    }
}

// original colour of (pixel q, channel): the saved one where the forward pass modified q
__device__ __forceinline__ bool aa_is_target(const float2* __restrict__ alpha, int64_t q, int qx, int qy, int W) {
    AaWeights w;
    return aa_target(alpha, q, qx, qy, W, w);
}

template <int PHASE>
__global__ void __launch_bounds__(256) k_aa_inplace_bwd(const float* __restrict__ out_color, const float* __restrict__ orig, const float2* __restrict__ alpha,
                                                        int64_t B, int H, int W, int C, float* __restrict__ g, float* __restrict__ g_scratch,
                                                        float* __restrict__ g_alpha) {
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= B * (int64_t)H * W) return;
    const int rem = (int)(pix % ((int64_t)H * W));
    const int py = rem / W, px = rem - py * W;
    const float2 a = alpha[pix];
    const float al = (px > 0) ? alpha[pix - 1].x : 0.f;
    const float au = (py > 0) ? alpha[pix - W].y : 0.f;
    if (a.x == 0.f && a.y == 0.f && al == 0.f && au == 0.f) return;         // the gradient of this pixel passes through unchanged
    const int64_t i0 = pix * C, dn = (int64_t)W * C;
    if (PHASE == 2) {
        for (int c = 0; c < C; ++c) g[i0 + c] = g_scratch[i0 + c];
        return;
    }
    const bool t0 = aa_is_target(alpha, pix, px, py, W);
    const bool tr = a.x != 0.f && aa_is_target(alpha, pix + 1, px + 1, py, W);
    const bool td = a.y != 0.f && aa_is_target(alpha, pix + W, px, py + 1, W);
    float dax = 0.f, day = 0.f;
    for (int c = 0; c < C; ++c) {
        const int64_t i = i0 + c;
        const float gv = g[i];
        float acc = gv;
        const float c0 = t0 ? orig[i] : out_color[i];
        if (a.x != 0.f) {   // pair (p, right): tau = a.x > 0 ? p : right ; c0 = c[p], c1 = c[right]
            const float gt = a.x > 0.f ? gv : g[i + C];
            acc -= a.x * gt;
            const float c1 = tr ? orig[i + C] : out_color[i + C];
            const float d = gt * (c1 - c0);
            if (d != 0.f) dax += d;
        }
        if (a.y != 0.f) {
            const float gt = a.y > 0.f ? gv : g[i + dn];
            acc -= a.y * gt;
            const float c1 = td ? orig[i + dn] : out_color[i + dn];
            const float d = gt * (c1 - c0);
            if (d != 0.f) day += d;
        }
        if (al != 0.f) acc += al * (al > 0.f ? g[i - C] : gv);
        if (au != 0.f) acc += au * (au > 0.f ? g[i - dn] : gv);
        g_scratch[i] = acc;
    }
    if (g_alpha) {          // this pixel owns its two pairs: plain stores (the streaming kernel adds the channels with atomics, in any order)
        g_alpha[2 * pix] = dax;
        g_alpha[2 * pix + 1] = day;
    }
}

__device__ __forceinline__ void pair_bwd(const float4* __restrict__ pv, float* __restrict__ gp, const PairGeom& g, float g_alpha, int d,
                                         int H, int W) {
    if (g.alpha == 0.f || g_alpha == 0.f) return;
    if (!(g.dc > 0.0f && g.dc < 1.0f)) return;  // clamped: no gradient
    float g_dc = -g.ds * g_alpha;
    float den = g.by - g.ay;
    float num = g.dc * den;
    float g_num = g_dc / den, g_den = -g_dc * num / (den * den);
    // num = ds (ax by - bx ay), den = by - ay   (pair frame)
    float gax = g_num * g.ds * g.by, gay = -g_num * g.ds * g.bx - g_den;
    float gbx = -g_num * g.ds * g.ay, gby = g_num * g.ds * g.ax + g_den;
    if (d) {  // undo the axis swap
        float t = gax; gax = gay; gay = t;
        t = gbx; gbx = gby; gby = t;
    }
    float hx = 0.5f * (float)W, hy = 0.5f * (float)H;
    // P = (x/w hx - fx, y/w hy - fy)
    float4 pa = pv[g.ia], pb = pv[g.ib];
    float iwa = 1.0f / pa.w, iwb = 1.0f / pb.w;
    atomicAdd(&gp[4 * (int64_t)g.ia + 0], gax * hx * iwa);
    atomicAdd(&gp[4 * (int64_t)g.ia + 1], gay * hy * iwa);
    atomicAdd(&gp[4 * (int64_t)g.ia + 3], -(gax * hx * pa.x + gay * hy * pa.y) * iwa * iwa);
    atomicAdd(&gp[4 * (int64_t)g.ib + 0], gbx * hx * iwb);
    atomicAdd(&gp[4 * (int64_t)g.ib + 1], gby * hy * iwb);
    atomicAdd(&gp[4 * (int64_t)g.ib + 3], -(gbx * hx * pb.x + gby * hy * pb.y) * iwb * iwb);
}

__global__ void __launch_bounds__(256) k_aa_analyze_bwd(const float4* __restrict__ pos, const int32_t* __restrict__ tri,
                                                        const int32_t* __restrict__ opp, int64_t B, int64_t V, int H, int W,
                                                        const float4* __restrict__ rast, const float2* __restrict__ alpha,
                                                        const float2* __restrict__ g_alpha, float* __restrict__ g_pos) {
    int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= B * (int64_t)H * W) return;
    float2 a = alpha[pix];
    if (a.x == 0.f && a.y == 0.f) return;
    float2 ga = g_alpha[pix];
    int64_t b = pix / ((int64_t)H * W);
    int rem = (int)(pix - b * (int64_t)H * W);
    int py = rem / W, px = rem - py * W;
    const float4* pv = pos + b * V;
    float* gp = g_pos + b * V * 4;
    float4 r0 = rast[pix];
    if (a.x != 0.f && ga.x != 0.f) pair_bwd(pv, gp, pair_analyze(pv, tri, opp, r0, rast[pix + 1], px, py, 0, H, W), ga.x, 0, H, W);
    if (a.y != 0.f && ga.y != 0.f) pair_bwd(pv, gp, pair_analyze(pv, tri, opp, r0, rast[pix + W], px, py, 1, H, W), ga.y, 1, H, W);
}

}  // namespace

static size_t adjacency_sort_tmp(int64_t n) {
    size_t tmp = 0;
    if (n > 0)
        (void)rocprim::radix_sort_pairs(nullptr, tmp, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n,
                                        0, 64, (hipStream_t)0);
    return (tmp + 255) & ~(size_t)255;
}

extern "C" int64_t gs_tri_adjacency_scratch_bytes(int64_t T) {
    int64_t n = 3 * std::max<int64_t>(T, 1);
    return (int64_t)adjacency_sort_tmp(n) + 2 * n * 8 + 2 * n * 4 + 1024;
}

extern "C" int gs_tri_adjacency(const int32_t* tri, int64_t T, int64_t V, void* scratch, int32_t* opp, gs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (T <= 0) return 0;
    GS_REQUIRE(tri && scratch && opp, "gs_tri_adjacency: null pointer");
    GS_REQUIRE(3 * T < (1ll << 32), "gs_tri_adjacency: too many triangles");
    int64_t n = 3 * T;
    size_t tmp_bytes = adjacency_sort_tmp(n);
    char* p = (char*)scratch;
    void* tmp = p;
    p += tmp_bytes;
    uint64_t* keys = (uint64_t*)p;
    p += n * 8;
    uint64_t* keys2 = (uint64_t*)p;
    p += n * 8;
    uint32_t* vals = (uint32_t*)p;
    p += n * 4;
    uint32_t* vals2 = (uint32_t*)p;
    hipLaunchKernelGGL(k_edge_keys, dim3((unsigned)gs::cdiv(n, 256)), dim3(256), 0, stream, tri, T, keys, vals);
    int bits = 1;
    while ((1ll << bits) < std::max<int64_t>(V, 2)) ++bits;
    GS_HIP_CHECK(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, vals, vals2, (size_t)n, 0, 32 + bits, stream));
    hipLaunchKernelGGL(k_edge_pair, dim3((unsigned)gs::cdiv(n, 256)), dim3(256), 0, stream, keys2, vals2, n, tri, opp);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_aa_analyze(const float* pos_clip, int64_t B, int64_t V, const int32_t* tri, int64_t T, const int32_t* opp, const float* rast,
                             int64_t H, int64_t W, float* alpha, gs_stream_t stream) {
    int64_t npix = B * H * W;
    if (npix == 0) return 0;
    GS_REQUIRE(rast && alpha, "gs_aa_analyze: null pointer");
    if (T == 0) {
        GS_HIP_CHECK(hipMemsetAsync(alpha, 0, (size_t)npix * 8, (hipStream_t)stream));
        return 0;
    }
    GS_REQUIRE(pos_clip && tri && opp, "gs_aa_analyze: null mesh pointer");
    hipLaunchKernelGGL(k_aa_analyze, dim3((unsigned)gs::cdiv(npix, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)pos_clip, tri, opp,
                       B, V, (int)H, (int)W, (const float4*)rast, (float2*)alpha);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_aa_apply_fwd(const float* color, const float* alpha, int64_t B, int64_t H, int64_t W, int64_t C, float* out,
                               gs_stream_t stream) {
    int64_t npix = B * H * W;
    if (npix == 0 || C == 0) return 0;
    GS_REQUIRE(color && alpha && out && color != out, "gs_aa_apply_fwd: null or aliased pointer");
    hipLaunchKernelGGL(k_aa_apply_fwd, dim3((unsigned)gs::cdiv(npix * C, 256)), dim3(256), 0, (hipStream_t)stream, color, (const float2*)alpha, B,
                       (int)H, (int)W, (int)C, out);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_aa_apply_bwd(const float* color, const float* alpha, int64_t B, int64_t H, int64_t W, int64_t C, const float* g_out,
                               float* g_color, float* g_alpha, gs_stream_t stream) {
    int64_t npix = B * H * W;
    if (npix == 0 || C == 0) return 0;
    GS_REQUIRE(color && alpha && g_out, "gs_aa_apply_bwd: null pointer");
    if (g_alpha) GS_HIP_CHECK(hipMemsetAsync(g_alpha, 0, (size_t)npix * 8, (hipStream_t)stream));
    hipLaunchKernelGGL(k_aa_apply_bwd, dim3((unsigned)gs::cdiv(npix * C, 256)), dim3(256), 0, (hipStream_t)stream, color, (const float2*)alpha, B,
                       (int)H, (int)W, (int)C, g_out, g_color, g_alpha);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_aa_apply_fwd_inplace(float* color, const float* alpha, int64_t B, int64_t H, int64_t W, int64_t C, float* scratch, gs_stream_t stream) {
    int64_t npix = B * H * W;
    if (npix == 0 || C == 0) return 0;
    GS_REQUIRE(color && alpha && scratch && color != scratch, "gs_aa_apply_fwd_inplace: null or aliased pointer");
    const dim3 grid((unsigned)gs::cdiv(npix, 256)), block(256);
    hipLaunchKernelGGL(k_aa_inplace_fwd<1>, grid, block, 0, (hipStream_t)stream, color, (const float2*)alpha, B, (int)H, (int)W, (int)C, scratch);
    hipLaunchKernelGGL(k_aa_inplace_fwd<2>, grid, block, 0, (hipStream_t)stream, color, (const float2*)alpha, B, (int)H, (int)W, (int)C, scratch);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_aa_apply_bwd_inplace(const float* out_color, const float* saved, const float* alpha, int64_t B, int64_t H, int64_t W, int64_t C,
                                       float* g, float* g_scratch, float* g_alpha, gs_stream_t stream) {
    int64_t npix = B * H * W;
    if (npix == 0 || C == 0) return 0;
    GS_REQUIRE(out_color && saved && alpha && g && g_scratch && g != g_scratch, "gs_aa_apply_bwd_inplace: null or aliased pointer");
    if (g_alpha) GS_HIP_CHECK(hipMemsetAsync(g_alpha, 0, (size_t)npix * 8, (hipStream_t)stream));
    const dim3 grid((unsigned)gs::cdiv(npix, 256)), block(256);
    hipLaunchKernelGGL(k_aa_inplace_bwd<1>, grid, block, 0, (hipStream_t)stream, out_color, saved, (const float2*)alpha, B, (int)H, (int)W, (int)C, g,
                       g_scratch, g_alpha);
    hipLaunchKernelGGL(k_aa_inplace_bwd<2>, grid, block, 0, (hipStream_t)stream, out_color, saved, (const float2*)alpha, B, (int)H, (int)W, (int)C, g,
                       g_scratch, (float*)nullptr);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_aa_analyze_bwd(const float* pos_clip, int64_t B, int64_t V, const int32_t* tri, int64_t T, const int32_t* opp,
                                 const float* rast, int64_t H, int64_t W, const float* alpha, const float* g_alpha, float* g_pos,
                                 gs_stream_t stream) {
    int64_t npix = B * H * W;
    if (npix == 0 || T == 0) return 0;
    GS_REQUIRE(pos_clip && tri && opp && rast && alpha && g_alpha && g_pos, "gs_aa_analyze_bwd: null pointer");
    hipLaunchKernelGGL(k_aa_analyze_bwd, dim3((unsigned)gs::cdiv(npix, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)pos_clip, tri,
                       opp, B, V, (int)H, (int)W, (const float4*)rast, (const float2*)alpha, (const float2*)g_alpha, g_pos);
    GS_LAUNCH_CHECK();
    return 0;
}

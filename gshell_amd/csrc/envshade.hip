// Monte-Carlo environment-light shading with shadow rays on gfx950 (fwd + bwd), and the bilateral denoiser.
//
// Replaces ou.optix_env_shade (render/optixutils/ops.py:81-108, :141-143; OptiX raygen program
// render/optixutils/c_src/envsampling/kernel.cu:463-541, BSDF math c_src/bsdf.h:21-275, helpers
// c_src/math_utils.h:80-163) and ou.bilateral_denoiser (ops.py:110-123, :145-147; c_src/denoising.cu:14-130).
//
// The reference runs one OptiX thread per pixel that loops over n^2 stratified (light sample, BSDF sample)
// pairs and traces one hardware any-hit ray per sample.  On CDNA4 the loop is turned sideways:
//   * a GROUP of G = min(64, pow2 <= n^2) lanes shades one pixel; lane j takes samples j, j+G, ...
//     With the headline n = 8 a whole wave64 shades one pixel, 2 rays per lane, and the per-pixel
//     sums are wave reductions (DPP shuffles) instead of a 128-iteration serial loop.
//   * the per-pixel PCG stream of the reference is reproduced exactly by LCG jump-ahead
//     (sample i consumes draws 2+5i .. 6+5i), so results do not depend on the lane mapping.
//   * the work is split "wavefront" style into three launches so that the ray traversal runs in a lean kernel:
//       k_shade_samples<false>  sampling + BSDF/light/MIS evaluation -> ray directions + unshadowed contributions
//       k_shade_trace           shadow rays through the implicit 4-ary BVH of bvh.hpp: every wave owns 1024 rays and
//                               refills idle lanes from an LDS stage; 1 visibility bit per ray
//       k_shade_accumulate      per-pixel sum of V * contribution
//     (a single fused kernel needed 200-244 VGPRs = 2 waves/SIMD and ran the traversal latency-starved).
//   * visibility is CACHED: 1 bit per ray (2.5 MB for 20 M rays).  The backward pass re-runs the identical sampling
//     (same PCG stream) but reads V from the cache instead of re-tracing -- exactly the values the reference's
//     second trace would produce, since it uses the same seed (ops.py:99-104) -- so backward costs no rays at all.
//     Per-pixel gradients are group-reduced and written once; the light gradient is one float atomicAdd per
//     sample and channel into the [Hl,Wl,3] probe (as kernel.cu:203-211 does).
//   * only COVERED pixels are launched (compact pixel list from the caller).
// This stage is ALU / latency bound (ray traversal), not HBM bound: bytes per pixel are ~100.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "../../include/gshell_hip.h"
#include "bvh.hpp"
#include "common.hpp"

namespace {

constexpr float PI_F = 3.14159265358979323846f;
constexpr float SPECULAR_EPSILON = 1e-4f;
constexpr float MIN_ROUGHNESS = 0.08f;

struct v3 {
    float x, y, z;
};
__device__ __forceinline__ v3 V3(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ v3 V3(float a) { return {a, a, a}; }
__device__ __forceinline__ v3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ v3 operator-(v3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ v3 operator*(v3 a, v3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ v3 operator*(float s, v3 a) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ v3 operator/(v3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ v3& operator+=(v3& a, v3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
__device__ __forceinline__ v3& operator-=(v3& a, v3 b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
__device__ __forceinline__ float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float sum(v3 a) { return a.x + a.y + a.z; }
__device__ __forceinline__ v3 cross(v3 a, v3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float luminance(v3 c) { return dot(c, V3(0.2126f, 0.7152f, 0.0722f)); }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }
__device__ __forceinline__ v3 safe_normalize(v3 v) {
    float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return l > 0.0f ? v / l : V3(0.f);
}
__device__ __forceinline__ void bwd_dot(v3 a, v3 b, v3& da, v3& db, float d) {
    da += b * d;
    db += a * d;
}
__device__ __forceinline__ void bwd_safe_normalize(v3 v, v3& dv, v3 d) {
    float l2 = v.x * v.x + v.y * v.y + v.z * v.z;
    if (l2 > 0.0f) {
        float fac = 1.0f / (l2 * sqrtf(l2));
        dv.x += (d.x * (v.y * v.y + v.z * v.z) - d.y * (v.x * v.y) - d.z * (v.x * v.z)) * fac;
        dv.y += (d.y * (v.x * v.x + v.z * v.z) - d.x * (v.y * v.x) - d.z * (v.y * v.z)) * fac;
        dv.z += (d.z * (v.x * v.x + v.y * v.y) - d.x * (v.z * v.x) - d.y * (v.z * v.y)) * fac;
    }
}
// Pixar orthonormal basis (math_utils.h:155-162)
__device__ __forceinline__ void onb(v3 n, v3& b1, v3& b2) {
    float sign = copysignf(1.0f, n.z);
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    b1 = V3(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    b2 = V3(b, sign + n.y * n.y * a, -n.y);
}
__device__ __forceinline__ v3 tolocal(v3 a, v3 u, v3 v, v3 w) { return V3(dot(a, u), dot(a, v), dot(a, w)); }
__device__ __forceinline__ v3 toworld(v3 a, v3 u, v3 v, v3 w) { return u * a.x + v * a.y + w * a.z; }

// ---- PCG hash / LCG stream (kernel.cu:30-45) ------------------------------------------------------
__device__ __forceinline__ uint32_t pcg_out(uint32_t s) {
    uint32_t word = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (word >> 22u) ^ word;
}
constexpr uint32_t LCG_A = 747796405u, LCG_C = 2891336453u;
__device__ __forceinline__ uint32_t lcg_next(uint32_t s) { return s * LCG_A + LCG_C; }
__device__ __forceinline__ uint32_t lcg_jump(uint32_t s, uint32_t k) {  // state after k steps
    uint32_t a = LCG_A, c = LCG_C, A = 1u, C = 0u;
    while (k) {
        if (k & 1u) {
            A *= a;
            C = C * a + c;
        }
        c = (a + 1u) * c;
        a *= a;
        k >>= 1;
    }
    return A * s + C;
}
__device__ __forceinline__ float u01(uint32_t s) { return (float)(pcg_out(s) & 0xFFFFFFu) / (float)0x1000000; }

// ---- BSDF (bsdf.h) ----------------------------------------------------------------------------------
__device__ __forceinline__ float fwd_lambert(v3 n, v3 wi) { return fmaxf(dot(n, wi) / PI_F, 0.0f); }
__device__ __forceinline__ v3 fwd_fresnel(v3 f0, v3 f90, float c) {
    float cc = clampf(c, SPECULAR_EPSILON, 1.0f - SPECULAR_EPSILON);
    float scale = powf(1.0f - cc, 5.0f);
    return f0 * (1.0f - scale) + f90 * scale;
}
__device__ __forceinline__ float fwd_ndf(float a2, float c) {
    float cc = clampf(c, SPECULAR_EPSILON, 1.0f - SPECULAR_EPSILON);
    float d = (cc * a2 - cc) * cc + 1.0f;
    return a2 / (d * d * PI_F);
}
__device__ __forceinline__ float fwd_lambda(float a2, float c) {
    float cc = clampf(c, SPECULAR_EPSILON, 1.0f - SPECULAR_EPSILON);
    float c2 = cc * cc;
    float t2 = (1.0f - c2) / c2;
    return 0.5f * (sqrtf(1.0f + a2 * t2) - 1.0f);
}
__device__ __forceinline__ void bwd_lambda(float a2, float c, float& da2, float& dc, float d) {
    float cc = clampf(c, SPECULAR_EPSILON, 1.0f - SPECULAR_EPSILON);
    float c2 = cc * cc;
    float t2 = (1.0f - c2) / c2;
    da2 += d * (0.25f * t2) / sqrtf(a2 * t2 + 1.0f);
    if (c > SPECULAR_EPSILON && c < 1.0f - SPECULAR_EPSILON) dc += d * -(0.5f * a2) / (cc * cc * cc * sqrtf(a2 / c2 - a2 + 1.0f));
}
__device__ __forceinline__ float fwd_smith(float a2, float ci, float co) { return 1.0f / (1.0f + fwd_lambda(a2, ci) + fwd_lambda(a2, co)); }

__device__ __forceinline__ v3 fwd_pbr_specular(v3 col, v3 nrm, v3 wo, v3 wi, float alpha, float min_rough) {
    float al = clampf(alpha, min_rough * min_rough, 1.0f);
    float a2 = al * al;
    v3 h = safe_normalize(wo + wi);
    float woN = dot(wo, nrm), wiN = dot(wi, nrm), woH = dot(wo, h), nH = dot(nrm, h);
    float D = fwd_ndf(a2, nH), G = fwd_smith(a2, woN, wiN);
    v3 F = fwd_fresnel(col, V3(1.0f), woH);
    v3 w = F * (D * G * 0.25f / woN);
    bool front = (woN > SPECULAR_EPSILON) & (wiN > SPECULAR_EPSILON);
    return front ? w : V3(0.f);
}

__device__ __forceinline__ void bwd_pbr_specular(v3 col, v3 nrm, v3 wo, v3 wi, float alpha, float min_rough, v3& d_col, v3& d_nrm, v3& d_wo,
                                                 v3& d_wi, float& d_alpha, v3 d_out) {
    float al = clampf(alpha, min_rough * min_rough, 1.0f);
    float a2 = al * al;
    v3 hsum = wo + wi;
    v3 h = safe_normalize(hsum);
    float woN = dot(wo, nrm), wiN = dot(wi, nrm), woH = dot(wo, h), nH = dot(nrm, h);
    float D = fwd_ndf(a2, nH), G = fwd_smith(a2, woN, wiN);
    v3 F = fwd_fresnel(col, V3(1.0f), woH);
    bool front = (woN > SPECULAR_EPSILON) & (wiN > SPECULAR_EPSILON);
    if (!front) return;
    v3 dF = d_out * (D * G * 0.25f / woN);
    float dD = sum(d_out * F) * (G * 0.25f / woN);
    float dG = sum(d_out * F) * (D * 0.25f / woN);
    float d_woN = -sum(d_out * F) * (D * G * 0.25f / (woN * woN));
    float d_woH = 0.f, d_wiN = 0.f, d_nH = 0.f, d_a2 = 0.f;
    {  // Fresnel bwd (f90 == 1)
        float cc = clampf(woH, SPECULAR_EPSILON, 1.0f - SPECULAR_EPSILON);
        float scale = powf(fmaxf(1.0f - cc, 0.0f), 5.0f);
        d_col += dF * (1.0f - scale);
        if (woH >= SPECULAR_EPSILON && woH < 1.0f - SPECULAR_EPSILON) d_woH += sum(dF * (V3(1.0f) - col)) * -5.0f * powf(1.0f - woH, 4.0f);
    }
    {  // Smith bwd
        float li = fwd_lambda(a2, woN), lo = fwd_lambda(a2, wiN);
        float dl = -dG / ((1.0f + li + lo) * (1.0f + li + lo));
        bwd_lambda(a2, woN, d_a2, d_woN, dl);
        bwd_lambda(a2, wiN, d_a2, d_wiN, dl);
    }
    {  // NDF bwd
        float cc = clampf(nH, SPECULAR_EPSILON, 1.0f - SPECULAR_EPSILON);
        float c2 = cc * cc;
        float den = (a2 - 1.0f) * c2 + 1.0f;
        float den3 = den * den * den;
        d_a2 += dD * (1.0f - (a2 + 1.0f) * c2) / (PI_F * den3);
        if (nH > SPECULAR_EPSILON && nH < 1.0f - SPECULAR_EPSILON) d_nH += dD * -(4.0f * (a2 - 1.0f) * a2 * nH) / (PI_F * den3);
    }
    v3 d_h = V3(0.f);
    bwd_dot(nrm, h, d_nrm, d_h, d_nH);
    bwd_dot(wo, h, d_wo, d_h, d_woH);
    bwd_dot(wi, nrm, d_wi, d_nrm, d_wiN);
    bwd_dot(wo, nrm, d_wo, d_nrm, d_woN);
    v3 d_hsum = V3(0.f);
    bwd_safe_normalize(hsum, d_hsum, d_h);
    d_wo += d_hsum;
    d_wi += d_hsum;
    if (alpha > min_rough * min_rough) d_alpha += d_a2 * 2.0f * alpha;
}

// ---- sampling pdfs (kernel.cu:218-400); constants w.r.t. differentiation ----------------------------
__device__ __forceinline__ float ndf_ggx(float alpha, float c) {
    float a2 = alpha * alpha;
    float d = ((c * a2 - c) * c + 1.0f);
    return a2 / (d * d * PI_F);
}
__device__ __forceinline__ float g1_ggx(float a2, float c) {
    if (c <= 0.0f) return 0.0f;
    float c2 = c * c;
    float t2 = fmaxf(1.0f - c2, 0.0f) / c2;
    return 2.0f / (1.0f + sqrtf(1.0f + a2 * t2));
}
__device__ __forceinline__ float ggx_pdf(v3 N, v3 wo, v3 wi, float alpha) {
    v3 W = safe_normalize(N), U, Vv;
    onb(W, U, Vv);
    v3 wo_l = tolocal(wo, U, Vv, W), wi_l = tolocal(wi, U, Vv, W);
    float pdf = 0.0f;
    if (wo_l.z > 0.0f && wi_l.z > 0.0f) {
        v3 m = safe_normalize(wi_l + wo_l);
        float woH = dot(m, wo_l);
        float D = ndf_ggx(alpha, m.z);
        float G1 = g1_ggx(alpha * alpha, wo_l.z);
        pdf = G1 * D * fmaxf(0.0f, dot(wo_l, m)) / wo_l.z;
        pdf /= (4.0f * woH);
    }
    return pdf;
}
__device__ __forceinline__ void update_pdf(float& pdf, float opdf, float b) {
    if (b > 0.000001f) pdf += opdf * b;
}
__device__ __forceinline__ float bsdf_pdf(float pD, float pS, v3 N, v3 wo, v3 wi, float alpha) {
    float NdotL = dot(N, wi), NdotV = dot(N, wo);
    if (fminf(NdotV, NdotL) < 1e-6f) return 1.0f;
    float pdf = 0.0f;
    if (pD > 0.0f) update_pdf(pdf, fmaxf(dot(N, wi), 0.0f) / PI_F, pD);
    if (pS > 0.0f) update_pdf(pdf, ggx_pdf(N, wo, wi, alpha), 1.0f - pD);
    return pdf;
}
__device__ __forceinline__ v3 cosine_sample(v3 N, float u, float v, float& pdf) {
    N = safe_normalize(N);
    v3 dx, dy;
    onb(N, dx, dy);
    float phi = 2.0f * PI_F * u;
    float ct = sqrtf(v), st = sqrtf(1.0f - v);
    float sp, cp;
    sincosf(phi, &sp, &cp);
    pdf = fmaxf(0.000001f, ct / PI_F);
    return safe_normalize(dx * (cp * st) + dy * (sp * st) + N * ct);
}
__device__ __forceinline__ v3 sample_vndf(float alpha, v3 wo, float ux, float uy, float& pdf) {
    v3 Vh = safe_normalize(V3(alpha * wo.x, alpha * wo.y, wo.z));
    v3 T1 = (Vh.z < 0.9999f) ? safe_normalize(cross(V3(0.f, 0.f, 1.f), Vh)) : V3(1.f, 0.f, 0.f);
    v3 T2 = cross(Vh, T1);
    float r = sqrtf(ux), phi = (2.0f * PI_F) * uy;
    float sp, cp;
    sincosf(phi, &sp, &cp);
    float t1 = r * cp, t2 = r * sp;
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
    v3 Nh = T1 * t1 + T2 * t2 + Vh * sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2));
    v3 h = safe_normalize(V3(alpha * Nh.x, alpha * Nh.y, fmaxf(0.0f, Nh.z)));
    pdf = g1_ggx(alpha * alpha, wo.z) * ndf_ggx(alpha, h.z) * fmaxf(0.0f, dot(wo, h)) / wo.z;
    return h;
}
__device__ __forceinline__ v3 ggx_sample(v3 N, v3 wo, float u, float v, float alpha, float& pdf) {
    v3 W = safe_normalize(N), U, Vv;
    onb(W, U, Vv);
    v3 wo_l = safe_normalize(tolocal(wo, U, Vv, W));
    if (!(wo_l.z > 0.0f)) {
        pdf = 0.0f;
        return V3(0.f);
    }
    v3 h = sample_vndf(alpha, wo_l, u, v, pdf);
    float woH = dot(wo_l, h);
    v3 wi_l = h * (woH * 2.0f) - wo_l;
    pdf /= (4.0f * woH);
    return safe_normalize(toworld(wi_l, U, Vv, W));
}
__device__ __forceinline__ v3 bsdf_sample(float pD, float pS, v3 N, v3 wo, float sx, float sy, float sz, float alpha, float& pdf) {
    pdf = 0.0f;
    v3 wi;
    if (sz < pD) {
        if (pD < 0.0001f) {
            pdf = 1.0f;
            return N;
        }
        wi = cosine_sample(N, sx, sy, pdf);
        pdf *= pD;
        if (pS > 0.0f) update_pdf(pdf, ggx_pdf(N, wo, wi, alpha), 1.0f - pD);
    } else {
        wi = ggx_sample(N, wo, sx, sy, alpha, pdf);
        pdf *= 1.0f - pD;
        if (pD > 0.0f) update_pdf(pdf, fmaxf(dot(N, wi), 0.0f) / PI_F, pD);
    }
    return wi;
}

// ---- light probe (kernel.cu:123-211) ------------------------------------------------------------------
struct Probe {
    const float* light;  // [Hl,Wl,3]
    const float* pdf;    // [Hl,Wl]
    const float* rows;   // [Hl]
    const float* cols;   // [Hl,Wl]
    int Hl, Wl;
    int iters_rows, iters_cols;
};
__device__ __forceinline__ void dir_to_tc(v3 d, float& u, float& v) {
    u = atan2f(d.x, -d.z) / (2.0f * PI_F) + 0.5f;
    v = acosf(clampf(d.y, -1.0f, 1.0f)) / PI_F;
}
__device__ __forceinline__ v3 tc_to_dir(float u, float v) {
    float sphi, cphi, sth, cth;
    sincosf((u * 2.0f - 1.0f) * PI_F, &sphi, &cphi);
    sincosf(v * PI_F, &sth, &cth);
    return V3(sth * sphi, cth, -sth * cphi);
}
__device__ __forceinline__ float sample_cdf(const float* __restrict__ cdf, int size, int iters, float x, int& idx, float& pdf) {
    x = fminf(x, 0.99999994f);
    unsigned lo = 0, hi = (unsigned)size - 1;
    for (int i = 0; i < iters; ++i) {
        unsigned mid = (lo + hi) / 2;
        float c = cdf[mid];
        lo = x >= c ? mid : lo;
        hi = x < c ? mid : hi;
    }
    idx = (int)hi;
    float sample;
    if (idx == 0) {
        pdf = cdf[0];
        sample = x;
    } else {
        float d0 = cdf[idx], d1 = cdf[idx - 1];
        pdf = d0 - d1;
        sample = x - d1;
    }
    return fminf(sample / pdf, 0.99999994f);
}
__device__ __forceinline__ float light_pdf(const Probe& P, v3 dir) {
    float u, v;
    dir_to_tc(dir, u, v);
    int x = min(max((int)(u * (float)P.Wl), 0), P.Wl - 1);
    int y = min(max((int)(v * (float)P.Hl), 0), P.Hl - 1);
    float w = (float)P.Hl * (float)P.Wl / (2.0f * PI_F * PI_F * fmaxf(sinf(v * PI_F), 0.0001f));
    return P.pdf[y * P.Wl + x] * w;
}
__device__ __forceinline__ v3 light_sample(const Probe& P, float u, float v, float& pdf) {
    float row_pdf, col_pdf;
    int x, y;
    float ry = sample_cdf(P.rows, P.Hl, P.iters_rows, v, y, row_pdf);
    float rx = sample_cdf(P.cols + (int64_t)y * P.Wl, P.Wl, P.iters_cols, u, x, col_pdf);
    v3 d = tc_to_dir(((float)x + rx) / (float)P.Wl, ((float)y + ry) / (float)P.Hl);
    pdf = light_pdf(P, d);
    return d;
}

struct ShadeArgs {
    BvhView bvh;
    Probe probe;
    const float *ro, *pos, *nrm, *view_pos, *kd, *ks;  // image tensors [B,H,W,3]; view_pos [B,3]
    const int32_t* pix;                                // [n_cov] linear indices of the covered pixels
    int64_t n_cov;
    const int32_t* perms;                              // [P, n*n]
    int P;
    int64_t HW;
    int bsdf, n, G;  // G lanes per pixel
    uint32_t seed;
    int64_t view_offset, view_stride;  // RNG pixel index uses the GLOBAL view id b*view_stride + view_offset (view-sharded jobs)
    float shadow_scale;
    // ray buffers, slot r = k*2S + which*S + i  (which: 0 light sample, 1 BSDF sample)
    float4* ray_dk;            // [n_cov*2S]  (direction, k = MIS weight x sample weight): kept for the backward pass
    float* ray_contrib;        // [n_cov*2S, 6]  unshadowed (diff rgb, spec rgb) contribution
    uint64_t* vis_bits;        // [ceil(n_cov*2S / 64)]  bit = 1 -> unoccluded
    float *diff, *spec;                                  // fwd outputs [B,H,W,3]
    const float *g_diff, *g_spec;                        // bwd inputs
    float *g_pos, *g_nrm, *g_kd, *g_ks, *g_light;        // bwd outputs
    int tex_stride, g_tex_stride;                       // floats between pixels in kd / ks (g_kd / g_ks): 3, or 6 when both are channel halves of one [B,H,W,6] tensor
    // light gradient by binning instead of atomics (backward from saved samples, see k_light_*):
    float4* rec;               // [n_rays] (d loss / d light texel rgb, texel id as bits) per sample, LG_NONE = nothing to add
    uint32_t* hist;            // [nbins][workgroups of k_shade_grad]
    int64_t n_wg;
    int nbins;
};
constexpr int LG_TEXELS = 1024;            // probe texels per bin: 12 KB of LDS accumulators
constexpr uint32_t LG_NONE = 0xffffffffu;

struct PixelCtx {
    v3 pos, nrm, view, kd, ks, wo;
    float alpha, pD, pS;
};

// BSDF * light * MIS * weight of one sample, WITHOUT the visibility factor (which is linear and applied later).
// BWD: accumulates the gradient terms, already multiplied by `vis` (= V of this ray, cached by the forward pass).
// `k_saved` >= 0: the forward pass's k of this sample (backward from saved samples) instead of pdf_sum / weight; `k_out` = k.
template <bool BWD>
__device__ __forceinline__ void eval_sample(const ShadeArgs& A, const PixelCtx& c, v3 dir, float pdf_sum, float weight, float vis, v3 g_diff, v3 g_spec,
                                            v3& out_d, v3& out_s, v3& a_pos, v3& a_nrm, v3& a_kd, v3& a_ks, float k_saved, float& k_out, float4* rec = nullptr) {
    float u, v;
    dir_to_tc(dir, u, v);
    int lx = min(max((int)(u * (float)A.probe.Wl), 0), A.probe.Wl - 1);
    int ly = min(max((int)(v * (float)A.probe.Hl), 0), A.probe.Hl - 1);
    v3 light_col = ld3(A.probe.light + ((int64_t)ly * A.probe.Wl + lx) * 3);
    v3 d_ = V3(fwd_lambert(c.nrm, dir)), s_ = V3(0.f);
    v3 spec_col = V3(0.f);
    if (A.bsdf == 0) {
        spec_col = (V3(0.04f) * (1.0f - c.ks.z) + c.kd * c.ks.z) * (1.0f - c.ks.x);
        s_ = fwd_pbr_specular(spec_col, c.nrm, c.wo, dir, c.alpha, MIN_ROUGHNESS);
    }
    float k = k_saved >= 0.0f ? k_saved : (1.0f / fmaxf(pdf_sum, 0.0001f)) * weight;
    k_out = k;
    if (!BWD) {
        out_d = d_ * light_col * k;
        out_s = s_ * light_col * k;
        return;
    }
    k *= vis;
    if (k == 0.f) return;
    v3 lg = (g_diff * d_ + g_spec * s_) * k;
    // (a 64-bit compare-and-swap carrying two channels per atomic, atomics.hpp, LOSES here: bright texels are hit by thousands of
    // samples, the retries cost more than the third atomic -- measured 3.46 ms against 2.76 ms for the whole backward)
    float* gl = A.g_light + ((int64_t)ly * A.probe.Wl + lx) * 3;
#ifndef GS_EXPERIMENT_NO_LIGHT_GRAD
    if (rec) {
        *rec = make_float4(lg.x, lg.y, lg.z, __uint_as_float((uint32_t)(ly * A.probe.Wl + lx)));
    } else {
        if (lg.x != 0.f) atomicAdd(&gl[0], lg.x);
        if (lg.y != 0.f) atomicAdd(&gl[1], lg.y);
        if (lg.z != 0.f) atomicAdd(&gl[2], lg.z);
    }
#endif
    v3 dd = g_diff * light_col * k, ds = g_spec * light_col * k;
    v3 d_nrm = V3(0.f);
    if (A.bsdf == 0) {
        v3 d_spec_col = V3(0.f), d_wo = V3(0.f), d_wi = V3(0.f);
        float d_alpha = 0.f;
        bwd_pbr_specular(spec_col, c.nrm, c.wo, dir, c.alpha, MIN_ROUGHNESS, d_spec_col, d_nrm, d_wo, d_wi, d_alpha, ds);
        // spec_col = (0.04 (1 - ks.z) + kd ks.z) (1 - ks.x)
        a_kd -= d_spec_col * ((c.ks.x - 1.0f) * c.ks.z);
        a_ks.x += sum(d_spec_col * ((V3(0.04f) - c.kd) * c.ks.z - V3(0.04f)));
        a_ks.z -= sum(d_spec_col * (c.kd - V3(0.04f))) * (c.ks.x - 1.0f);
        a_ks.y += d_alpha * 2.0f * c.ks.y;
        v3 d_wo_raw = V3(0.f);
        bwd_safe_normalize(c.view - c.pos, d_wo_raw, d_wo);
        a_pos -= d_wo_raw;
    }
    if (dot(c.nrm, dir) > 0.0f) d_nrm += dir * (sum(dd) / PI_F);
    a_nrm += d_nrm;
}

__device__ __forceinline__ float group_sum(float v, int G) {
    for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ v3 group_sum(v3 v, int G) { return V3(group_sum(v.x, G), group_sum(v.y, G), group_sum(v.z, G)); }

__device__ __forceinline__ float vis_of(const ShadeArgs& A, int64_t r) {
    bool visible = (A.vis_bits[r >> 6] >> (r & 63)) & 1ull;
    return (visible ? 1.0f : 0.0f) * A.shadow_scale + (1.0f - A.shadow_scale);
}

// Pass 1 (fwd): sample generation.  Pass 3 (bwd): the same sampling with the cached visibility -> gradients.
template <bool BWD>
#ifndef GS_SAMPLES_WAVES
#define GS_SAMPLES_WAVES 4      // 128 VGPRs instead of 136: four waves per SIMD hide the CDF searches (measured 4.98 -> 4.83 ms for the forward)
#endif
__global__ void __launch_bounds__(256, BWD ? 1 : GS_SAMPLES_WAVES) k_shade_samples(ShadeArgs A) {
    const int tid = threadIdx.x;
    const int G = A.G;
    int64_t k = ((int64_t)blockIdx.x * 256 + tid) / G;   // covered-pixel slot
    int j = tid & (G - 1);
    if (k >= A.n_cov) return;  // whole groups retire together (G divides 64)
    int64_t gid = A.pix[k];
    PixelCtx c;
    int64_t b = gid / A.HW;
    c.pos = ld3(A.pos + 3 * gid);
    c.nrm = ld3(A.nrm + 3 * gid);
    c.view = ld3(A.view_pos + 3 * b);
    c.kd = ld3(A.kd + A.tex_stride * gid);
    c.ks = ld3(A.ks + A.tex_stride * gid);
    v3 g_diff = V3(0.f), g_spec = V3(0.f);
    if (BWD) {
        g_diff = ld3(A.g_diff + 3 * gid);
        g_spec = ld3(A.g_spec + 3 * gid);
    }
    const int n = A.n, S = n * n;
    float strata = 1.0f / (float)n, sample_frac = 1.0f / (float)(n * n);
    c.alpha = c.ks.y * c.ks.y;
    c.wo = safe_normalize(c.view - c.pos);
    float metallic = c.ks.z;
    v3 spec_color = V3(0.04f) * (1.0f - metallic) + c.kd * metallic;
    float diffuse_w = (1.0f - metallic) * luminance(c.kd);
    float specular_w;
    {  // albedo() (kernel.cu:82-95)
        v3 W = safe_normalize(c.nrm), U, Vv;
        onb(W, U, Vv);
        v3 wo_l = safe_normalize(tolocal(c.wo, U, Vv, W));
        float cosNO = wo_l.z;
        specular_w = (cosNO > 0.0f) ? luminance(fwd_fresnel(spec_color, V3(1.0f), cosNO)) : 0.0f;
    }
    c.pD = (diffuse_w + specular_w) > 0.0f ? diffuse_w / (diffuse_w + specular_w) : 1.0f;
    c.pS = 1.0f - c.pD;

    // pixel linear index (z*H + y)*W + x of the reference (kernel.cu:504), with z = the view's index in the GLOBAL batch:
    // rank r of a view-sharded job holds global views r, r + world, ... so an N-GPU step draws the 1-GPU step's samples
    uint32_t rng0 = pcg_out(A.seed) ^ pcg_out((uint32_t)(gid + (b * (A.view_stride - 1) + A.view_offset) * A.HW));
    uint32_t light_idx = pcg_out(rng0) % (uint32_t)A.P;
    uint32_t bsdf_idx = pcg_out(lcg_next(rng0)) % (uint32_t)A.P;
    const int32_t* perm_l = A.perms + (int64_t)light_idx * S;
    const int32_t* perm_b = A.perms + (int64_t)bsdf_idx * S;

    v3 a_pos = V3(0.f), a_nrm = V3(0.f), a_kd = V3(0.f), a_ks = V3(0.f);
    const int64_t r0 = k * 2 * S;
    for (int i = j; i < S; i += G) {
        uint32_t st = lcg_jump(rng0, 2u + 5u * (uint32_t)i);
        float r0f = u01(st); st = lcg_next(st);
        float r1 = u01(st); st = lcg_next(st);
        float r2 = u01(st); st = lcg_next(st);
        float r3 = u01(st); st = lcg_next(st);
        float r4 = u01(st);
        v3 od, os;
        // light importance sample
        int pl = perm_l[i];
        float sx = ((float)(pl % n) + r0f) * strata, sy = ((float)(pl / n) + r1) * strata;
        float pdf_light, pdf_b;
        v3 dir = light_sample(A.probe, sx, sy, pdf_light);
        pdf_b = bsdf_pdf(c.pD, c.pS, c.nrm, c.wo, dir, c.alpha);
        int64_t r = r0 + i;
        float kk;
        eval_sample<BWD>(A, c, dir, pdf_light + pdf_b, sample_frac, BWD ? vis_of(A, r) : 0.f, g_diff, g_spec, od, os, a_pos, a_nrm, a_kd, a_ks, -1.0f, kk);
        if (!BWD) {
            // k >= 0: its sign bit tells the trace kernel that the ray is dead (contribution exactly zero) without the 24-byte record
            const bool live = (od.x != 0.f) | (od.y != 0.f) | (od.z != 0.f) | (os.x != 0.f) | (os.y != 0.f) | (os.z != 0.f);
            A.ray_dk[r] = make_float4(dir.x, dir.y, dir.z, live ? kk : -kk);
            float2* rc = reinterpret_cast<float2*>(A.ray_contrib + 6 * r);          // 24-byte records, 8-byte aligned
            rc[0] = make_float2(od.x, od.y); rc[1] = make_float2(od.z, os.x); rc[2] = make_float2(os.y, os.z);
        }
        // BSDF sample
        int pb = perm_b[i];
        sx = ((float)(pb % n) + r2) * strata;
        sy = ((float)(pb / n) + r3) * strata;
        dir = bsdf_sample(c.pD, c.pS, c.nrm, c.wo, sx, sy, r4, c.alpha, pdf_b);
        pdf_light = light_pdf(A.probe, dir);
        r = r0 + S + i;
        eval_sample<BWD>(A, c, dir, pdf_light + pdf_b, sample_frac, BWD ? vis_of(A, r) : 0.f, g_diff, g_spec, od, os, a_pos, a_nrm, a_kd, a_ks, -1.0f, kk);
        if (!BWD) {
            // k >= 0: its sign bit tells the trace kernel that the ray is dead (contribution exactly zero) without the 24-byte record
            const bool live = (od.x != 0.f) | (od.y != 0.f) | (od.z != 0.f) | (os.x != 0.f) | (os.y != 0.f) | (os.z != 0.f);
            A.ray_dk[r] = make_float4(dir.x, dir.y, dir.z, live ? kk : -kk);
            float2* rc = reinterpret_cast<float2*>(A.ray_contrib + 6 * r);          // 24-byte records, 8-byte aligned
            rc[0] = make_float2(od.x, od.y); rc[1] = make_float2(od.z, os.x); rc[2] = make_float2(os.y, os.z);
        }
    }
    if (BWD) {
        a_pos = group_sum(a_pos, G);
        a_nrm = group_sum(a_nrm, G);
        a_kd = group_sum(a_kd, G);
        a_ks = group_sum(a_ks, G);
        if (j == 0) {
            float* p;
            p = A.g_pos + 3 * gid; p[0] = a_pos.x; p[1] = a_pos.y; p[2] = a_pos.z;
            p = A.g_nrm + 3 * gid; p[0] = a_nrm.x; p[1] = a_nrm.y; p[2] = a_nrm.z;
            p = A.g_kd + A.g_tex_stride * gid;  p[0] = a_kd.x;  p[1] = a_kd.y;  p[2] = a_kd.z;
            p = A.g_ks + A.g_tex_stride * gid;  p[0] = a_ks.x;  p[1] = a_ks.y;  p[2] = a_ks.z;
        }
    }
}

// Backward from the SAVED samples: the forward pass's ray buffer holds every sample's direction and k (MIS x sample weight; the
// reference treats both as constants in its backward pass too), so the gradient pass is a stream over 16 bytes per ray + the
// visibility bit -- no RNG replay, no CDF searches (17 dependent loads per light sample), no BSDF sampling: 233 -> ~100 VGPRs.
// Same arithmetic on the same k in the same order as k_shade_samples<true>: bit-identical per-pixel gradients.
#ifndef GS_GRAD_WAVES
#define GS_GRAD_WAVES 3
#endif
__global__ void __launch_bounds__(256, GS_GRAD_WAVES) k_shade_grad(ShadeArgs A) {
    __shared__ uint32_t s_hist[1024];
    const int tid = threadIdx.x;
    const int G = A.G;
    const bool binned = A.rec != nullptr;
    if (binned) {
        for (int b = tid; b < A.nbins; b += 256) s_hist[b] = 0u;
        __syncthreads();
    }
    int64_t k = ((int64_t)blockIdx.x * 256 + tid) / G;   // covered-pixel slot
    int j = tid & (G - 1);
    if (k < A.n_cov) {
        int64_t gid = A.pix[k];
        PixelCtx c;
        int64_t b = gid / A.HW;
        c.pos = ld3(A.pos + 3 * gid);
        c.nrm = ld3(A.nrm + 3 * gid);
        c.view = ld3(A.view_pos + 3 * b);
        c.kd = ld3(A.kd + A.tex_stride * gid);
        c.ks = ld3(A.ks + A.tex_stride * gid);
        const v3 g_diff = ld3(A.g_diff + 3 * gid), g_spec = ld3(A.g_spec + 3 * gid);
        const int S = A.n * A.n;
        c.alpha = c.ks.y * c.ks.y;
        c.wo = safe_normalize(c.view - c.pos);
        c.pD = c.pS = 0.f;
        v3 a_pos = V3(0.f), a_nrm = V3(0.f), a_kd = V3(0.f), a_ks = V3(0.f);
        const int64_t r0 = k * 2 * S;
        for (int i = j; i < S; i += G) {
#pragma unroll 1
            for (int which = 0; which < 2; ++which) {
                const int64_t r = r0 + which * S + i;
                float4 dk = A.ray_dk[r];
                dk.w = fabsf(dk.w);                                  // the sign bit is the trace kernel's "dead ray" flag
                const float vis = vis_of(A, r);
                float4 rec = make_float4(0.f, 0.f, 0.f, __uint_as_float(LG_NONE));
                if (dk.w * vis != 0.f) {
                    v3 od, os;
                    float kk;
                    eval_sample<true>(A, c, V3(dk.x, dk.y, dk.z), 0.f, 0.f, vis, g_diff, g_spec, od, os, a_pos, a_nrm, a_kd, a_ks, dk.w, kk,
                                      binned ? &rec : nullptr);
                }
                if (binned) {
                    if (rec.x == 0.f && rec.y == 0.f && rec.z == 0.f) rec.w = __uint_as_float(LG_NONE);
                    A.rec[r] = rec;
                    const uint32_t tx = __float_as_uint(rec.w);
                    if (tx != LG_NONE) atomicAdd(&s_hist[tx / LG_TEXELS], 1u);
                }
            }
        }
        a_pos = group_sum(a_pos, G);
        a_nrm = group_sum(a_nrm, G);
        a_kd = group_sum(a_kd, G);
        a_ks = group_sum(a_ks, G);
        if (j == 0) {
            float* p;
            p = A.g_pos + 3 * gid; p[0] = a_pos.x; p[1] = a_pos.y; p[2] = a_pos.z;
            p = A.g_nrm + 3 * gid; p[0] = a_nrm.x; p[1] = a_nrm.y; p[2] = a_nrm.z;
            p = A.g_kd + A.g_tex_stride * gid;  p[0] = a_kd.x;  p[1] = a_kd.y;  p[2] = a_kd.z;
            p = A.g_ks + A.g_tex_stride * gid;  p[0] = a_ks.x;  p[1] = a_ks.y;  p[2] = a_ks.z;
        }
    }
    if (binned) {
        __syncthreads();
        for (int b = tid; b < A.nbins; b += 256) A.hist[(int64_t)b * A.n_wg + blockIdx.x] = s_hist[b];
    }
}

// ---- light gradient without 36 M global float atomics (1.7 of the backward's 2.5 ms at the benchmark size) -------------------
// Float atomics are a flat ~21 G/s on gfx950 and the probe (256^2 x 3 floats) is far too large to privatise per workgroup.
// Instead the gradient pass writes one 16-byte record per sample (rgb + texel), the records are counting-sorted by probe BIN
// (1024 consecutive texels = 12 KB of accumulators) with per-workgroup histograms -- no global atomics -- and every bin is then
// reduced in LDS by a few workgroups.  All buffers live in the forward pass's (now dead) ray scratch.
//   k_shade_grad  : records + per-workgroup bin histogram            k_light_scan   : per bin, exclusive scan over the workgroups
//   k_light_base  : exclusive scan of the bin totals                 k_light_scatter: records -> bin-sorted order
//   k_light_reduce: LDS accumulation per (bin, slice), a few float atomics per texel and slice to finish
__global__ void __launch_bounds__(256) k_light_scan(uint32_t* __restrict__ hist, int64_t n_wg, int nbins, uint32_t* __restrict__ totals) {
    // one workgroup per bin: every thread owns a contiguous stretch of the bin's per-workgroup counts (sum, then one scan of the 256
    // sums, then the stretch again) -- two barriers, where a chunk-by-chunk scan took three per 256 counts (36 chunks at 4 x 512^2)
    __shared__ uint32_t s_wave[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* h = hist + (int64_t)blockIdx.x * n_wg;
    const int64_t per = (n_wg + 255) / 256, w0 = min(n_wg, tid * per), w1 = min(n_wg, w0 + per);
    uint32_t sum = 0u;
    for (int64_t w = w0; w < w1; ++w) sum += h[w];
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (int q = 0; q < wave; ++q) run += s_wave[q];
    for (int64_t w = w0; w < w1; ++w) {          // in place: count -> offset inside the bin
        const uint32_t v = h[w];
        h[w] = run;
        run += v;
    }
    if (tid == 255) totals[blockIdx.x] = run;    // (the last thread's stretch may be empty: run is then the bin's total all the same)
}

__global__ void k_light_base(const uint32_t* __restrict__ totals, int nbins, uint32_t* __restrict__ base) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t run = 0;
        for (int b = 0; b < nbins; ++b) {
            base[b] = run;
            run += totals[b];
        }
        base[nbins] = run;
    }
}

__global__ void __launch_bounds__(256) k_light_scatter(const float4* __restrict__ rec, int64_t n_rays, int64_t rays_per_wg, const uint32_t* __restrict__ hist,
                                                       const uint32_t* __restrict__ base, int nbins, float4* __restrict__ sorted) {
    __shared__ uint32_t s_cur[1024];
    const int tid = threadIdx.x;
    for (int b = tid; b < nbins; b += 256) s_cur[b] = base[b] + hist[(int64_t)b * gridDim.x + blockIdx.x];
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * rays_per_wg, r1 = min(n_rays, r0 + rays_per_wg);
    for (int64_t r = r0 + tid; r < r1; r += 256) {
        const float4 v = rec[r];
        const uint32_t tx = __float_as_uint(v.w);
        if (tx != LG_NONE) sorted[atomicAdd(&s_cur[tx / LG_TEXELS], 1u)] = v;
    }
}

#ifndef GS_LG_SLICES
#define GS_LG_SLICES 16
#endif
constexpr int LG_SLICES = GS_LG_SLICES;
constexpr int LG_NT = 1024;
// Accumulation in 64-bit FIXED POINT: ds_add_f32 costs 3 - 4 cycles per lane and CU on gfx950, ds_add_u64 0.36 - 0.65 (tools/micro/
// lds_atomic.hip), and the kernel is nothing but 3 adds per record.  The workgroup first takes max |v| over its records (they come from
// L2), then adds round(v * 2^e) with e such that n records of that size cannot overflow 2^61: the sum is exact to 2^-37 of the largest
// record -- closer to the true sum than any order of float adds, and independent of the order.  Records that are not finite (a diverged
// run) take the float path, which propagates them as before.
// `partial` != NULL: the slice's sums are STORED to partial[slice][texel][3] (zeros included) and k_light_sum adds the slices to g_light
// in slice order: no float atomics on global memory either.
__device__ __forceinline__ double fixed_scale(float vmax, uint32_t n) {
    // 2^e with |v| 2^e < 2^(61 - ceil(log2 n)) for every |v| <= vmax
    const int ev = (int)((__float_as_uint(vmax) >> 23) & 0xffu) - 126;          // vmax < 2^ev (denormals: ev = -126, still an upper bound)
    const int en = 32 - __clz((int)max(n, 1u) - 1 > 0 ? (int)max(n, 1u) - 1 : 0);           // ceil(log2 n), 0 for n = 1
    return ldexp(1.0, 61 - ev - en);
}

__global__ void __launch_bounds__(LG_NT) k_light_reduce(const float4* __restrict__ sorted, const uint32_t* __restrict__ base, int64_t n_texels,
                                                      float* __restrict__ g_light, float* __restrict__ partial) {
    __shared__ long long s_fix[LG_TEXELS * 3];
    __shared__ float s_max[LG_NT / 64];
    float* const s_acc = reinterpret_cast<float*>(s_fix);          // the float path uses the first half of the same block
    const int b = blockIdx.x, slice = blockIdx.y, tid = threadIdx.x;
    const uint32_t lo = base[b], hi = base[b + 1];
    const uint32_t n = hi - lo, per = (n + LG_SLICES - 1) / LG_SLICES;
    const uint32_t s0 = lo + min(n, slice * per), s1 = lo + min(n, (slice + 1) * per);
    float* out = partial ? partial + ((int64_t)slice * gridDim.x + b) * (LG_TEXELS * 3) : nullptr;
    if (s0 >= s1) {
        if (out)
            for (int t = tid; t < LG_TEXELS * 3; t += LG_NT) out[t] = 0.f;
        return;
    }
    // pass 1: the largest magnitude (NaN and inf surface as a non-finite maximum: fmaxf would drop a NaN, so they are tested for)
    float vmax = 0.f;
    bool finite = true;
    for (uint32_t q = s0 + tid; q < s1; q += LG_NT) {
        const float4 v = sorted[q];
        const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fabsf(v.z));
        finite = finite && (v.x - v.x == 0.f) && (v.y - v.y == 0.f) && (v.z - v.z == 0.f);
        vmax = fmaxf(vmax, m);
    }
    if (!finite) vmax = __builtin_inff();
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d, 64));
    if ((tid & 63) == 0) s_max[tid >> 6] = vmax;
    for (int t = tid; t < LG_TEXELS * 3; t += LG_NT) s_fix[t] = 0ll;
    __syncthreads();
    vmax = 0.f;
    for (int w = 0; w < LG_NT / 64; ++w) vmax = fmaxf(vmax, s_max[w]);
    const bool fixed = vmax - vmax == 0.f;                       // workgroup-uniform
    if (fixed) {
        const double scale = fixed_scale(vmax, s1 - s0);
        for (uint32_t q = s0 + tid; q < s1; q += LG_NT) {
            const float4 v = sorted[q];
            const uint32_t t = (__float_as_uint(v.w) - (uint32_t)b * LG_TEXELS) * 3;
            unsigned long long* a = reinterpret_cast<unsigned long long*>(s_fix) + t;
            if (v.x != 0.f) atomicAdd(a, (unsigned long long)__double2ll_rn((double)v.x * scale));
            if (v.y != 0.f) atomicAdd(a + 1, (unsigned long long)__double2ll_rn((double)v.y * scale));
            if (v.z != 0.f) atomicAdd(a + 2, (unsigned long long)__double2ll_rn((double)v.z * scale));
        }
        __syncthreads();
        const double inv = 1.0 / scale;
        const int64_t f0 = (int64_t)b * LG_TEXELS * 3, f1 = min(n_texels * 3, f0 + LG_TEXELS * 3);
        for (int t = tid; t < LG_TEXELS * 3; t += LG_NT) {
            const float v = (float)((double)s_fix[t] * inv);
            if (out) out[t] = v;
            else if (v != 0.f && f0 + t < f1) atomicAdd(&g_light[f0 + t], v);
        }
        return;
    }
    for (uint32_t q = s0 + tid; q < s1; q += LG_NT) {
        const float4 v = sorted[q];
        const uint32_t t = (__float_as_uint(v.w) - (uint32_t)b * LG_TEXELS) * 3;
        if (v.x != 0.f) atomicAdd(&s_acc[t], v.x);
        if (v.y != 0.f) atomicAdd(&s_acc[t + 1], v.y);
        if (v.z != 0.f) atomicAdd(&s_acc[t + 2], v.z);
    }
    __syncthreads();
    const int64_t f0 = (int64_t)b * LG_TEXELS * 3, f1 = min(n_texels * 3, f0 + LG_TEXELS * 3);
    for (int t = tid; t < LG_TEXELS * 3; t += LG_NT) {
        const float v = s_acc[t];
        if (out) out[t] = v;
        else if (v != 0.f && f0 + t < f1) atomicAdd(&g_light[f0 + t], v);
    }
}

// g_light[f] += sum over the slices of partial[slice][f], slices added in order (deterministic given the partials)
__global__ void __launch_bounds__(256) k_light_sum(const float* __restrict__ partial, int64_t n_padded, int64_t n_floats, float* __restrict__ g_light) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= n_floats) return;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LG_SLICES; ++k) s += partial[(int64_t)k * n_padded + f];
    if (s != 0.f) g_light[f] += s;
}

// Pass 2 (fwd): shadow rays through the implicit 4-ary BVH of bvh.hpp.
// Rays of one pixel scatter over the hemisphere, so lanes of a wave finish after very different numbers of steps
// (measured: the longest lane runs 2.1x the mean).  Instead of one ray per lane for the life of the wave, every wave
// owns a chunk of TRACE_CHUNK consecutive rays and lanes pull the next live ray whenever enough of them are idle:
//   * 64 rays at a time are staged into LDS by the whole wave (coalesced), dead rays dropped on the way -- a sample whose
//     unshadowed contribution is exactly zero (direction below the horizon of the shading normal: Lambert and the
//     front-facing specular test both vanish) cannot reach the outputs or any gradient whatever V is: not traced;
//   * idle lanes take staged rays by their rank in the idle mask (ballot + mbcnt, no atomics);
//   * results are bits in LDS (all visible, cleared by an LDS atomic on a hit) copied out as 64-bit words at the end.
#ifndef GS_LIGHT_BINNED
#define GS_LIGHT_BINNED 1
#endif
constexpr int TRACE_CHUNK = 1024;  // rays per wave
#ifndef TRACE_REFILL
#define TRACE_REFILL 16               // refill when at least this many lanes are idle
#endif
#ifndef TRACE_INNER
#define TRACE_INNER 4                 // traversal steps between two passes of the staging / refill logic
#endif
#ifndef TRACE_INNER_BREAK
#define TRACE_INNER_BREAK 1           // leave the inner loop when no lane has a ray
#endif

#ifndef TRACE_WAVES
#define TRACE_WAVES 1                 // waves per workgroup: each wave owns its own chunk and shares nothing, but a workgroup's slot is held until its
                                      // slowest wave is done -- forward family with 4 / 2 / 1 waves: 3.61 / 3.53 / 3.48 ms (one box, one call)
#endif
constexpr int TRACE_NT = 64 * TRACE_WAVES;
__global__ void __launch_bounds__(TRACE_NT, 6) k_shade_trace(ShadeArgs A, int64_t n_rays, int rays_per_pixel) {
    __shared__ int32_t stack[BVH_STACK * TRACE_NT];
    __shared__ float s_ray[TRACE_WAVES][6][64];
    __shared__ int32_t s_id[TRACE_WAVES][64];
    __shared__ uint32_t s_vis[TRACE_WAVES][TRACE_CHUNK / 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t chunk0 = ((int64_t)blockIdx.x * TRACE_WAVES + wave) * TRACE_CHUNK;
    if (chunk0 >= n_rays) return;
    const int chunk_n = (int)min((int64_t)TRACE_CHUNK, n_rays - chunk0);
    if (lane < TRACE_CHUNK / 32) s_vis[wave][lane] = 0xffffffffu;
    int32_t* const st = stack + threadIdx.x;
    int fetched = 0;               // rays of the chunk already staged (wave-uniform)
    int staged = 0, taken = 0;     // live rays in the stage buffer / handed out so far (wave-uniform)
    bool active = false;
    int my = 0;
    BvhRay ray;
    const bool any_tris = A.bvh.T > 0;
    for (;;) {
        if (taken == staged && fetched < chunk_n) {
            // stage the next 64 rays, live ones only (order preserved)
            const int i = fetched + lane;
            bool live = false;
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f;
            if (i < chunk_n && any_tris) {
                const int64_t r = chunk0 + i;
                const float4 d = A.ray_dk[r];
                live = !(__float_as_uint(d.w) >> 31);              // sign bit of k = dead ray (k_shade_samples)
                if (live) {
                    const int64_t gid = A.pix[r / rays_per_pixel];
                    if (A.ro) {
                        const float* o = A.ro + 3 * gid;
                        o0 = o[0]; o1 = o[1]; o2 = o[2];
                    } else {            // ro = gb_pos + gb_normal * 0.001, the reference's call site (render/render.py:131), rounded like its two ATen ops
                        const float *pp = A.pos + 3 * gid, *nn = A.nrm + 3 * gid;
                        o0 = pp[0] + nn[0] * 0.001f; o1 = pp[1] + nn[1] * 0.001f; o2 = pp[2] + nn[2] * 0.001f;
                    }
                    d0 = d.x; d1 = d.y; d2 = d.z;
                    live = (d0 == d0 && d1 == d1 && d2 == d2) && !(d0 == 0.f && d1 == 0.f && d2 == 0.f);
                }
            }
            const uint64_t lm = __ballot(live);
            if (live) {
                const int slot = __popcll(lm & ((1ull << lane) - 1ull));
                s_ray[wave][0][slot] = o0; s_ray[wave][1][slot] = o1; s_ray[wave][2][slot] = o2;
                s_ray[wave][3][slot] = d0; s_ray[wave][4][slot] = d1; s_ray[wave][5][slot] = d2;
                s_id[wave][slot] = i;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            fetched = min(fetched + 64, chunk_n);
            staged = __popcll(lm);
            taken = 0;
            continue;
        }
        const uint64_t idle = __ballot(!active);
        const int n_idle = __popcll(idle);
        if (taken < staged && (n_idle >= TRACE_REFILL || n_idle == 64)) {
            const int rank = __popcll(idle & ((1ull << lane) - 1ull));
            if (!active && rank < staged - taken) {
                const int slot = taken + rank;
                bvh_ray_init(ray, s_ray[wave][0][slot], s_ray[wave][1][slot], s_ray[wave][2][slot], s_ray[wave][3][slot], s_ray[wave][4][slot],
                             s_ray[wave][5][slot]);
                my = s_id[wave][slot];
                active = true;
            }
            taken = min(taken + n_idle, staged);
            __builtin_amdgcn_wave_barrier();
        } else if (n_idle == 64) {
            if (fetched >= chunk_n) break;   // nothing in flight, nothing staged, nothing left
            continue;
        }
        // TRACE_INNER traversal steps per pass of the hand-out logic above: staging / refill / exit tests are ~90 instructions per pass against
        // the ~150 of a node visit, and a lane whose ray ends here waits for the refill threshold anyway
#pragma unroll 1
        for (int it = 0; it < TRACE_INNER; ++it) {
            if (active) {
                const int state = bvh_step(A.bvh, ray, st, TRACE_NT);
                if (state != BVH_CONTINUE) {
                    if (state == BVH_HIT) atomicAnd(&s_vis[wave][my >> 5], ~(1u << (my & 31)));
                    active = false;
                }
            }
#if TRACE_INNER_BREAK
            if (__ballot(active) == 0ull) break;
#endif
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < TRACE_CHUNK / 64 && lane * 64 < chunk_n)
        A.vis_bits[(chunk0 >> 6) + lane] = (uint64_t)s_vis[wave][2 * lane] | ((uint64_t)s_vis[wave][2 * lane + 1] << 32);
}

// Tried and removed (round 5, VERDICT r4 item 5 i): ONE QUEUE for the whole launch instead of a 1024-ray chunk per wave -- a persistent grid (one wave
// per wave slot of the chip) claiming 64 rays at a time with an atomic on a launch-wide counter, visibility bits pre-set by a memset and cleared by
// 64-bit global atomics on a hit.  Bit-identical visibility (tests/test_ray_stage_fullsize_parity_gpu.py green), but gs_env_shade_fwd 4.65 ms against
// 3.39 with the chunks (same box, same call; the iteration 16.1 - 16.3 against 14.75): a claim stalls ALL of a wave's rays for the atomic's round
// trip 16 times as often as a chunk ends, and the chunks' sequential ray-record reads are gone.  The per-chunk tail it was meant to remove is the
// smaller cost.  With the shared-origin prefix and the two-launch split below, the ray stage has had its three structural variants; the chunked
// kernel stays.
// Tried and removed (round 4, profiles/r04_trace_variants.txt): a SHARED-ORIGIN variant for 2 n^2 % 64 == 0 -- the 64 rays a wave stages belong
// to one pixel, so the nodes whose box contains that origin (about 7 of the ~15 a miss visits) were found once per batch (lane k tests
// child k) and then slab-tested by all lanes in lock step, the divergent traversal starting below them from per-ray child masks (12 bytes per
// ray staged in LDS).  Visibility bits identical, but 5.40 ms against 3.58 for the forward family (110 VGPRs = 4 waves / SIMD; capped to 5 / 6
// waves: 6.7 / 7.6 ms with spills): the per-batch chain (K list: ~8 dependent record fetches; then ~8 uniform-address records per lane)
// is pure latency that four waves do not cover, and even free of latency the lock-step part still costs ~900 instructions per ray against
// the ~1900 it replaces -- a ceiling of ~18 % for the kernel.
// Pass 3 (fwd): per-pixel sum of V * contribution
__global__ void __launch_bounds__(256) k_shade_accumulate(ShadeArgs A) {
    const int G = A.G;
    int64_t k = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    int j = threadIdx.x & (G - 1);
    if (k >= A.n_cov) return;
    const int S2 = 2 * A.n * A.n;
    const int64_t r0 = k * S2;
    v3 ad = V3(0.f), as = V3(0.f);
    for (int i = j; i < S2; i += G) {
        int64_t r = r0 + i;
        float Vis = vis_of(A, r);
        const float* rc = A.ray_contrib + 6 * r;
        ad += V3(rc[0], rc[1], rc[2]) * Vis;
        as += V3(rc[3], rc[4], rc[5]) * Vis;
    }
    ad = group_sum(ad, G);
    as = group_sum(as, G);
    if (j == 0) {
        int64_t gid = A.pix[k];
        float* od = A.diff + 3 * gid;
        float* os = A.spec + 3 * gid;
        od[0] = ad.x; od[1] = ad.y; od[2] = ad.z;
        os[0] = as.x; os[1] = as.y; os[2] = as.z;
    }
}

int cdf_iters(int size) { return (int)std::ceil(std::log2((float)(size - 1))) + 1; }

int fill_args(ShadeArgs& A, const gs_bvh* bvh, const int32_t* pix, int64_t n_cov, const float* ro, const float* pos, const float* nrm,
              const float* view_pos, const float* kd, const float* ks, const float* light, const float* pdf, const float* rows, const float* cols,
              int64_t Hl, int64_t Wl, const int32_t* perms, int64_t P, int64_t B, int64_t H, int64_t W, int64_t view_offset, int64_t view_stride,
              int bsdf, int n, uint32_t seed, float shadow_scale, uint64_t* vis_bits) {
    GS_REQUIRE(bvh != nullptr, "env_shade: bvh is null");
    GS_REQUIRE(pix && pos && nrm && view_pos && kd && ks && light && pdf && rows && cols && perms && vis_bits, "env_shade: null pointer");
    GS_REQUIRE(bsdf >= 0 && bsdf <= 2, "env_shade: BSDF id must be 0 (pbr), 1 (diffuse) or 2 (white)");
    GS_REQUIRE(n >= 1 && n <= 64 && P >= 1 && Hl >= 2 && Wl >= 2, "env_shade: bad sample / probe configuration");
    GS_REQUIRE(B * H * W < (1ll << 31), "env_shade: too many pixels for 32-bit pixel ids");
    A.bvh = bvh_view(bvh);
    A.probe = {light, pdf, rows, cols, (int)Hl, (int)Wl, cdf_iters((int)Hl), cdf_iters((int)Wl)};
    A.pix = pix; A.n_cov = n_cov; A.ro = ro; A.pos = pos; A.nrm = nrm; A.view_pos = view_pos; A.kd = kd; A.ks = ks;
    A.tex_stride = (ks == kd + 3 && n_cov > 0) ? 6 : 3;        // ks = kd + 3: kd | ks interleaved in one 6-channel tensor (include/gshell_hip.h)
    A.g_tex_stride = 3;
    A.perms = perms; A.P = (int)P; A.HW = H * W; A.bsdf = bsdf; A.n = n;
    int G = 1;
    while (G * 2 <= std::min(n * n, 64)) G *= 2;
    A.G = G;
    GS_REQUIRE(view_offset >= 0 && view_stride >= 1, "env_shade: view_offset >= 0 and view_stride >= 1 (1-GPU: 0, 1)");
    A.view_offset = view_offset; A.view_stride = view_stride;
    A.seed = seed; A.shadow_scale = shadow_scale; A.vis_bits = vis_bits;
    return 0;
}

// ---- bilateral denoiser (denoising.cu:14-130) ---------------------------------------------------------
// out(p) = sum_t w(p,t) col(t),  w = exp(-|p-t|^2 / 2 sigma^2) * clamp(n_p . n_t, 1e-4, 1)^128 * exp(-|z_t - z_p| / max(dz * |p-t|, 1e-4))
// over the (2R+1)^2 window, R = 2 ceil(2.5 sigma) + 1 = 11 at the steady-state sigma = 2 (529 taps / pixel, 0.55 G taps per
// 4 x 512^2 call).  The reference kernel re-fetches 8 floats per tap through the texture path and evaluates powf + 2 expf +
// sqrtf per tap.  Here a 32 x 16 output tile is owned by a 512-thread workgroup that stages its (32+2R) x (16+2R) halo ONCE
// into LDS as two float4 planes (normal.xyz, z | dz, rgb) -- 68 KB at R = 11, two workgroups per CU -- and then every tap is two
// conflict-free ds_read_b128 (a wave = 2 tile rows of 32 consecutive pixels; the row stride is padded to 8 mod 16 pixels so the
// two half-rows of a 16-lane service group land on different halves of the 256-byte bank row), x^128 as seven squarings, ONE
// v_exp_f32 (the spatial Gaussian is folded into the exponent from a per-window-row LDS table) and ONE v_rcp_f32.
// Pixels outside the image are staged with a zero normal: their weight is clamp(0)^128 = 0 exactly, as if skipped.
#ifndef GS_BILATERAL_DUAL
#define GS_BILATERAL_DUAL 1
#endif
constexpr float FLT_EPS_D = 0.0001f;
constexpr int BIL_TX = 32, BIL_TY = 16, BIL_NT = BIL_TX * BIL_TY;
constexpr float LOG2E = 1.4426950408889634f;

__host__ __device__ inline int bil_row_stride(int R) {   // pixels; >= 32 + 2R, == 8 (mod 16)
    int w = BIL_TX + 2 * R;
    return w + ((8 - (w & 15)) & 15);
}

// DUAL: a SECOND colour image (the specular radiance next to the diffuse one) filtered with the same weights -- the weights depend
// on the guides only, and they are 20 of the ~24 VALU operations of a tap.  Third LDS plane (102 KB at R = 11: one workgroup / CU).
struct BilSecond {
    const float* col; float* out; const float* g_out; float* g_col;
};
template <bool BWD, bool DUAL>
__global__ void __launch_bounds__(BIL_NT, 2) k_bilateral_tile(const float* __restrict__ col, const float* __restrict__ nrm, const float* __restrict__ zdz,
                                                               int H, int W, float sigma, int R, float* __restrict__ out,
                                                               const float* __restrict__ g_out, float* __restrict__ g_col,
                                                               const float* __restrict__ mask, BilSecond S2) {
    extern __shared__ __attribute__((aligned(16))) float4 bil_smem[];
    // `mask` (optional): pixels with mask <= 0 are pixels whose filtered value nobody reads (no triangle covers them: the
    // composite multiplies their buffers by alpha = 0, the shader never looks at their gradient).  They still act as TAPS of
    // their neighbours, but their own 529-tap loops are skipped -- 85 % of the pixels of the benchmark's views; a tile without a
    // wanted pixel does not even stage its halo.  Their outputs are written as (0, 0, 0, 1e-4) / zero gradient.
    bool wanted = true;
    if (mask) {
        const int mx = blockIdx.x * BIL_TX + (threadIdx.x & (BIL_TX - 1)), my = blockIdx.y * BIL_TY + threadIdx.x / BIL_TX;
        wanted = mx < W && my < H && mask[(int64_t)blockIdx.z * H * W + (int64_t)my * W + mx] > 0.0f;
        if (!__syncthreads_or(wanted)) {
            if (mx < W && my < H) {
                const int64_t ci = (int64_t)blockIdx.z * H * W + (int64_t)my * W + mx;
                if (!BWD) *reinterpret_cast<float4*>(out + 4 * ci) = make_float4(0.f, 0.f, 0.f, 0.0001f);
                else { g_col[3 * ci] = 0.f; g_col[3 * ci + 1] = 0.f; g_col[3 * ci + 2] = 0.f; }
                if (DUAL) {
                    if (!BWD) *reinterpret_cast<float4*>(S2.out + 4 * ci) = make_float4(0.f, 0.f, 0.f, 0.0001f);
                    else { S2.g_col[3 * ci] = 0.f; S2.g_col[3 * ci + 1] = 0.f; S2.g_col[3 * ci + 2] = 0.f; }
                }
            }
            return;
        }
    }
    const int RS = bil_row_stride(R), TH = BIL_TY + 2 * R, TW = BIL_TX + 2 * R;
    float4* sA = bil_smem;                 // [TH][RS]  (nx, ny, nz, z)
    float4* sB = sA + TH * RS;             // [TH][RS]  (dz, c0, c1, c2)   c = colour (fwd) / upstream gradient (bwd)
    float4* sC = sB + TH * RS;             // DUAL: [TH][RS]  (c0, c1, c2, -) of the second image
    float2* sT = reinterpret_cast<float2*>(sB + (DUAL ? 2 : 1) * TH * RS);     // [(R+1)^2]  (log2e * d^2 / (2 sigma^2), d) for d^2 = fx^2 + fy^2
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * BIL_TX, y0 = blockIdx.y * BIL_TY;
    const int64_t base = (int64_t)blockIdx.z * H * W;
    for (int i = tid; i < (R + 1) * (R + 1); i += BIL_NT) {
        int fy = i / (R + 1), fx = i - fy * (R + 1);
        float d2 = (float)(fx * fx + fy * fy);
        sT[i] = make_float2(d2 / (2.0f * sigma * sigma) * LOG2E, sqrtf(d2));
    }
    for (int i = tid; i < TH * TW; i += BIL_NT) {
        int ty = i / TW, tx = i - ty * TW;
        int x = x0 + tx - R, y = y0 + ty - R;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 0.f, 0.f), c2 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x >= 0 && x < W && y >= 0 && y < H) {
            int64_t p = base + (int64_t)y * W + x;
            float2 zz = *reinterpret_cast<const float2*>(zdz + 2 * p);
            a = make_float4(nrm[3 * p], nrm[3 * p + 1], nrm[3 * p + 2], zz.x);
            if (BWD) {
                float4 g = *reinterpret_cast<const float4*>(g_out + 4 * p);
                b = make_float4(zz.y, g.x, g.y, g.z);
                if (DUAL) c2 = *reinterpret_cast<const float4*>(S2.g_out + 4 * p);
            } else {
                b = make_float4(zz.y, col[3 * p], col[3 * p + 1], col[3 * p + 2]);
                if (DUAL) c2 = make_float4(S2.col[3 * p], S2.col[3 * p + 1], S2.col[3 * p + 2], 0.f);
            }
        }
        sA[ty * RS + tx] = a;
        sB[ty * RS + tx] = b;
        if (DUAL) sC[ty * RS + tx] = c2;
    }
    __syncthreads();
    const int lx = tid & (BIL_TX - 1), ly = tid / BIL_TX;
    const int x = x0 + lx, y = y0 + ly;
    const float4 ca = sA[(ly + R) * RS + lx + R];
    const float cdz = sB[(ly + R) * RS + lx + R].x;
    float acc_w = 0.f, ax = 0.f, ay = 0.f, az = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
    const int R_eff = (mask && __ballot(wanted) == 0ull) ? -1 : R;       // a wave (two tile rows) without a wanted pixel: no taps
    for (int fy = -R_eff; fy <= R_eff; ++fy) {
        const float4* rowA = sA + (ly + R + fy) * RS + lx + R;
        const float4* rowB = sB + (ly + R + fy) * RS + lx + R;
        const float4* rowC = sC + (ly + R + fy) * RS + lx + R;
        const float2* rowT = sT + (fy < 0 ? -fy : fy) * (R + 1);
#pragma unroll 2
        for (int fx = -R; fx <= R; ++fx) {
            const float4 ta = rowA[fx];
            const float4 tb = rowB[fx];
            const float2 tt = rowT[fx < 0 ? -fx : fx];       // wave-uniform address: LDS broadcast
            float d = __builtin_fmaf(ta.x, ca.x, __builtin_fmaf(ta.y, ca.y, ta.z * ca.z));
            float wn = fminf(fmaxf(d, FLT_EPS_D), 1.0f);
            wn *= wn; wn *= wn; wn *= wn; wn *= wn; wn *= wn; wn *= wn; wn *= wn;          // ^128
            // the reference's backward uses the TAP's dz in the depth weight (denoising.cu:118); replicated
            float den = fmaxf((BWD ? tb.x : cdz) * tt.y, FLT_EPS_D);
            float q = fabsf(ta.w - ca.w) * __builtin_amdgcn_rcpf(den);
            float w = wn * __builtin_amdgcn_exp2f(-__builtin_fmaf(q, LOG2E, tt.x));
            ax = __builtin_fmaf(tb.y, w, ax);
            ay = __builtin_fmaf(tb.z, w, ay);
            az = __builtin_fmaf(tb.w, w, az);
            if (DUAL) {
                const float4 tc = rowC[fx];
                bx = __builtin_fmaf(tc.x, w, bx);
                by = __builtin_fmaf(tc.y, w, by);
                bz = __builtin_fmaf(tc.z, w, bz);
            }
            if (!BWD) acc_w += w;
        }
    }
    if (x >= W || y >= H) return;
    const int64_t ci = base + (int64_t)y * W + x;
    if (!wanted) ax = ay = az = bx = by = bz = acc_w = 0.f;
    if (DUAL) {
        if (!BWD) {
            *reinterpret_cast<float4*>(S2.out + 4 * ci) = make_float4(bx, by, bz, fmaxf(acc_w, 0.0001f));
        } else {
            S2.g_col[3 * ci] = bx;
            S2.g_col[3 * ci + 1] = by;
            S2.g_col[3 * ci + 2] = bz;
        }
    }
    if (!BWD) {
        *reinterpret_cast<float4*>(out + 4 * ci) = make_float4(ax, ay, az, fmaxf(acc_w, 0.0001f));
    } else {
        g_col[3 * ci] = ax;
        g_col[3 * ci + 1] = ay;
        g_col[3 * ci + 2] = az;
    }
}

// Windows too wide for the LDS tile (sigma > ~3.5): one thread per pixel, taps through L1/L2 (the reference's structure).
template <bool BWD>
__global__ void __launch_bounds__(256) k_bilateral_direct(const float* __restrict__ col, const float* __restrict__ nrm, const float* __restrict__ zdz,
                                                          int64_t B, int H, int W, float sigma, int rad, float* __restrict__ out,
                                                          const float* __restrict__ g_out, float* __restrict__ g_col,
                                                          const float* __restrict__ mask) {
    int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    int64_t b = blockIdx.z;
    if (x >= W || y >= H) return;
    int64_t base = b * (int64_t)H * W;
    int64_t ci = base + (int64_t)y * W + x;
    if (mask && !(mask[ci] > 0.0f)) rad = -1;
    float cnx = nrm[3 * ci], cny = nrm[3 * ci + 1], cnz = nrm[3 * ci + 2];
    float cz = zdz[2 * ci], cdz = zdz[2 * ci + 1];
    float variance = sigma * sigma;
    float acc_w = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
    for (int fy = -rad; fy <= rad; ++fy) {
        int yy = y + fy;
        if (yy < 0 || yy >= H) continue;
        for (int fx = -rad; fx <= rad; ++fx) {
            int xx = x + fx;
            if (xx < 0 || xx >= W) continue;
            int64_t ti = base + (int64_t)yy * W + xx;
            float tnx = nrm[3 * ti], tny = nrm[3 * ti + 1], tnz = nrm[3 * ti + 2];
            float tz = zdz[2 * ti], tdz = zdz[2 * ti + 1];
            float dist_sqr = (float)(fx * fx + fy * fy);
            float dist = sqrtf(dist_sqr);
            float w_xy = expf(-dist_sqr / (2.0f * variance));
            float w_normal = powf(fminf(fmaxf(tnx * cnx + tny * cny + tnz * cnz, FLT_EPS_D), 1.0f), 128.0f);
            float w_depth = expf(-(fabsf(tz - cz) / fmaxf((BWD ? tdz : cdz) * dist, FLT_EPS_D)));
            float w = w_xy * w_normal * w_depth;
            if (!BWD) {
                ax += col[3 * ti] * w;
                ay += col[3 * ti + 1] * w;
                az += col[3 * ti + 2] * w;
                acc_w += w;
            } else {
                ax += w * g_out[4 * ti];
                ay += w * g_out[4 * ti + 1];
                az += w * g_out[4 * ti + 2];
            }
        }
    }
    if (!BWD) {
        out[4 * ci] = ax;
        out[4 * ci + 1] = ay;
        out[4 * ci + 2] = az;
        out[4 * ci + 3] = fmaxf(acc_w, 0.0001f);
    } else {
        g_col[3 * ci] = ax;
        g_col[3 * ci + 1] = ay;
        g_col[3 * ci + 2] = az;
    }
}

int bilateral_radius(float sigma) { return 2 * (int)std::ceil(sigma * 2.5f) + 1; }

size_t bilateral_smem_bytes(int R, int planes = 2) {
    return (size_t)planes * (BIL_TY + 2 * R) * bil_row_stride(R) * sizeof(float4) + (size_t)(R + 1) * (R + 1) * sizeof(float2);
}

template <bool BWD>
int launch_bilateral(const float* col, const float* nrm, const float* zdz, const float* mask, int64_t B, int64_t H, int64_t W, float sigma, float* out,
                     const float* g_out, float* g_col, hipStream_t stream) {
    const int R = bilateral_radius(sigma);
    const size_t smem = bilateral_smem_bytes(R);
    if (smem <= 80 * 1024) {      // two workgroups per CU (160 KB of LDS)
        // set on every launch: the attribute is per device / per function, and a per-process flag would be neither
        GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bilateral_tile<BWD, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        dim3 grid((unsigned)gs::cdiv(W, BIL_TX), (unsigned)gs::cdiv(H, BIL_TY), (unsigned)B);
        hipLaunchKernelGGL((k_bilateral_tile<BWD, false>), grid, dim3(BIL_NT), smem, stream, col, nrm, zdz, (int)H, (int)W, sigma, R, out, g_out, g_col, mask,
                           BilSecond{});
    } else {
        dim3 grid((unsigned)gs::cdiv(W, 16), (unsigned)gs::cdiv(H, 16), (unsigned)B);
        hipLaunchKernelGGL(k_bilateral_direct<BWD>, grid, dim3(256), 0, stream, col, nrm, zdz, B, (int)H, (int)W, sigma, R, out, g_out, g_col, mask);
    }
    GS_LAUNCH_CHECK();
    return 0;
}

// two colour images with shared guides: one dual launch while the three LDS planes fit (sigma <= ~2.4), else two single passes
template <bool BWD>
int launch_bilateral2(const float* col_a, const float* col_b, const float* nrm, const float* zdz, const float* mask, int64_t B, int64_t H, int64_t W,
                      float sigma, float* out_a, float* out_b, const float* g_out_a, const float* g_out_b, float* g_col_a, float* g_col_b,
                      hipStream_t stream) {
    const int R = bilateral_radius(sigma);
    const size_t smem = bilateral_smem_bytes(R, 3);
    if (smem > 160 * 1024 || !GS_BILATERAL_DUAL) {
        if (int rc = launch_bilateral<BWD>(col_a, nrm, zdz, mask, B, H, W, sigma, out_a, g_out_a, g_col_a, stream)) return rc;
        return launch_bilateral<BWD>(col_b, nrm, zdz, mask, B, H, W, sigma, out_b, g_out_b, g_col_b, stream);
    }
    GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bilateral_tile<BWD, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    dim3 grid((unsigned)gs::cdiv(W, BIL_TX), (unsigned)gs::cdiv(H, BIL_TY), (unsigned)B);
    hipLaunchKernelGGL((k_bilateral_tile<BWD, true>), grid, dim3(BIL_NT), smem, stream, col_a, nrm, zdz, (int)H, (int)W, sigma, R, out_a, g_out_a, g_col_a,
                       mask, BilSecond{col_b, out_b, g_out_b, g_col_b});
    GS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int64_t gs_env_shade_vis_words(int64_t n_cov, int n_samples_x) {
    return (n_cov * 2 * n_samples_x * n_samples_x + 63) / 64 + 1;
}

extern "C" int64_t gs_env_shade_scratch_bytes(int64_t n_cov, int n_samples_x) {
    return n_cov * 2 * n_samples_x * n_samples_x * (int64_t)(4 + 6) * 4 + 256;
}

// Forward pass over the covered pixels in CHUNKS whose per-sample records (40 B per ray) fit `scratch_bytes` (< 0: one chunk).  Every pixel's
// samples, shadow rays and sums are independent of its neighbours in the list (the RNG hashes the GLOBAL pixel index, a ray's visibility bit does
// not depend on which rays travel with it), so the outputs are bit-identical for every chunk size; a chunk is a multiple of 64 pixels, which keeps
// its visibility bits on whole words of `vis_bits`.  With more than one chunk the records of all but the last are gone afterwards: the caller
// back-propagates with gs_env_shade_bwd (sampler replay from the cached bits), not gs_env_shade_bwd_saved.
static int env_shade_fwd(const gs_bvh* bvh, const int32_t* pix, int64_t n_cov, const float* ro, const float* gb_pos, const float* gb_normal,
                         const float* view_pos, const float* gb_kd, const float* gb_ks, const float* light, const float* pdf,
                         const float* rows, const float* cols, int64_t Hl, int64_t Wl, const int32_t* perms, int64_t P, int64_t B,
                         int64_t H, int64_t W, int64_t view_offset, int64_t view_stride, int bsdf, int n_samples_x, uint32_t rnd_seed,
                         float shadow_scale, void* scratch, int64_t scratch_bytes, uint64_t* vis_bits, float* diff, float* spec, gs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (B * H * W == 0) return 0;
    if (diff) GS_HIP_CHECK(hipMemsetAsync(diff, 0, (size_t)B * H * W * 12, stream));
    if (spec) GS_HIP_CHECK(hipMemsetAsync(spec, 0, (size_t)B * H * W * 12, stream));
    if (n_cov == 0) return 0;
    GS_REQUIRE(scratch != nullptr, "gs_env_shade_fwd: null scratch");
    const int64_t S2 = 2ll * n_samples_x * n_samples_x;
    int64_t chunk = n_cov;
    if (scratch_bytes >= 0 && scratch_bytes < gs_env_shade_scratch_bytes(n_cov, n_samples_x)) {
        chunk = (scratch_bytes - 256) / (S2 * 40) / 64 * 64;
        GS_REQUIRE(chunk >= 64, "gs_env_shade_fwd_bounded: the scratch does not hold the records of 64 pixels");
    }
    for (int64_t off = 0; off < n_cov; off += chunk) {
        const int64_t cnt = n_cov - off < chunk ? n_cov - off : chunk;
        ShadeArgs A{};
        int rc = fill_args(A, bvh, pix + off, cnt, ro, gb_pos, gb_normal, view_pos, gb_kd, gb_ks, light, pdf, rows, cols, Hl, Wl, perms, P, B, H, W, view_offset,
                           view_stride, bsdf, n_samples_x, rnd_seed, shadow_scale, vis_bits + off * S2 / 64);
        if (rc) return rc;
        const int64_t n_rays = cnt * S2;
        A.ray_dk = (float4*)scratch;
        A.ray_contrib = (float*)(A.ray_dk + n_rays);
        A.diff = diff;
        A.spec = spec;
        int64_t lanes = cnt * A.G;
        hipLaunchKernelGGL(k_shade_samples<false>, dim3((unsigned)gs::cdiv(lanes, 256)), dim3(256), 0, stream, A);
        hipLaunchKernelGGL(k_shade_trace, dim3((unsigned)gs::cdiv(n_rays, TRACE_WAVES * TRACE_CHUNK)), dim3(TRACE_NT), 0, stream, A, n_rays, (int)S2);
        if (diff && spec) hipLaunchKernelGGL(k_shade_accumulate, dim3((unsigned)gs::cdiv(lanes, 256)), dim3(256), 0, stream, A);
        GS_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int gs_env_shade_fwd(const gs_bvh* bvh, const int32_t* pix, int64_t n_cov, const float* ro, const float* gb_pos, const float* gb_normal,
                                const float* view_pos, const float* gb_kd, const float* gb_ks, const float* light, const float* pdf,
                                const float* rows, const float* cols, int64_t Hl, int64_t Wl, const int32_t* perms, int64_t P, int64_t B,
                                int64_t H, int64_t W, int64_t view_offset, int64_t view_stride, int bsdf, int n_samples_x, uint32_t rnd_seed,
                                float shadow_scale, void* scratch, uint64_t* vis_bits, float* diff, float* spec, gs_stream_t stream_) {
    return env_shade_fwd(bvh, pix, n_cov, ro, gb_pos, gb_normal, view_pos, gb_kd, gb_ks, light, pdf, rows, cols, Hl, Wl, perms, P, B, H, W, view_offset,
                         view_stride, bsdf, n_samples_x, rnd_seed, shadow_scale, scratch, -1, vis_bits, diff, spec, stream_);
}

extern "C" int gs_env_shade_fwd_bounded(const gs_bvh* bvh, const int32_t* pix, int64_t n_cov, const float* ro, const float* gb_pos, const float* gb_normal,
                                        const float* view_pos, const float* gb_kd, const float* gb_ks, const float* light, const float* pdf,
                                        const float* rows, const float* cols, int64_t Hl, int64_t Wl, const int32_t* perms, int64_t P, int64_t B,
                                        int64_t H, int64_t W, int64_t view_offset, int64_t view_stride, int bsdf, int n_samples_x, uint32_t rnd_seed,
                                        float shadow_scale, void* scratch, int64_t scratch_bytes, uint64_t* vis_bits, float* diff, float* spec,
                                        gs_stream_t stream_) {
    GS_REQUIRE(scratch_bytes >= 0, "gs_env_shade_fwd_bounded: negative scratch size");
    return env_shade_fwd(bvh, pix, n_cov, ro, gb_pos, gb_normal, view_pos, gb_kd, gb_ks, light, pdf, rows, cols, Hl, Wl, perms, P, B, H, W, view_offset,
                         view_stride, bsdf, n_samples_x, rnd_seed, shadow_scale, scratch, scratch_bytes, vis_bits, diff, spec, stream_);
}

// Gradients of the pixels of A (n_cov of them) from their per-sample records at A.ray_dk: [n_rays] float4 (direction, k) | [n_rays] 6 floats (dead
// contributions, reused as scratch).  g_light is ACCUMULATED, the per-pixel gradients are written.
static void launch_saved_grad(ShadeArgs& A, int64_t n_cov, int n_samples_x, int64_t Hl, int64_t Wl, hipStream_t stream) {
    // the forward scratch = [n_rays] float4 (direction, k) | [n_rays] 6 floats of unshadowed contributions (dead by now).
    // Records go over the contributions, the sorted records over (direction, k) once the gradient kernel has read them; the
    // histograms / offsets use the tail of the contribution region (8 bytes per ray are left).
    const int64_t S2 = 2ll * n_samples_x * n_samples_x, n_rays = n_cov * S2;
    const int64_t n_wg = gs::cdiv(n_cov * A.G, 256), rays_per_wg = (256 / A.G) * S2;
    const int64_t n_texels = Hl * Wl, nbins = gs::cdiv(n_texels, LG_TEXELS);
    const int64_t tail_bytes = n_rays * 8, need = (n_wg * nbins + 2 * nbins + 2) * 4;
    const bool binned = GS_LIGHT_BINNED && nbins <= 1024 && need <= tail_bytes && n_rays < (1ll << 32);
    const int64_t need_aligned = (need + 15) / 16 * 16, partial_bytes = (int64_t)LG_SLICES * nbins * LG_TEXELS * 3 * 4;
    const bool partials = binned && need_aligned + partial_bytes <= tail_bytes;
    if (binned) {
        A.rec = (float4*)(A.ray_dk + n_rays);
        A.hist = (uint32_t*)(A.rec + n_rays);
        A.nbins = (int)nbins;
        A.n_wg = n_wg;
    }
    hipLaunchKernelGGL(k_shade_grad, dim3((unsigned)n_wg), dim3(256), 0, stream, A);
    if (binned) {
        uint32_t* totals = A.hist + n_wg * nbins;
        uint32_t* base = totals + nbins;
        hipLaunchKernelGGL(k_light_scan, dim3((unsigned)nbins), dim3(256), 0, stream, A.hist, n_wg, (int)nbins, totals);
        hipLaunchKernelGGL(k_light_base, dim3(1), dim3(64), 0, stream, totals, (int)nbins, base);
        hipLaunchKernelGGL(k_light_scatter, dim3((unsigned)n_wg), dim3(256), 0, stream, A.rec, n_rays, rays_per_wg, A.hist, base, (int)nbins, A.ray_dk);
        float* partial = partials ? (float*)((char*)A.hist + need_aligned) : nullptr;
        hipLaunchKernelGGL(k_light_reduce, dim3((unsigned)nbins, LG_SLICES), dim3(LG_NT), 0, stream, A.ray_dk, base, n_texels, A.g_light, partial);
        if (partial)
            hipLaunchKernelGGL(k_light_sum, dim3((unsigned)gs::cdiv(n_texels * 3, 256)), dim3(256), 0, stream, partial, nbins * (int64_t)LG_TEXELS * 3,
                               n_texels * 3, A.g_light);
    }
}

static int env_shade_bwd(const gs_bvh* bvh, const int32_t* pix, int64_t n_cov, const float* gb_pos, const float* gb_normal,
                         const float* view_pos, const float* gb_kd, const float* gb_ks, const float* light, const float* pdf,
                         const float* rows, const float* cols, int64_t Hl, int64_t Wl, const int32_t* perms, int64_t P, int64_t B,
                         int64_t H, int64_t W, int64_t view_offset, int64_t view_stride, int bsdf, int n_samples_x, uint32_t rnd_seed,
                         float shadow_scale, const uint64_t* vis_bits, const void* saved_rays, const float* g_diff, const float* g_spec, float* g_pos,
                         float* g_normal, float* g_kd, float* g_ks, float* g_light, gs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (B * H * W == 0) return 0;
    GS_REQUIRE(g_diff && g_spec && g_pos && g_normal && g_kd && g_ks && g_light, "gs_env_shade_bwd: null pointer");
    size_t nb = (size_t)B * H * W * 12;
    GS_HIP_CHECK(hipMemsetAsync(g_pos, 0, nb, stream));
    GS_HIP_CHECK(hipMemsetAsync(g_normal, 0, nb, stream));
    const bool g_tex6 = g_ks == g_kd + 3;          // g_kd | g_ks interleaved in one [B,H,W,6] tensor
    GS_HIP_CHECK(hipMemsetAsync(g_kd, 0, g_tex6 ? 2 * nb : nb, stream));
    if (!g_tex6) GS_HIP_CHECK(hipMemsetAsync(g_ks, 0, nb, stream));
    if (n_cov == 0) return 0;
    ShadeArgs A{};
    int rc = fill_args(A, bvh, pix, n_cov, gb_pos /* ro unused in bwd */, gb_pos, gb_normal, view_pos, gb_kd, gb_ks, light, pdf, rows, cols, Hl, Wl,
                       perms, P, B, H, W, view_offset, view_stride, bsdf, n_samples_x, rnd_seed, shadow_scale, const_cast<uint64_t*>(vis_bits));
    if (rc) return rc;
    A.g_diff = g_diff; A.g_spec = g_spec; A.g_pos = g_pos; A.g_nrm = g_normal; A.g_kd = g_kd; A.g_ks = g_ks; A.g_light = g_light;
    A.g_tex_stride = g_tex6 ? 6 : 3;
    if (saved_rays) {
        A.ray_dk = (float4*)const_cast<void*>(saved_rays);
        launch_saved_grad(A, n_cov, n_samples_x, Hl, Wl, stream);
    } else {
#if GS_ORACLE_KERNELS
        hipLaunchKernelGGL(k_shade_samples<true>, dim3((unsigned)gs::cdiv(n_cov * A.G, 256)), dim3(256), 0, stream, A);
#else
        GS_ORACLE_ONLY("gs_env_shade_bwd (sampler-replay backward)");      // shipped: gs_env_shade_bwd_saved / gs_env_shade_bwd_bounded
#endif
    }
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_env_shade_bwd(const gs_bvh* bvh, const int32_t* pix, int64_t n_cov, const float* gb_pos, const float* gb_normal,
                                const float* view_pos, const float* gb_kd, const float* gb_ks, const float* light, const float* pdf,
                                const float* rows, const float* cols, int64_t Hl, int64_t Wl, const int32_t* perms, int64_t P, int64_t B,
                                int64_t H, int64_t W, int64_t view_offset, int64_t view_stride, int bsdf, int n_samples_x, uint32_t rnd_seed,
                                float shadow_scale, const uint64_t* vis_bits, const float* g_diff, const float* g_spec, float* g_pos, float* g_normal, float* g_kd, float* g_ks, float* g_light,
                                gs_stream_t stream_) {
    return env_shade_bwd(bvh, pix, n_cov, gb_pos, gb_normal, view_pos, gb_kd, gb_ks, light, pdf, rows, cols, Hl, Wl, perms, P, B, H, W, view_offset,
                         view_stride, bsdf, n_samples_x, rnd_seed, shadow_scale, vis_bits, nullptr, g_diff, g_spec, g_pos, g_normal, g_kd, g_ks, g_light,
                         stream_);
}

extern "C" int gs_env_shade_bwd_saved(const gs_bvh* bvh, const int32_t* pix, int64_t n_cov, const float* gb_pos, const float* gb_normal,
                                      const float* view_pos, const float* gb_kd, const float* gb_ks, const float* light, const float* pdf,
                                      const float* rows, const float* cols, int64_t Hl, int64_t Wl, const int32_t* perms, int64_t P, int64_t B,
                                      int64_t H, int64_t W, int64_t view_offset, int64_t view_stride, int bsdf, int n_samples_x, uint32_t rnd_seed,
                                      float shadow_scale, const uint64_t* vis_bits, const void* fwd_scratch, const float* g_diff, const float* g_spec,
                                      float* g_pos, float* g_normal, float* g_kd, float* g_ks, float* g_light, gs_stream_t stream_) {
    GS_REQUIRE(fwd_scratch != nullptr, "gs_env_shade_bwd_saved: the forward pass's scratch buffer is required");
    return env_shade_bwd(bvh, pix, n_cov, gb_pos, gb_normal, view_pos, gb_kd, gb_ks, light, pdf, rows, cols, Hl, Wl, perms, P, B, H, W, view_offset,
                         view_stride, bsdf, n_samples_x, rnd_seed, shadow_scale, vis_bits, fwd_scratch, g_diff, g_spec, g_pos, g_normal, g_kd, g_ks, g_light,
                         stream_);
}

// Backward of a frame whose records do not fit a fixed budget (see gs_env_shade_fwd_bounded): per chunk of covered pixels (a multiple of 64, the
// forward's rule) the sampler REGENERATES the chunk's records into `scratch` (k_shade_samples<false>: same RNG streams, no rays -- visibility comes
// from the forward's cached bits) and the saved-samples gradient kernels run on them.  Per-pixel gradients bit-identical to gs_env_shade_bwd_saved,
// the probe gradient up to float-atomic order.
extern "C" int gs_env_shade_bwd_bounded(const gs_bvh* bvh, const int32_t* pix, int64_t n_cov, const float* gb_pos, const float* gb_normal,
                                        const float* view_pos, const float* gb_kd, const float* gb_ks, const float* light, const float* pdf,
                                        const float* rows, const float* cols, int64_t Hl, int64_t Wl, const int32_t* perms, int64_t P, int64_t B,
                                        int64_t H, int64_t W, int64_t view_offset, int64_t view_stride, int bsdf, int n_samples_x, uint32_t rnd_seed,
                                        float shadow_scale, const uint64_t* vis_bits, void* scratch, int64_t scratch_bytes, const float* g_diff,
                                        const float* g_spec, float* g_pos, float* g_normal, float* g_kd, float* g_ks, float* g_light, gs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (B * H * W == 0) return 0;
    GS_REQUIRE(g_diff && g_spec && g_pos && g_normal && g_kd && g_ks && g_light, "gs_env_shade_bwd_bounded: null pointer");
    size_t nb = (size_t)B * H * W * 12;
    GS_HIP_CHECK(hipMemsetAsync(g_pos, 0, nb, stream));
    GS_HIP_CHECK(hipMemsetAsync(g_normal, 0, nb, stream));
    const bool g_tex6 = g_ks == g_kd + 3;
    GS_HIP_CHECK(hipMemsetAsync(g_kd, 0, g_tex6 ? 2 * nb : nb, stream));
    if (!g_tex6) GS_HIP_CHECK(hipMemsetAsync(g_ks, 0, nb, stream));
    if (n_cov == 0) return 0;
    GS_REQUIRE(scratch != nullptr && scratch_bytes >= 0, "gs_env_shade_bwd_bounded: null scratch");
    const int64_t S2 = 2ll * n_samples_x * n_samples_x;
    int64_t chunk = n_cov;
    if (scratch_bytes < gs_env_shade_scratch_bytes(n_cov, n_samples_x)) {
        chunk = (scratch_bytes - 256) / (S2 * 40) / 64 * 64;
        GS_REQUIRE(chunk >= 64, "gs_env_shade_bwd_bounded: the scratch does not hold the records of 64 pixels");
    }
    for (int64_t off = 0; off < n_cov; off += chunk) {
        const int64_t cnt = n_cov - off < chunk ? n_cov - off : chunk;
        ShadeArgs A{};
        int rc = fill_args(A, bvh, pix + off, cnt, gb_pos /* ro: read by the trace kernel only */, gb_pos, gb_normal, view_pos, gb_kd, gb_ks, light, pdf, rows, cols,
                           Hl, Wl, perms, P, B, H, W, view_offset, view_stride, bsdf, n_samples_x, rnd_seed, shadow_scale,
                           const_cast<uint64_t*>(vis_bits) + off * S2 / 64);
        if (rc) return rc;
        A.g_diff = g_diff; A.g_spec = g_spec; A.g_pos = g_pos; A.g_nrm = g_normal; A.g_kd = g_kd; A.g_ks = g_ks; A.g_light = g_light;
        A.g_tex_stride = g_tex6 ? 6 : 3;
        const int64_t n_rays = cnt * S2;
        A.ray_dk = (float4*)scratch;
        A.ray_contrib = (float*)(A.ray_dk + n_rays);
        hipLaunchKernelGGL(k_shade_samples<false>, dim3((unsigned)gs::cdiv(cnt * A.G, 256)), dim3(256), 0, stream, A);
        launch_saved_grad(A, cnt, n_samples_x, Hl, Wl, stream);
        GS_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int gs_bilateral_fwd(const float* col, const float* nrm, const float* zdz, int64_t B, int64_t H, int64_t W, float sigma, float* out,
                                gs_stream_t stream) {
    if (B * H * W == 0) return 0;
    GS_REQUIRE(col && nrm && zdz && out && sigma > 0.f, "gs_bilateral_fwd: null pointer / sigma <= 0");
    return launch_bilateral<false>(col, nrm, zdz, nullptr, B, H, W, sigma, out, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int gs_bilateral_fwd_masked(const float* col, const float* nrm, const float* zdz, const float* mask, int64_t B, int64_t H, int64_t W,
                                       float sigma, float* out, gs_stream_t stream) {
    if (B * H * W == 0) return 0;
    GS_REQUIRE(col && nrm && zdz && out && sigma > 0.f, "gs_bilateral_fwd_masked: null pointer / sigma <= 0");
    return launch_bilateral<false>(col, nrm, zdz, mask, B, H, W, sigma, out, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int gs_bilateral_fwd_masked2(const float* col_a, const float* col_b, const float* nrm, const float* zdz, const float* mask, int64_t B,
                                        int64_t H, int64_t W, float sigma, float* out_a, float* out_b, gs_stream_t stream) {
    if (B * H * W == 0) return 0;
    GS_REQUIRE(col_a && col_b && nrm && zdz && out_a && out_b && sigma > 0.f, "gs_bilateral_fwd_masked2: null pointer / sigma <= 0");
    return launch_bilateral2<false>(col_a, col_b, nrm, zdz, mask, B, H, W, sigma, out_a, out_b, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int gs_bilateral_bwd_masked2(const float* nrm, const float* zdz, const float* mask, int64_t B, int64_t H, int64_t W, float sigma,
                                        const float* g_out_a, const float* g_out_b, float* g_col_a, float* g_col_b, gs_stream_t stream) {
    if (B * H * W == 0) return 0;
    GS_REQUIRE(nrm && zdz && g_out_a && g_out_b && g_col_a && g_col_b && sigma > 0.f, "gs_bilateral_bwd_masked2: null pointer / sigma <= 0");
    return launch_bilateral2<true>(nullptr, nullptr, nrm, zdz, mask, B, H, W, sigma, nullptr, nullptr, g_out_a, g_out_b, g_col_a, g_col_b,
                                   (hipStream_t)stream);
}

extern "C" int gs_bilateral_bwd(const float* nrm, const float* zdz, int64_t B, int64_t H, int64_t W, float sigma, const float* g_out, float* g_col,
                                gs_stream_t stream) {
    if (B * H * W == 0) return 0;
    GS_REQUIRE(nrm && zdz && g_out && g_col && sigma > 0.f, "gs_bilateral_bwd: null pointer / sigma <= 0");
    return launch_bilateral<true>(nullptr, nrm, zdz, nullptr, B, H, W, sigma, nullptr, g_out, g_col, (hipStream_t)stream);
}

extern "C" int gs_bilateral_bwd_masked(const float* nrm, const float* zdz, const float* mask, int64_t B, int64_t H, int64_t W, float sigma,
                                       const float* g_out, float* g_col, gs_stream_t stream) {
    if (B * H * W == 0) return 0;
    GS_REQUIRE(nrm && zdz && g_out && g_col && sigma > 0.f, "gs_bilateral_bwd_masked: null pointer / sigma <= 0");
    return launch_bilateral<true>(nullptr, nrm, zdz, mask, B, H, W, sigma, nullptr, g_out, g_col, (hipStream_t)stream);
}

// ---- compile-time variants of this file (common.hpp): non-default values announce themselves through gs_build_flags(); switches that give
// wrong results (timing-only ablations) compile only under -DGS_EXPERIMENT
GS_TUNABLE(GS_SAMPLES_WAVES, 4)
GS_TUNABLE(GS_GRAD_WAVES, 3)
GS_TUNABLE(GS_LG_SLICES, 16)
GS_TUNABLE(GS_LIGHT_BINNED, 1)
GS_TUNABLE(TRACE_REFILL, 16)
GS_TUNABLE(TRACE_INNER, 4)
GS_TUNABLE(TRACE_INNER_BREAK, 1)
GS_TUNABLE(TRACE_WAVES, 1)
GS_TUNABLE(GS_BILATERAL_DUAL, 1)
#ifdef GS_EXPERIMENT_NO_LIGHT_GRAD
GS_EXPERIMENT_ONLY(GS_EXPERIMENT_NO_LIGHT_GRAD)
#endif

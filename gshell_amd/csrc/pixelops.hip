// Per-pixel / per-vertex fused ops of the G-Shell render path on gfx950 (fwd + bwd):
//   prepare_shading_normal  replaces ru.prepare_shading_normal (render/renderutils/ops.py:197-229,
//                           CUDA render/renderutils/c_src/normal.cu:18-181)
//   image_loss              replaces ru.image_loss (render/renderutils/ops.py:479-503, c_src/loss.cu:15-210)
//   auto_normals            replaces mesh.auto_normals (render/mesh.py:212-237: 3 scatter_add_ + normalise)
//   texture_linear_clamp    replaces dr.texture(..., filter_mode='linear', boundary_mode='clamp') as used for
//                           the jitter taps (render/render.py:59, :110)
// All are HBM-streaming kernels: one lane per pixel / face / vertex, float3 rows read as 3 dwords
// (12-byte rows cannot be vectorised to dwordx4 without re-laying out the reference's [..,3] tensors).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/gshell_hip.h"
#include "common.hpp"

namespace {

struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ f3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float* p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float sum(f3 a) { return a.x + a.y + a.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ void bwd_cross(f3 a, f3 b, f3& da, f3& db, f3 d) {
    da.x += d.z * b.y - d.y * b.z;
    da.y += d.x * b.z - d.z * b.x;
    da.z += d.y * b.x - d.x * b.y;
    db.x += d.y * a.z - d.z * a.y;
    db.y += d.z * a.x - d.x * a.z;
    db.z += d.x * a.y - d.y * a.x;
}
// reference safeNormalize: v / |v| or 0 (render/renderutils/c_src/vec3f.h:90-94)
__device__ __forceinline__ f3 safe_normalize(f3 v) {
    float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return l > 0.0f ? v * (1.0f / l) : mk(0.f, 0.f, 0.f);
}
__device__ __forceinline__ void bwd_safe_normalize(f3 v, f3& dv, f3 d) {
    float l2 = v.x * v.x + v.y * v.y + v.z * v.z;
    if (l2 > 0.0f) {
        float l = sqrtf(l2);
        float fac = 1.0f / (l2 * l);
        dv.x += (d.x * (v.y * v.y + v.z * v.z) - d.y * (v.x * v.y) - d.z * (v.x * v.z)) * fac;
        dv.y += (d.y * (v.x * v.x + v.z * v.z) - d.x * (v.y * v.x) - d.z * (v.y * v.z)) * fac;
        dv.z += (d.z * (v.x * v.x + v.y * v.y) - d.x * (v.z * v.x) - d.y * (v.z * v.y)) * fac;
    }
}

// ---- prepare_shading_normal ---------------------------------------------------------------------
constexpr float NORMAL_THRESHOLD = 0.1f;

__device__ __forceinline__ f3 perturb_fwd(f3 pn, f3 sn, f3 st, bool opengl, f3& raw, f3& bitng_raw, f3& bitng) {
    bitng_raw = cross(st, sn);
    bitng = safe_normalize(bitng_raw);
    raw = st * pn.x + bitng * ((opengl ? -1.0f : 1.0f) * pn.y) + sn * fmaxf(pn.z, 0.0f);
    return safe_normalize(raw);
}

__device__ __forceinline__ f3 bend_fwd(f3 view, f3 sn, f3 gn) {
    float dp = dot(view, sn);
    float t = fminf(fmaxf(dp / NORMAL_THRESHOLD, 0.0f), 1.0f);
    return gn * (1.0f - t) + sn * t;
}

__device__ __forceinline__ void bend_bwd(f3 view, f3 sn, f3 gn, f3& dview, f3& dsn, f3& dgn, f3 d) {
    float dp = dot(view, sn);
    float t = fminf(fmaxf(dp / NORMAL_THRESHOLD, 0.0f), 1.0f);
    if (dp > NORMAL_THRESHOLD)
        dsn = dsn + d;
    else {
        dgn = dgn + d * (1.0f - t);
        dsn = dsn + d * t;
        float dt = sum(d * (sn - gn));
        float ddp = (dp < 0.0f || dp > NORMAL_THRESHOLD) ? 0.0f : dt / NORMAL_THRESHOLD;
        dview = dview + sn * ddp;
        dsn = dsn + view * ddp;
    }
}

struct PsnArgs {
    const float *pos, *view_pos, *perturbed, *nrm, *tng, *gnrm;
    int64_t n, pix_per_view;  // view_pos index = view_full ? i : i / pix_per_view
    int view_full, two_sided, opengl;
    float* out;
    // bwd
    const float* g_out;
    float *g_pos, *g_view, *g_perturbed, *g_nrm, *g_tng, *g_gnrm;
};

template <bool BWD>
__global__ void __launch_bounds__(256) k_shading_normal(PsnArgs a) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    f3 pos = ld3(a.pos + 3 * i);
    f3 vp = ld3(a.view_pos + 3 * (a.view_full ? i : i / a.pix_per_view));
    f3 pn = a.perturbed ? ld3(a.perturbed + 3 * i) : mk(0.f, 0.f, 1.f);
    f3 sn_raw = ld3(a.nrm + 3 * i), st_raw = ld3(a.tng + 3 * i), gn = ld3(a.gnrm + 3 * i);
    f3 sn = safe_normalize(sn_raw), st = safe_normalize(st_raw);
    f3 view_raw = vp - pos;
    f3 view = safe_normalize(view_raw);
    f3 raw, bitng_raw, bitng;
    f3 shn = perturb_fwd(pn, sn, st, a.opengl, raw, bitng_raw, bitng);
    bool flip = a.two_sided && dot(view, gn) < 0.0f;
    if (!BWD) {
        st3(a.out + 3 * i, flip ? bend_fwd(view, -shn, -gn) : bend_fwd(view, shn, gn));
        return;
    }
    f3 d = ld3(a.g_out + 3 * i);
    f3 dview = mk(0, 0, 0), dshn = mk(0, 0, 0), dgn = mk(0, 0, 0);
    if (flip) {
        bend_bwd(view, -shn, -gn, dview, dshn, dgn, d);
        dshn = -dshn;
        dgn = -dgn;
    } else
        bend_bwd(view, shn, gn, dview, dshn, dgn, d);
    // perturb bwd (normal.cu:28-61)
    f3 draw = mk(0, 0, 0);
    bwd_safe_normalize(raw, draw, dshn);
    f3 dpn = mk(0, 0, 0), dsn = mk(0, 0, 0), dst = mk(0, 0, 0), dbit = mk(0, 0, 0);
    if (pn.z > 0.0f) {
        dsn = dsn + draw * pn.z;
        dpn.z += sum(draw * sn);
    }
    float sgn = a.opengl ? -1.0f : 1.0f;
    dbit = dbit + draw * (sgn * pn.y);
    dpn.y += sgn * sum(draw * bitng);
    dst = dst + draw * pn.x;
    dpn.x += sum(draw * st);
    f3 dbit_raw = mk(0, 0, 0);
    bwd_safe_normalize(bitng_raw, dbit_raw, dbit);
    bwd_cross(st, sn, dst, dsn, dbit_raw);
    f3 dview_raw = mk(0, 0, 0), dsn_raw = mk(0, 0, 0), dst_raw = mk(0, 0, 0);
    bwd_safe_normalize(view_raw, dview_raw, dview);
    bwd_safe_normalize(sn_raw, dsn_raw, dsn);
    bwd_safe_normalize(st_raw, dst_raw, dst);
    if (a.g_pos) st3(a.g_pos + 3 * i, -dview_raw);
    if (a.g_view) st3(a.g_view + 3 * i, dview_raw);
    if (a.g_perturbed) st3(a.g_perturbed + 3 * i, dpn);
    if (a.g_nrm) st3(a.g_nrm + 3 * i, dsn_raw);
    if (a.g_tng) st3(a.g_tng + 3 * i, dst_raw);
    if (a.g_gnrm) st3(a.g_gnrm + 3 * i, dgn);
}

// ---- image loss -------------------------------------------------------------------------------------
enum { LOSS_L1 = 0, LOSS_MSE = 1, LOSS_RELMSE = 2, LOSS_SMAPE = 3 };
enum { TM_NONE = 0, TM_LOG_SRGB = 1 };

__device__ __forceinline__ float fwd_srgb(float x) {
    return x > 0.0031308f ? powf(fmaxf(x, 0.0031308f), 1.0f / 2.4f) * 1.055f - 0.055f : 12.92f * fmaxf(x, 0.0f);
}
__device__ __forceinline__ float bwd_srgb(float x, float d) {
    if (x > 0.0031308f) return d * 0.439583f / powf(x, 0.583333f);
    if (x > 0.0f) return d * 12.92f;
    return 0.0f;
}
__device__ __forceinline__ float tonemap(float x, int tm) {
    float c = fminf(fmaxf(x, 0.0f), 65535.0f);
    return tm == TM_LOG_SRGB ? fwd_srgb(logf(c + 1.0f)) : c;
}
// Chain rule through the log-sRGB tonemapper exactly as bwdTonemapLogSRGB does it (loss.cu:50-67): only on the OPEN interval.
__device__ __forceinline__ float tonemap_log_srgb_bwd(float x, float d) {
    if (!(x > 0.0f && x < 65535.0f)) return 0.0f;
    return bwd_srgb(logf(x + 1.0f), d) * (1.0f / (x + 1.0f));
}
__device__ __forceinline__ float sgnf(float x) { return x == 0.0f ? 0.0f : (x < 0.0f ? -1.0f : 1.0f); }

__device__ __forceinline__ float loss_fwd(float a, float b, int loss) {
    switch (loss) {
        case LOSS_MSE: return (a - b) * (a - b);
        case LOSS_RELMSE: return (a - b) * (a - b) / (a * a + b * b + 0.1f);
        case LOSS_SMAPE: return fabsf(a - b) / (a + b + 0.01f);
        default: return fabsf(a - b);
    }
}
__device__ __forceinline__ void loss_bwd(float a, float b, int loss, float d, float& da, float& db) {
    switch (loss) {
        case LOSS_MSE:
            da = d * 2.0f * (a - b);
            db = -da;
            break;
        case LOSS_RELMSE: {
            float den = b * b + a * a + 0.1f;
            da = d * 2.0f * (a - b) * (b * (b + a) + 0.1f) / (den * den);
            db = -d * 2.0f * (a - b) * (a * (b + a) + 0.1f) / (den * den);
            break;
        }
        case LOSS_SMAPE: {
            float den = b + a + 0.01f;
            da = d * sgnf(a - b) * (2.0f * b + 0.01f) / (den * den);
            db = -d * sgnf(a - b) * (2.0f * a + 0.01f) / (den * den);
            break;
        }
        default:
            da = d * sgnf(a - b);
            db = -da;
    }
}

// d loss / d (img, target) of one element as imgLossBwdKernel computes it (loss.cu:137-209) -- which is NOT the derivative of the
// forward kernel outside (0, 65535): the loss derivative is re-evaluated on the UNCLAMPED inputs (tonemapped without the
// forward's clamp, :157-163), the log-sRGB chain rule applies on the open interval only, and finally an input that is itself
// <= 0 or >= 65535 gets a zero gradient (:197-202) while the OTHER input keeps the derivative formed from the unclamped pair.
// Pinned by tests/golden/ref_image_loss.npz (the reference kernel compiled for the host).
__device__ __forceinline__ void image_loss_bwd_elem(float x, float y, int loss, int tm, float d, float& gx, float& gy) {
    float a = x, b = y;
    if (tm == TM_LOG_SRGB) {
        a = fwd_srgb(logf(x + 1.0f));
        b = fwd_srgb(logf(y + 1.0f));
    }
    float da, db;
    loss_bwd(a, b, loss, d, da, db);
    if (tm == TM_LOG_SRGB) {
        da = tonemap_log_srgb_bwd(x, da);
        db = tonemap_log_srgb_bwd(y, db);
    }
    gx = (x <= 0.0f || x >= 65535.0f) ? 0.0f : da;
    gy = (y <= 0.0f || y >= 65535.0f) ? 0.0f : db;
}

// partial[blockIdx] = sum over the block's elements (wave64 shuffle reduction, then 4 waves through LDS)
__global__ void __launch_bounds__(256) k_image_loss_fwd(const float* __restrict__ img, const float* __restrict__ target, int64_t n, int loss,
                                                        int tm, float* __restrict__ partial) {
    int64_t i0 = (int64_t)blockIdx.x * 1024;
    float acc = 0.f;
    for (int k = 0; k < 4; ++k) {
        int64_t i = i0 + k * 256 + threadIdx.x;
        if (i < n) acc += loss_fwd(tonemap(img[i], tm), tonemap(target[i], tm), loss);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

__global__ void __launch_bounds__(256) k_image_loss_bwd(const float* __restrict__ img, const float* __restrict__ target, int64_t n, int loss,
                                                        int tm, const float* __restrict__ g_scalar, float scale, float* __restrict__ g_img,
                                                        float* __restrict__ g_target) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float d = g_scalar[0] * scale;
    float x = img[i], y = target[i];
    float da, db;
    image_loss_bwd_elem(x, y, loss, tm, d, da, db);
    if (g_img) g_img[i] = da;
    if (g_target) g_target[i] = db;
}

// ---- softplus with first and second derivative -------------------------------------------------------------
// torch.nn.Softplus(beta, threshold = 20) as three elementwise kernels: value, input gradient, and the gradient OF the
// input gradient (the eikonal term differentiates the SDF network's input gradient again: torch expands that double
// backward into ~9 launches per layer over [50 k, 256] tensors).  Same formulas as ATen (Activation.cpp softplus_backward,
// derivatives.yaml softplus_double_backward): z = exp(beta x); s = z / (z + 1).
__global__ void __launch_bounds__(256) k_softplus(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ gg, int64_t n,
                                                  float beta, int mode, float* __restrict__ o0, float* __restrict__ o1) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float xv = x[i], bx = xv * beta;
    const bool lin = bx > 20.0f;
    if (mode == 0) {                                   // y
        o0[i] = lin ? xv : __logf(1.0f + __expf(bx)) / beta;      // hardware exp / log: |error| <= 1e-9 on the value (as in mlp.hip)
        return;
    }
    const float z = __expf(bx), s = z / (z + 1.0f);
    if (mode == 1) {                                   // g_x = g * s
        o0[i] = lin ? g[i] : g[i] * s;
        return;
    }
    const float gv = g[i], ggv = gg[i];                // mode 2: d/dg and d/dx of (g * s) contracted with gg
    if (o0) o0[i] = lin ? ggv : ggv * s;
    if (o1) o1[i] = lin ? 0.0f : ggv * gv * beta * z / ((z + 1.0f) * (z + 1.0f));
}

// ---- whole-frame loss / regulariser sums --------------------------------------------------------------
// One pass over the stacked, antialiased frame buffers [B*H*W, C] (render.render_mesh keeps every buffer as a channel
// slice of ONE tensor) producing the nine pixel sums behind the alpha MSE, the two mSDF image terms
// (gshell_tets_geometry.py:280-285 of the reference), the monochrome-lighting prior (regularizer.py:34-41) and the
// material / normal smoothness terms (regularizer.py:21-31).  As torch ops these are ~230 launches over 1 M pixels.
// The scalar combination (means, the specular / diffuse ratio, lambdas) stays in torch on the nine numbers.
enum { FS_ALPHA = 0, FS_MSDF0, FS_MSDF1, FS_ERR, FS_SPEC, FS_DIFF, FS_KD, FS_KS, FS_NRM, FS_IMG, FS_COUNT };      // FS_IMG: the *_img entry points only

struct FrameOffs {   // channel offset of each buffer inside a pixel record, -1 = absent
    int shaded, msdf, diff, spec, kdg, ksg, nrmg;
};

// regularizer._log_srgb with torch's gradient conventions (closed clamp interval, 12.92 slope at and below the knee)
__device__ __forceinline__ float fl_logsrgb(float x, float* grad) {
    const float c = fminf(fmaxf(x, 0.0f), 65535.0f);
    const float L = logf(c + 1.0f);
    float f, gs;
    if (L <= 0.0031308f) {
        f = L * 12.92f;
        gs = 12.92f;
    } else {
        const float b = fmaxf(L, 0.0031308f);
        f = powf(b, 1.0f / 2.4f) * 1.055f - 0.055f;
        gs = (1.055f / 2.4f) * powf(b, 1.0f / 2.4f - 1.0f);
    }
    if (grad) *grad = (x >= 0.0f && x <= 65535.0f) ? gs / (c + 1.0f) : 0.0f;
    return f;
}

template <bool BWD>
__global__ void __launch_bounds__(256) k_frame_sums(const float* __restrict__ st, const float* __restrict__ ref, int64_t n, int C, FrameOffs o,
                                                    float* __restrict__ partial, const float* __restrict__ g9, float* __restrict__ g_st, int n_sums,
                                                    int img_loss, int img_tm) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // Both directions read the frame tile as whole lines into LDS ([256 pixels][C], 16-byte pieces where the tile allows) and every thread walks
    // ITS record there (stride C floats: conflict-free for odd C).  Backward: the thread first gathers the channels it needs, then zeroes its own
    // record and assembles the gradient channels IN PLACE; the tile leaves as whole lines.  (A thread reading its 180-byte record from HBM made
    // every load instruction touch 64 lines: 0.12 ms for 188 MB; writing it that way: 0.245 ms.)
    extern __shared__ __attribute__((aligned(16))) float fs_tile[];          // [256][C]
    float acc[FS_COUNT];
#pragma unroll
    for (int k = 0; k < FS_COUNT; ++k) acc[k] = 0.0f;
    const int64_t p0 = (int64_t)blockIdx.x * 256, cnt = min((int64_t)256, n - p0) * C;
    const bool vec4 = (cnt & 3) == 0 && ((reinterpret_cast<uintptr_t>(st + p0 * C) | (BWD ? reinterpret_cast<uintptr_t>(g_st + p0 * C) : 0)) & 15) == 0;
    if (vec4) {
        const float4* src = reinterpret_cast<const float4*>(st + p0 * C);
        float4* dst = reinterpret_cast<float4*>(fs_tile);
        for (int64_t q = threadIdx.x; q < cnt / 4; q += 256) dst[q] = src[q];
    } else {
        for (int64_t q = threadIdx.x; q < cnt; q += 256) fs_tile[q] = st[p0 * C + q];
    }
    __syncthreads();
    if (i < n) {
        float* rec = fs_tile + threadIdx.x * C;
        // the channels this thread reads (absent buffers: zeros, never used)
        float sh[4] = {0.f, 0.f, 0.f, 0.f}, msd = 0.f, dfl[3] = {0.f, 0.f, 0.f}, spl[3] = {0.f, 0.f, 0.f}, kdv[4] = {0.f, 0.f, 0.f, 0.f},
              ksv[4] = {0.f, 0.f, 0.f, 0.f}, nrv[4] = {0.f, 0.f, 0.f, 0.f};
        if (o.shaded >= 0)
            for (int c = 0; c < 4; ++c) sh[c] = rec[o.shaded + c];
        if (o.msdf >= 0) msd = rec[o.msdf];
        if (o.diff >= 0 && o.spec >= 0)
            for (int c = 0; c < 3; ++c) { dfl[c] = rec[o.diff + c]; spl[c] = rec[o.spec + c]; }
        if (o.kdg >= 0)
            for (int c = 0; c < 4; ++c) kdv[c] = rec[o.kdg + c];
        if (o.ksg >= 0)
            for (int c = 0; c < 4; ++c) ksv[c] = rec[o.ksg + c];
        if (o.nrmg >= 0)
            for (int c = 0; c < 4; ++c) nrv[c] = rec[o.nrmg + c];
        float* g = BWD ? rec : nullptr;
        if (BWD)
            for (int c = 0; c < C; ++c) rec[c] = 0.0f;          // this thread's record only: no barrier needed
        const float m = ref[4 * i + 3];
        if (o.shaded >= 0) {
            const float a = sh[3];
            acc[FS_ALPHA] = (a - m) * (a - m);
            if (BWD) g[o.shaded + 3] = g9[FS_ALPHA] * 2.0f * (a - m);
            if (img_loss >= 0) {
                // the colour term of the image loss (gshell_tets_geometry.py:277: loss_fn(shaded rgb * m, reference rgb * m), the
                // element sum of renderutils' image_loss, loss.cu) -- so that the frame has ONE consumer and ONE gradient tensor
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float x = sh[c] * m, y = ref[4 * i + c] * m;
                    const float tx = tonemap(x, img_tm), ty = tonemap(y, img_tm);
                    acc[FS_IMG] += loss_fwd(tx, ty, img_loss);
                    if (BWD) {
                        float da, db;
                        image_loss_bwd_elem(x, y, img_loss, img_tm, g9[FS_IMG], da, db);
                        g[o.shaded + c] = da * m;
                    }
                }
            }
        }
        if (o.msdf >= 0) {
            const float x = msd;
            const float k0 = m == 0.0f ? 1.0f : 0.0f, k1 = m == 1.0f ? 1.0f : 0.0f;
            acc[FS_MSDF0] = fabsf(fmaxf(x, 0.0f) * k0);                      // | clamp(x, min=0) * [m == 0] - 0 |
            acc[FS_MSDF1] = fabsf(fminf(x, 0.0f) * k1 - 1.0f);               // | clamp(x, max=0) * [m == 1] - 1 |
            if (BWD) g[o.msdf] = g9[FS_MSDF0] * (x > 0.0f ? k0 : 0.0f) - g9[FS_MSDF1] * (x <= 0.0f ? k1 : 0.0f);
        }
        if (o.diff >= 0 && o.spec >= 0) {
            const float dl = (dfl[0] + dfl[1] + dfl[2]) / 3.0f;
            const float sl = (spl[0] + spl[1] + spl[2]) / 3.0f;
            const float v = fmaxf(fmaxf(ref[4 * i], ref[4 * i + 1]), ref[4 * i + 2]);
            float gx;
            const float t1 = fl_logsrgb((dl + sl) * m, &gx), t2 = fl_logsrgb(v * m, nullptr);
            acc[FS_ERR] = fabsf(t1 - t2);
            acc[FS_SPEC] = sl;
            acc[FS_DIFF] = dl;
            if (BWD) {
                const float ge = g9[FS_ERR] * sgnf(t1 - t2) * gx * m;
                const float gd = (ge + g9[FS_DIFF]) / 3.0f, gs = (ge + g9[FS_SPEC]) / 3.0f;
                g[o.diff] = gd; g[o.diff + 1] = gd; g[o.diff + 2] = gd;
                g[o.spec] = gs; g[o.spec + 1] = gs; g[o.spec + 2] = gs;
            }
        }
        if (o.kdg >= 0) {
            const float s3 = (kdv[0] + kdv[1] + kdv[2]) / 3.0f, w = kdv[3];
            acc[FS_KD] = s3 * w;
            if (BWD) {
                const float gk = g9[FS_KD] * w / 3.0f;
                g[o.kdg] = gk; g[o.kdg + 1] = gk; g[o.kdg + 2] = gk;
                g[o.kdg + 3] = g9[FS_KD] * s3;
            }
        }
        if (o.ksg >= 0) {
            const float s3 = ksv[0] + ksv[1] + ksv[2], w = ksv[3];
            acc[FS_KS] = s3 * w;
            if (BWD) {
                const float gk = g9[FS_KS] * w;
                g[o.ksg] = gk; g[o.ksg + 1] = gk; g[o.ksg + 2] = gk;
                g[o.ksg + 3] = g9[FS_KS] * s3;
            }
        }
        if (o.nrmg >= 0) {
            const float s3 = nrv[0] + nrv[1] + nrv[2], w = nrv[3];
            acc[FS_NRM] = s3 * w;
            if (BWD) {
                const float gk = g9[FS_NRM] * w;
                g[o.nrmg] = gk; g[o.nrmg + 1] = gk; g[o.nrmg + 2] = gk;
                g[o.nrmg + 3] = g9[FS_NRM] * s3;
            }
        }
    }
    if (BWD) {
        __syncthreads();
        if (vec4) {
            const float4* src = reinterpret_cast<const float4*>(fs_tile);
            float4* dst = reinterpret_cast<float4*>(g_st + p0 * C);
            for (int64_t q = threadIdx.x; q < cnt / 4; q += 256) dst[q] = src[q];
        } else {
            for (int64_t q = threadIdx.x; q < cnt; q += 256) g_st[p0 * C + q] = fs_tile[q];
        }
        return;
    }
    __shared__ float ws[4][FS_COUNT];
#pragma unroll
    for (int k = 0; k < FS_COUNT; ++k) {
        float v = acc[k];
        for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < n_sums) partial[(int64_t)blockIdx.x * n_sums + threadIdx.x] = (ws[0][threadIdx.x] + ws[1][threadIdx.x]) + (ws[2][threadIdx.x] + ws[3][threadIdx.x]);
}

// ---- auto normals -----------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_face_normals_scatter(const float* __restrict__ v, const int32_t* __restrict__ tri, int64_t T,
                                                              float* __restrict__ acc) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    int64_t i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
    f3 v0 = ld3(v + 3 * i0), v1 = ld3(v + 3 * i1), v2 = ld3(v + 3 * i2);
    f3 n = cross(v1 - v0, v2 - v0);
    int64_t idx[3] = {i0, i1, i2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        atomicAdd(&acc[3 * idx[k] + 0], n.x);
        atomicAdd(&acc[3 * idx[k] + 1], n.y);
        atomicAdd(&acc[3 * idx[k] + 2], n.z);
    }
}

// v_nrm = safe_normalize(where(dot > 1e-20, acc, (0,0,1)));  util.safe_normalize = x / sqrt(max(dot, 1e-20))
__global__ void __launch_bounds__(256) k_vertex_normalize(const float* __restrict__ acc, int64_t V, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    f3 a = ld3(acc + 3 * i);
    float d = dot(a, a);
    if (!(d > 1e-20f)) {
        a = mk(0.f, 0.f, 1.f);
        d = 1.0f;
    }
    float l = sqrtf(fmaxf(d, 1e-20f));
    st3(out + 3 * i, mk(a.x / l, a.y / l, a.z / l));
}

__global__ void __launch_bounds__(256) k_vertex_normalize_bwd(const float* __restrict__ acc, int64_t V, const float* __restrict__ g_out,
                                                              float* __restrict__ g_acc) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    f3 a = ld3(acc + 3 * i);
    float d = dot(a, a);
    f3 g = mk(0, 0, 0);
    if (d > 1e-20f) {
        f3 go = ld3(g_out + 3 * i);
        float l = sqrtf(d);
        float il = 1.0f / l;
        float proj = dot(go, a) * il * il * il;   // d/da (a/|a|) = (I - a a^T / |a|^2) / |a|
        g = go * il - a * proj;
    }
    st3(g_acc + 3 * i, g);
}

__global__ void __launch_bounds__(256) k_face_normals_bwd(const float* __restrict__ v, const int32_t* __restrict__ tri, int64_t T,
                                                          const float* __restrict__ g_acc, float* __restrict__ g_v) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    int64_t i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
    f3 v0 = ld3(v + 3 * i0), v1 = ld3(v + 3 * i1), v2 = ld3(v + 3 * i2);
    f3 gn = ld3(g_acc + 3 * i0) + ld3(g_acc + 3 * i1) + ld3(g_acc + 3 * i2);
    f3 da = mk(0, 0, 0), db = mk(0, 0, 0);
    bwd_cross(v1 - v0, v2 - v0, da, db, gn);
    f3 d0 = -(da + db);
    atomicAdd(&g_v[3 * i0 + 0], d0.x);
    atomicAdd(&g_v[3 * i0 + 1], d0.y);
    atomicAdd(&g_v[3 * i0 + 2], d0.z);
    atomicAdd(&g_v[3 * i1 + 0], da.x);
    atomicAdd(&g_v[3 * i1 + 1], da.y);
    atomicAdd(&g_v[3 * i1 + 2], da.z);
    atomicAdd(&g_v[3 * i2 + 0], db.x);
    atomicAdd(&g_v[3 * i2 + 1], db.y);
    atomicAdd(&g_v[3 * i2 + 2], db.z);
}

// ---- depth + depth-slope guide of the denoiser (reference render/render.py:273-279, ~14 ATen launches on one-channel images) ---------
// out[p] = (z0, |z1 - z0|),  z0 = max(c.z, eps) / max(c.w, eps),  z1 = max(c.z + |dd[2]|, eps) / max(c.w + |dd[3]|, eps),  with c the
// interpolated clip position [n,4] and dd its screen-space derivative image [n,8] (the reference reads channels 2 and 3 of it as is).
// torch.clamp(min) propagates NaN: so does the explicit form below.  Same operations in the same order: bit-identical.
__device__ __forceinline__ float clamp_min_nan(float x, float lo) { return x != x ? x : fmaxf(x, lo); }
__global__ void __launch_bounds__(256) k_depth_zgrad(const float4* __restrict__ clip, const float4* __restrict__ dd, int64_t n, float eps,
                                                     float2* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 c = clip[i];
    const float4 d = dd[2 * i];                      // channels 0..3 of the 8
    const float z0 = clamp_min_nan(c.z, eps) / clamp_min_nan(c.w, eps);
    const float z1 = clamp_min_nan(c.z + fabsf(d.z), eps) / clamp_min_nan(c.w + fabsf(d.w), eps);
    out[i] = make_float2(z0, fabsf(z1 - z0));
}

// ---- bilinear texture tap, clamp addressing ----------------------------------------------------------
// texel centres at (i + .5) / W ;  tex [B,H,W,C], uv [B,h,w,2] -> out [B,h,w,C]
template <bool BWD>
__global__ void __launch_bounds__(256) k_texture_linear(const float* __restrict__ tex, int64_t B, int H, int W, int C,
                                                        const float* __restrict__ uv, int64_t n_per_view, float* __restrict__ out,
                                                        const float* __restrict__ g_out, float* __restrict__ g_tex) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * n_per_view) return;
    int64_t b = i / n_per_view;
    float x = uv[2 * i] * (float)W - 0.5f, y = uv[2 * i + 1] * (float)H - 0.5f;
    float xf = floorf(x), yf = floorf(y);
    float fx = x - xf, fy = y - yf;
    int x0 = min(max((int)xf, 0), W - 1), x1 = min(max((int)xf + 1, 0), W - 1);
    int y0 = min(max((int)yf, 0), H - 1), y1 = min(max((int)yf + 1, 0), H - 1);
    float w00 = (1.0f - fx) * (1.0f - fy), w10 = fx * (1.0f - fy), w01 = (1.0f - fx) * fy, w11 = fx * fy;
    int64_t base = b * (int64_t)H * W;
    int64_t o00 = (base + (int64_t)y0 * W + x0) * C, o10 = (base + (int64_t)y0 * W + x1) * C;
    int64_t o01 = (base + (int64_t)y1 * W + x0) * C, o11 = (base + (int64_t)y1 * W + x1) * C;
    for (int c = 0; c < C; ++c) {
        if (!BWD)
            out[i * C + c] = tex[o00 + c] * w00 + tex[o10 + c] * w10 + tex[o01 + c] * w01 + tex[o11 + c] * w11;
        else {
            float g = g_out[i * C + c];
            if (g != 0.0f) {
                atomicAdd(&g_tex[o00 + c], g * w00);
                atomicAdd(&g_tex[o10 + c], g * w10);
                atomicAdd(&g_tex[o01 + c], g * w01);
                atomicAdd(&g_tex[o11 + c], g * w11);
            }
        }
    }
}

// ---- shade() buffer assembly + background composite in one pass --------------------------------------------------
// Replaces the torch tail of render.shade / render_mesh (reference render/render.py:74 ks_grad mask, :105-112 normal
// regulariser, :160-186 kd * (1 - metalness), shaded = diffuse * kd + specular and the ~11 torch.cat((buffer, alpha)) of the
// buffer dictionary, :352-359 + :417-433 composite of every buffer over its background, and the division col / weight that
// follows the bilateral denoiser, optixutils/ops.py:145-147): ~60 forward and ~80 backward ATen launches over 4 x 512^2
// frames.  One thread per pixel writes the 45 composited channels
//   shaded(4) z_grad(4) normal(4) geometric_normal(4) kd(4) ks(4) kd_grad(4) ks_grad(4) normal_grad(4) diffuse_light(4)
//   specular_light(4) [msdf_image(1)]
// staged through LDS so that the [P,45] frame is written in full lines.  alpha is 1 (3-channel kd), so the lerp of the
// composite selects foreground where a triangle covers the pixel and background elsewhere, exactly as torch.lerp with a in {0,1}.
struct AssembleArgs {
    const float *rast, *tex, *texj, *n_in, *n_jit, *mask_tap, *n_shade, *n_geo, *depth, *dcw, *scw, *msdf, *bg;
    int64_t P, HW;
    int bg_views;      // 1: one background image broadcast over the views
    int cw_channels;   // 4: (sum w c, sum w) from the bilateral filter; 3: raw radiance (no denoiser)
    int C;             // 44, or 45 with the mSDF image
    float* out;
    const float* g_out;
    float *g_tex, *g_texj, *g_n_in, *g_n_jit, *g_n_shade, *g_n_geo, *g_dcw, *g_scw, *g_msdf;
};

constexpr int AS_MAXC = 45;

template <bool BWD>
__global__ void __launch_bounds__(256) k_shade_assemble(AssembleArgs A) {
    __shared__ float tile[256 * AS_MAXC];
    const int t = threadIdx.x, C = A.C;
    const int64_t p0 = (int64_t)blockIdx.x * 256, p = p0 + t;
    const int64_t n_here = min((int64_t)256, A.P - p0);
    if (BWD) {      // coalesced load of the upstream gradient tile
        for (int64_t i = t; i < n_here * C; i += 256) tile[i] = A.g_out[p0 * C + i];
        __syncthreads();
    }
    if (p < A.P) {
        float* o = tile + t * C;
        const bool cover = A.rast[4 * p + 3] > 0.0f;
        if (!BWD) {
#pragma unroll 5
            for (int c = 0; c < C; ++c) o[c] = 0.0f;
        }
        if (!cover) {
            if (!BWD) {
                const float* b = A.bg + 3 * (A.bg_views == 1 ? p % A.HW : p);
                o[0] = b[0]; o[1] = b[1]; o[2] = b[2];
            } else {
                for (int c = 0; c < 6; ++c) { A.g_tex[6 * p + c] = 0.f; A.g_texj[6 * p + c] = 0.f; }
                for (int c = 0; c < 3; ++c) { A.g_n_in[3 * p + c] = 0.f; A.g_n_jit[3 * p + c] = 0.f; A.g_n_shade[3 * p + c] = 0.f; A.g_n_geo[3 * p + c] = 0.f; }
                for (int c = 0; c < A.cw_channels; ++c) { A.g_dcw[A.cw_channels * p + c] = 0.f; A.g_scw[A.cw_channels * p + c] = 0.f; }
                if (A.g_msdf) A.g_msdf[p] = 0.f;
            }
        } else {
            float kd[3], ks[3], kdj[3], ksj[3], ni[3], nj[3], dif[3], spc[3];
            for (int c = 0; c < 3; ++c) {
                kd[c] = A.tex[6 * p + c]; ks[c] = A.tex[6 * p + 3 + c];
                kdj[c] = A.texj[6 * p + c]; ksj[c] = A.texj[6 * p + 3 + c];
                ni[c] = A.n_in[3 * p + c]; nj[c] = A.n_jit[3 * p + c];
            }
            const int cw = A.cw_channels;
            const float dw = cw == 4 ? A.dcw[4 * p + 3] : 1.0f, sw = cw == 4 ? A.scw[4 * p + 3] : 1.0f;
            for (int c = 0; c < 3; ++c) {
                dif[c] = cw == 4 ? A.dcw[4 * p + c] / dw : A.dcw[3 * p + c];
                spc[c] = cw == 4 ? A.scw[4 * p + c] / sw : A.scw[3 * p + c];
            }
            const float om = 1.0f - ks[2];                 // 1 - metalness
            const float gw = A.mask_tap[p];                // mask (= 1 here) * jittered mask tap
            const float ksm[3] = {0.0f, 1.0f, 1.0f};       // the o-component of ks is left out of its regulariser
            if (!BWD) {
                for (int c = 0; c < 3; ++c) {
                    const float kdm = kd[c] * om;
                    o[c] = dif[c] * kdm + spc[c];
                    o[8 + c] = A.n_shade[3 * p + c];
                    o[12 + c] = A.n_geo[3 * p + c];
                    o[16 + c] = kdm;
                    o[20 + c] = ks[c];
                    o[24 + c] = fabsf(kdj[c] - kd[c]);
                    o[28 + c] = fabsf(ksj[c] - ks[c]) * ksm[c];
                    o[32 + c] = fabsf(nj[c] - ni[c]) * gw;
                    o[36 + c] = dif[c];
                    o[40 + c] = spc[c];
                }
                o[4] = A.depth[2 * p]; o[5] = A.depth[2 * p + 1]; o[6] = 0.0f;
                for (int b = 0; b < 11; ++b) o[4 * b + 3] = 1.0f;
                if (C > 44) o[44] = A.msdf[p];
            } else {
                const float* g = o;
                float g_ks2 = 0.0f, g_dw = 0.0f, g_sw = 0.0f;
                for (int c = 0; c < 3; ++c) {
                    const float kdm = kd[c] * om;
                    const float g_kdm = g[c] * dif[c] + g[16 + c];
                    const float g_dif = g[c] * kdm + g[36 + c];
                    const float g_spc = g[c] + g[40 + c];
                    const float s_kd = sgnf(kdj[c] - kd[c]) * g[24 + c];
                    const float s_ks = sgnf(ksj[c] - ks[c]) * ksm[c] * g[28 + c];
                    const float s_n = sgnf(nj[c] - ni[c]) * gw * g[32 + c];
                    A.g_tex[6 * p + c] = g_kdm * om - s_kd;
                    g_ks2 -= g_kdm * kd[c];
                    A.g_tex[6 * p + 3 + c] = g[20 + c] - s_ks;          // metalness term added below
                    A.g_texj[6 * p + c] = s_kd;
                    A.g_texj[6 * p + 3 + c] = s_ks;
                    A.g_n_in[3 * p + c] = -s_n;
                    A.g_n_jit[3 * p + c] = s_n;
                    A.g_n_shade[3 * p + c] = g[8 + c];
                    A.g_n_geo[3 * p + c] = g[12 + c];
                    if (cw == 4) {
                        A.g_dcw[4 * p + c] = g_dif / dw;
                        A.g_scw[4 * p + c] = g_spc / sw;
                        g_dw -= g_dif * dif[c] / dw;
                        g_sw -= g_spc * spc[c] / sw;
                    } else {
                        A.g_dcw[3 * p + c] = g_dif;
                        A.g_scw[3 * p + c] = g_spc;
                    }
                }
                A.g_tex[6 * p + 5] += g_ks2;
                if (cw == 4) { A.g_dcw[4 * p + 3] = g_dw; A.g_scw[4 * p + 3] = g_sw; }
                if (A.g_msdf) A.g_msdf[p] = C > 44 ? g[44] : 0.0f;
            }
        }
    }
    if (!BWD) {
        __syncthreads();
        for (int64_t i = t; i < n_here * C; i += 256) A.out[p0 * C + i] = tile[i];
    }
}

}  // namespace

static int assemble_check(const AssembleArgs& A) {
    GS_REQUIRE(A.rast && A.tex && A.texj && A.n_in && A.n_jit && A.mask_tap && A.n_shade && A.n_geo && A.depth && A.dcw && A.scw && A.bg,
               "gs_shade_assemble: null pointer");
    GS_REQUIRE((A.C == 44 && !A.msdf) || (A.C == 45 && A.msdf), "gs_shade_assemble: C must be 44, or 45 with an mSDF image");
    GS_REQUIRE(A.cw_channels == 3 || A.cw_channels == 4, "gs_shade_assemble: radiance inputs have 3 (raw) or 4 (filtered sum, weight) channels");
    return 0;
}

extern "C" int gs_shade_assemble_fwd(const float* rast, const float* tex, const float* tex_jitter, const float* n_interp, const float* n_jitter,
                                     const float* mask_tap, const float* n_shade, const float* n_geo, const float* depth, const float* diffuse,
                                     const float* specular, int cw_channels, const float* msdf_image, const float* background, int bg_views,
                                     int64_t B, int64_t H, int64_t W, float* out, gs_stream_t stream) {
    if (B * H * W == 0) return 0;
    AssembleArgs A{};
    A.rast = rast; A.tex = tex; A.texj = tex_jitter; A.n_in = n_interp; A.n_jit = n_jitter; A.mask_tap = mask_tap; A.n_shade = n_shade; A.n_geo = n_geo;
    A.depth = depth; A.dcw = diffuse; A.scw = specular; A.msdf = msdf_image; A.bg = background; A.P = B * H * W; A.HW = H * W; A.bg_views = bg_views;
    A.cw_channels = cw_channels; A.C = msdf_image ? 45 : 44; A.out = out;
    if (int rc = assemble_check(A)) return rc;
    GS_REQUIRE(out && (bg_views == 1 || bg_views == B), "gs_shade_assemble_fwd: null output / background views");
    hipLaunchKernelGGL(k_shade_assemble<false>, dim3((unsigned)gs::cdiv(A.P, 256)), dim3(256), 0, (hipStream_t)stream, A);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_shade_assemble_bwd(const float* rast, const float* tex, const float* tex_jitter, const float* n_interp, const float* n_jitter,
                                     const float* mask_tap, const float* n_shade, const float* n_geo, const float* depth, const float* diffuse,
                                     const float* specular, int cw_channels, const float* msdf_image, int64_t B, int64_t H, int64_t W,
                                     const float* g_out, float* g_tex, float* g_tex_jitter, float* g_n_interp, float* g_n_jitter, float* g_n_shade,
                                     float* g_n_geo, float* g_diffuse, float* g_specular, float* g_msdf_image, gs_stream_t stream) {
    if (B * H * W == 0) return 0;
    AssembleArgs A{};
    A.rast = rast; A.tex = tex; A.texj = tex_jitter; A.n_in = n_interp; A.n_jit = n_jitter; A.mask_tap = mask_tap; A.n_shade = n_shade; A.n_geo = n_geo;
    A.depth = depth; A.dcw = diffuse; A.scw = specular; A.msdf = msdf_image; A.bg = rast /* unused */; A.P = B * H * W; A.HW = H * W; A.bg_views = 1;
    A.cw_channels = cw_channels; A.C = msdf_image ? 45 : 44; A.g_out = g_out;
    A.g_tex = g_tex; A.g_texj = g_tex_jitter; A.g_n_in = g_n_interp; A.g_n_jit = g_n_jitter; A.g_n_shade = g_n_shade; A.g_n_geo = g_n_geo;
    A.g_dcw = g_diffuse; A.g_scw = g_specular; A.g_msdf = g_msdf_image;
    if (int rc = assemble_check(A)) return rc;
    GS_REQUIRE(g_out && g_tex && g_tex_jitter && g_n_interp && g_n_jitter && g_n_shade && g_n_geo && g_diffuse && g_specular && (!msdf_image || g_msdf_image),
               "gs_shade_assemble_bwd: null pointer");
    hipLaunchKernelGGL(k_shade_assemble<true>, dim3((unsigned)gs::cdiv(A.P, 256)), dim3(256), 0, (hipStream_t)stream, A);
    GS_LAUNCH_CHECK();
    return 0;
}

namespace {
}  // namespace

extern "C" int gs_shading_normal_fwd(const float* pos, const float* view_pos, int view_full, const float* perturbed_nrm, const float* smooth_nrm,
                                     const float* smooth_tng, const float* geom_nrm, int64_t B, int64_t pix_per_view, int two_sided, int opengl,
                                     float* out, gs_stream_t stream) {
    int64_t n = B * pix_per_view;
    if (n == 0) return 0;
    GS_REQUIRE(pos && view_pos && smooth_nrm && smooth_tng && geom_nrm && out, "gs_shading_normal_fwd: null pointer");
    PsnArgs a{};
    a.pos = pos; a.view_pos = view_pos; a.perturbed = perturbed_nrm; a.nrm = smooth_nrm; a.tng = smooth_tng; a.gnrm = geom_nrm;
    a.n = n; a.pix_per_view = pix_per_view; a.view_full = view_full; a.two_sided = two_sided; a.opengl = opengl; a.out = out;
    hipLaunchKernelGGL(k_shading_normal<false>, dim3((unsigned)gs::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_shading_normal_bwd(const float* pos, const float* view_pos, int view_full, const float* perturbed_nrm, const float* smooth_nrm,
                                     const float* smooth_tng, const float* geom_nrm, int64_t B, int64_t pix_per_view, int two_sided, int opengl,
                                     const float* g_out, float* g_pos, float* g_view_pos_full, float* g_perturbed_nrm, float* g_smooth_nrm,
                                     float* g_smooth_tng, float* g_geom_nrm, gs_stream_t stream) {
    int64_t n = B * pix_per_view;
    if (n == 0) return 0;
    GS_REQUIRE(pos && view_pos && smooth_nrm && smooth_tng && geom_nrm && g_out, "gs_shading_normal_bwd: null pointer");
    PsnArgs a{};
    a.pos = pos; a.view_pos = view_pos; a.perturbed = perturbed_nrm; a.nrm = smooth_nrm; a.tng = smooth_tng; a.gnrm = geom_nrm;
    a.n = n; a.pix_per_view = pix_per_view; a.view_full = view_full; a.two_sided = two_sided; a.opengl = opengl;
    a.g_out = g_out; a.g_pos = g_pos; a.g_view = g_view_pos_full; a.g_perturbed = g_perturbed_nrm; a.g_nrm = g_smooth_nrm;
    a.g_tng = g_smooth_tng; a.g_gnrm = g_geom_nrm;
    hipLaunchKernelGGL(k_shading_normal<true>, dim3((unsigned)gs::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t gs_image_loss_partials(int64_t n) { return gs::cdiv(std::max<int64_t>(n, 1), 1024); }

extern "C" int gs_image_loss_fwd(const float* img, const float* target, int64_t n, int loss, int tonemapper, float* partials,
                                 gs_stream_t stream) {
    GS_REQUIRE(loss >= 0 && loss <= 3 && tonemapper >= 0 && tonemapper <= 1, "gs_image_loss_fwd: bad loss / tonemapper id");
    GS_REQUIRE(partials && (n == 0 || (img && target)), "gs_image_loss_fwd: null pointer");
    hipLaunchKernelGGL(k_image_loss_fwd, dim3((unsigned)gs_image_loss_partials(n)), dim3(256), 0, (hipStream_t)stream, img, target, n, loss,
                       tonemapper, partials);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_image_loss_bwd(const float* img, const float* target, int64_t n, int loss, int tonemapper, const float* g_scalar_dev,
                                 float scale, float* g_img, float* g_target, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_REQUIRE(loss >= 0 && loss <= 3 && tonemapper >= 0 && tonemapper <= 1, "gs_image_loss_bwd: bad loss / tonemapper id");
    GS_REQUIRE(img && target && g_scalar_dev, "gs_image_loss_bwd: null pointer");
    hipLaunchKernelGGL(k_image_loss_bwd, dim3((unsigned)gs::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, img, target, n, loss, tonemapper,
                       g_scalar_dev, scale, g_img, g_target);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_softplus_fwd(const float* x, int64_t n, float beta, float* y, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_REQUIRE(x && y && beta > 0.0f, "gs_softplus_fwd: bad argument");
    hipLaunchKernelGGL(k_softplus, dim3((unsigned)gs::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr, (const float*)nullptr, n,
                       beta, 0, y, (float*)nullptr);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_softplus_bwd(const float* x, const float* g, int64_t n, float beta, float* g_x, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_REQUIRE(x && g && g_x && beta > 0.0f, "gs_softplus_bwd: bad argument");
    hipLaunchKernelGGL(k_softplus, dim3((unsigned)gs::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, g, (const float*)nullptr, n, beta, 1, g_x,
                       (float*)nullptr);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_softplus_bwd_bwd(const float* x, const float* g, const float* gg, int64_t n, float beta, float* d_g, float* d_x,
                                   gs_stream_t stream) {
    if (n == 0) return 0;
    GS_REQUIRE(x && g && gg && (d_g || d_x) && beta > 0.0f, "gs_softplus_bwd_bwd: bad argument");
    hipLaunchKernelGGL(k_softplus, dim3((unsigned)gs::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, g, gg, n, beta, 2, d_g, d_x);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t gs_frame_sums_partials(int64_t n_pixels) { return gs::cdiv(n_pixels, 256); }

static FrameOffs frame_offs(const int32_t* offs) { return FrameOffs{offs[0], offs[1], offs[2], offs[3], offs[4], offs[5], offs[6]}; }

extern "C" int gs_frame_sums_fwd(const float* stacked, const float* color_ref, int64_t n_pixels, int64_t C, const int32_t* offs_host,
                                 float* partials, gs_stream_t stream) {
    if (n_pixels == 0) return 0;
    GS_REQUIRE(stacked && color_ref && offs_host && partials && C > 0, "gs_frame_sums_fwd: null pointer");
    for (int k = 0; k < 7; ++k) GS_REQUIRE(offs_host[k] < 0 || offs_host[k] + (k == 1 ? 1 : 4) <= C, "gs_frame_sums_fwd: channel offset out of range");
    GS_REQUIRE(C <= 64, "gs_frame_sums_fwd: at most 64 channels");
    hipLaunchKernelGGL(k_frame_sums<false>, dim3((unsigned)gs::cdiv(n_pixels, 256)), dim3(256), (size_t)256 * C * sizeof(float), (hipStream_t)stream, stacked,
                       color_ref, n_pixels, (int)C, frame_offs(offs_host), partials, (const float*)nullptr, (float*)nullptr, 9, -1, 0);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_frame_sums_img_fwd(const float* stacked, const float* color_ref, int64_t n_pixels, int64_t C, const int32_t* offs_host, int loss,
                                     int tonemapper, float* partials10, gs_stream_t stream) {
    if (n_pixels == 0) return 0;
    GS_REQUIRE(stacked && color_ref && offs_host && partials10 && C > 0, "gs_frame_sums_img_fwd: null pointer");
    GS_REQUIRE(loss >= 0 && loss <= 3 && tonemapper >= 0 && tonemapper <= 1 && offs_host[0] >= 0, "gs_frame_sums_img_fwd: bad loss / tonemapper / no shaded buffer");
    for (int k = 0; k < 7; ++k) GS_REQUIRE(offs_host[k] < 0 || offs_host[k] + (k == 1 ? 1 : 4) <= C, "gs_frame_sums_img_fwd: channel offset out of range");
    GS_REQUIRE(C <= 64, "gs_frame_sums_img_fwd: at most 64 channels");
    hipLaunchKernelGGL(k_frame_sums<false>, dim3((unsigned)gs::cdiv(n_pixels, 256)), dim3(256), (size_t)256 * C * sizeof(float), (hipStream_t)stream, stacked,
                       color_ref, n_pixels, (int)C, frame_offs(offs_host), partials10, (const float*)nullptr, (float*)nullptr, 10, loss, tonemapper);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_frame_sums_bwd(const float* stacked, const float* color_ref, int64_t n_pixels, int64_t C, const int32_t* offs_host,
                                 const float* g_sums_dev, float* g_stacked, gs_stream_t stream) {
    if (n_pixels == 0) return 0;
    GS_REQUIRE(stacked && color_ref && offs_host && g_sums_dev && g_stacked && C > 0, "gs_frame_sums_bwd: null pointer");
    for (int k = 0; k < 7; ++k) GS_REQUIRE(offs_host[k] < 0 || offs_host[k] + (k == 1 ? 1 : 4) <= C, "gs_frame_sums_bwd: channel offset out of range");
    GS_REQUIRE(C <= 64, "gs_frame_sums_bwd: at most 64 channels");
    hipLaunchKernelGGL(k_frame_sums<true>, dim3((unsigned)gs::cdiv(n_pixels, 256)), dim3(256), (size_t)256 * C * sizeof(float), (hipStream_t)stream, stacked,
                       color_ref, n_pixels, (int)C, frame_offs(offs_host), (float*)nullptr, g_sums_dev, g_stacked, 9, -1, 0);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_frame_sums_img_bwd(const float* stacked, const float* color_ref, int64_t n_pixels, int64_t C, const int32_t* offs_host, int loss,
                                     int tonemapper, const float* g_sums10_dev, float* g_stacked, gs_stream_t stream) {
    if (n_pixels == 0) return 0;
    GS_REQUIRE(stacked && color_ref && offs_host && g_sums10_dev && g_stacked && C > 0, "gs_frame_sums_img_bwd: null pointer");
    GS_REQUIRE(loss >= 0 && loss <= 3 && tonemapper >= 0 && tonemapper <= 1 && offs_host[0] >= 0, "gs_frame_sums_img_bwd: bad loss / tonemapper / no shaded buffer");
    for (int k = 0; k < 7; ++k) GS_REQUIRE(offs_host[k] < 0 || offs_host[k] + (k == 1 ? 1 : 4) <= C, "gs_frame_sums_img_bwd: channel offset out of range");
    GS_REQUIRE(C <= 64, "gs_frame_sums_img_bwd: at most 64 channels");
    hipLaunchKernelGGL(k_frame_sums<true>, dim3((unsigned)gs::cdiv(n_pixels, 256)), dim3(256), (size_t)256 * C * sizeof(float), (hipStream_t)stream, stacked,
                       color_ref, n_pixels, (int)C, frame_offs(offs_host), (float*)nullptr, g_sums10_dev, g_stacked, 10, loss, tonemapper);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_auto_normals_fwd(const float* v_pos, int64_t V, const int32_t* tri, int64_t T, float* acc, float* v_nrm, gs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (V == 0) return 0;
    GS_REQUIRE(v_pos && acc && v_nrm && (T == 0 || tri), "gs_auto_normals_fwd: null pointer");
    GS_HIP_CHECK(hipMemsetAsync(acc, 0, (size_t)V * 12, stream));
    if (T > 0) hipLaunchKernelGGL(k_face_normals_scatter, dim3((unsigned)gs::cdiv(T, 256)), dim3(256), 0, stream, v_pos, tri, T, acc);
    hipLaunchKernelGGL(k_vertex_normalize, dim3((unsigned)gs::cdiv(V, 256)), dim3(256), 0, stream, acc, V, v_nrm);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_auto_normals_bwd(const float* v_pos, int64_t V, const int32_t* tri, int64_t T, const float* acc, const float* g_nrm,
                                   float* g_acc, float* g_pos, gs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (V == 0) return 0;
    GS_REQUIRE(v_pos && acc && g_nrm && g_acc && g_pos && (T == 0 || tri), "gs_auto_normals_bwd: null pointer");
    hipLaunchKernelGGL(k_vertex_normalize_bwd, dim3((unsigned)gs::cdiv(V, 256)), dim3(256), 0, stream, acc, V, g_nrm, g_acc);
    if (T > 0) hipLaunchKernelGGL(k_face_normals_bwd, dim3((unsigned)gs::cdiv(T, 256)), dim3(256), 0, stream, v_pos, tri, T, g_acc, g_pos);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_texture_linear_fwd(const float* tex, int64_t B, int64_t H, int64_t W, int64_t C, const float* uv, int64_t n_per_view,
                                     float* out, gs_stream_t stream) {
    if (B * n_per_view == 0 || C == 0) return 0;
    GS_REQUIRE(tex && uv && out && H > 0 && W > 0, "gs_texture_linear_fwd: null pointer / empty texture");
    hipLaunchKernelGGL(k_texture_linear<false>, dim3((unsigned)gs::cdiv(B * n_per_view, 256)), dim3(256), 0, (hipStream_t)stream, tex, B, (int)H,
                       (int)W, (int)C, uv, n_per_view, out, (const float*)nullptr, (float*)nullptr);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_texture_linear_bwd(int64_t B, int64_t H, int64_t W, int64_t C, const float* uv, int64_t n_per_view, const float* g_out,
                                     float* g_tex, gs_stream_t stream) {
    if (B * n_per_view == 0 || C == 0) return 0;
    GS_REQUIRE(uv && g_out && g_tex, "gs_texture_linear_bwd: null pointer");
    hipLaunchKernelGGL(k_texture_linear<true>, dim3((unsigned)gs::cdiv(B * n_per_view, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)nullptr, B, (int)H, (int)W, (int)C, uv, n_per_view, (float*)nullptr, g_out, g_tex);
    GS_LAUNCH_CHECK();
    return 0;
}

// ---- SDF sign-consistency regulariser over the static grid edges ------------------------------------------
// Replaces compute_sdf_reg_loss (geometry/gshell_tets_geometry.py:33-39): for every grid edge (a,b) whose end
// points differ in torch.sign(), BCE-with-logits(s_a, s_b > 0) + BCE-with-logits(s_b, s_a > 0), each averaged over
// the crossing edges.  The reference gathers 2E values, builds a boolean mask, compacts (host sync) and lets
// autograd scatter back (index_put backward = a device-wide sort): ~260 ms per iteration at tet-res 256.  Here it
// is one streaming pass over the [E,2] int32 edge list (8 B / edge) with block-level partial sums, and a backward
// pass that touches only the ~2 % crossing edges.
namespace {

__device__ __forceinline__ float sgn3(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float bce_logits(float x, float t) { return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void __launch_bounds__(256) k_sdf_reg_fwd(const float* __restrict__ sdf, const int2* __restrict__ edges, int64_t E,
                                                     float* __restrict__ part_loss, float* __restrict__ part_cnt) {
    float acc = 0.f, cnt = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        int2 ab = edges[e];
        float sa = sdf[ab.x], sb = sdf[ab.y];
        if (sgn3(sa) != sgn3(sb)) {
            acc += bce_logits(sa, sb > 0.f ? 1.f : 0.f) + bce_logits(sb, sa > 0.f ? 1.f : 0.f);
            cnt += 1.f;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_xor(acc, o, 64);
        cnt += __shfl_xor(cnt, o, 64);
    }
    __shared__ float wa[4], wc[4];
    if ((threadIdx.x & 63) == 0) {
        wa[threadIdx.x >> 6] = acc;
        wc[threadIdx.x >> 6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part_loss[blockIdx.x] = (wa[0] + wa[1]) + (wa[2] + wa[3]);
        part_cnt[blockIdx.x] = (wc[0] + wc[1]) + (wc[2] + wc[3]);
    }
}

// g_sdf += g * d/ds [ (sum_e l_e) / K ],  K = *count_dev (number of crossing edges)
__global__ void __launch_bounds__(256) k_sdf_reg_bwd(const float* __restrict__ sdf, const int2* __restrict__ edges, int64_t E,
                                                     const float* __restrict__ g_scalar, const float* __restrict__ count_dev,
                                                     float* __restrict__ g_sdf) {
    float K = count_dev[0];
    if (!(K > 0.f)) return;
    float g = g_scalar[0] / K;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        int2 ab = edges[e];
        float sa = sdf[ab.x], sb = sdf[ab.y];
        if (sgn3(sa) != sgn3(sb)) {
            atomicAdd(&g_sdf[ab.x], g * (sigmoidf(sa) - (sb > 0.f ? 1.f : 0.f)));
            atomicAdd(&g_sdf[ab.y], g * (sigmoidf(sb) - (sa > 0.f ? 1.f : 0.f)));
        }
    }
}


// ---- surface samples for the eikonal term (kaolin.ops.mesh.sample_points stand-in, gshell_tets_geometry.py:236) --------------------
// area[t] = |(v1 - v0) x (v2 - v0)| (non-finite -> 0) + 1e-20: the weights of the face draw (torch.multinomial stays in torch: its
// generator is part of the reproducibility contract); then p = (1-u) v0 + u (1-v) v1 + u v v2 with (u, v) = (sqrt(r0), r1).
__global__ void __launch_bounds__(256) k_tri_area(const float* __restrict__ v, const int32_t* __restrict__ tri, int64_t T, float* __restrict__ area) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const f3 v0 = ld3(v + 3 * (int64_t)tri[3 * t]), v1 = ld3(v + 3 * (int64_t)tri[3 * t + 1]), v2 = ld3(v + 3 * (int64_t)tri[3 * t + 2]);
    const f3 n = cross(v1 - v0, v2 - v0);
    const float a = sqrtf(dot(n, n));
    area[t] = (isfinite(a) ? a : 0.0f) + 1e-20f;
}

__global__ void __launch_bounds__(256) k_surface_points(const float* __restrict__ v, const int32_t* __restrict__ tri, const int64_t* __restrict__ fid,
                                                        const float* __restrict__ r, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t t = fid[i];
    const f3 v0 = ld3(v + 3 * (int64_t)tri[3 * t]), v1 = ld3(v + 3 * (int64_t)tri[3 * t + 1]), v2 = ld3(v + 3 * (int64_t)tri[3 * t + 2]);
    const float u = sqrtf(r[2 * i]), w = r[2 * i + 1];
    const float a = 1.0f - u, b = u * (1.0f - w), c = u * w;
    out[3 * i] = a * v0.x + b * v1.x + c * v2.x;
    out[3 * i + 1] = a * v0.y + b * v1.y + c * v2.y;
    out[3 * i + 2] = a * v0.z + b * v1.z + c * v2.z;
}

// the face draw folded in: face = first t with cdf[t] > x * cdf[T-1] (cdf = inclusive prefix sum of the areas), x = r[3 i + 2] --
// the inversion torch.multinomial performs after a 0.12 ms row renormalisation of its own
__global__ void __launch_bounds__(256) k_surface_points_cdf(const float* __restrict__ v, const int32_t* __restrict__ tri, const float* __restrict__ cdf,
                                                            int64_t T, const float* __restrict__ r, int64_t n, float* __restrict__ out,
                                                            int64_t* __restrict__ fid_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = r[3 * i + 2] * cdf[T - 1];
    int64_t lo = 0, hi = T - 1;                       // smallest t with cdf[t] > x (t = T - 1 if none: x < total always)
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cdf[mid] > x) hi = mid; else lo = mid + 1;
    }
    const int64_t t = lo;
    if (fid_out) fid_out[i] = t;
    const f3 v0 = ld3(v + 3 * (int64_t)tri[3 * t]), v1 = ld3(v + 3 * (int64_t)tri[3 * t + 1]), v2 = ld3(v + 3 * (int64_t)tri[3 * t + 2]);
    const float u = sqrtf(r[3 * i]), w = r[3 * i + 1];
    const float a = 1.0f - u, b = u * (1.0f - w), c = u * w;
    out[3 * i] = a * v0.x + b * v1.x + c * v2.x;
    out[3 * i + 1] = a * v0.y + b * v1.y + c * v2.y;
    out[3 * i + 2] = a * v0.z + b * v1.z + c * v2.z;
}

// ---- light probe sampling tables (render/light.py:46-59) in two launches -----------------------------------------------------
// pdf = max_c(base) sin(theta_y) / sum;  cols[y] = cumsum_x pdf[y] / its last entry;  rows = cumsum_y (row masses) / total.
// As torch ops this is 15 launches on a 256 x 256 array before every iteration (train_gshelltet_deepfashion.py:412).
//   k_light_rows : a wave owns a row (lane = W / 64 consecutive columns, wave scan of the lane totals): w -> pdf (unnormalised), the row's
//                  CDF -> cols (the total cancels in cols), the row's mass -> mass[y]
//   k_light_norm : every workgroup scans the H row masses itself (H <= 1024 values), then normalises its share of pdf and writes rows
// (a first version did all of it in ONE workgroup: 104 us on one CU against 140 us for the 15 launches -- no gain)
__global__ void __launch_bounds__(256) k_light_rows(const float* __restrict__ base, int H, int W, float* __restrict__ pdf, float* __restrict__ cols,
                                                    float* __restrict__ mass) {
    const int lane = threadIdx.x & 63, y = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= H) return;
    const int per = (W + 63) / 64;
    const float st = sinf(((float)y + 0.5f) / (float)H * 3.14159265358979323846f);
    float run = 0.f;
    for (int k = 0; k < per; ++k) {
        const int x = lane * per + k;
        if (x < W) {
            const float* b = base + ((int64_t)y * W + x) * 3;
            const float w = fmaxf(fmaxf(b[0], b[1]), b[2]) * st;
            pdf[(int64_t)y * W + x] = w;
            run += w;
        }
    }
    float inc = run;                                  // inclusive scan of the lane totals
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    const float m = __shfl(inc, 63, 64);
    const float inv = m > 0.f ? m : 1.0f;
    float c = inc - run;
    for (int k = 0; k < per; ++k) {
        const int x = lane * per + k;
        if (x < W) {
            c += pdf[(int64_t)y * W + x];
            cols[(int64_t)y * W + x] = c / inv;
        }
    }
    if (lane == 0) mass[y] = m;
}

__global__ void __launch_bounds__(256) k_light_norm(const float* __restrict__ mass, int H, int W, float* __restrict__ pdf, float* __restrict__ rows) {
    __shared__ float s_cdf[1024];
    __shared__ float s_part[4];
    // inclusive scan of the row masses: thread t owns 4 consecutive rows, wave scan, 4 wave totals
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float v[4], run = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int y = 4 * t + k;
        v[k] = y < H ? mass[y] : 0.f;
        run += v[k];
    }
    float inc = run;
    for (int o = 1; o < 64; o <<= 1) {
        const float u = __shfl_up(inc, o, 64);
        if (lane >= o) inc += u;
    }
    if (lane == 63) s_part[wave] = inc;
    __syncthreads();
    float off = 0.f;
    for (int w = 0; w < wave; ++w) off += s_part[w];
    const float total = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    float c = off + inc - run;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c += v[k];
        if (4 * t + k < H) s_cdf[4 * t + k] = c;
    }
    __syncthreads();
    const float tinv = total > 0.f ? total : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + t; i < (int64_t)H * W; i += (int64_t)gridDim.x * 256) {
        pdf[i] = pdf[i] / total;
        rows[i] = s_cdf[i / W] / tinv;
    }
}

// ---- mSDF open / close regularisers (gshell_tets_geometry.py:326-358) -----------------------------------------------
// Huber (delta = 1) distance of the clamped mSDF values to -eps (all N grid values, "open") / +eps (the boundary vertices of
// triangles some view saw, "close").  As ATen ops: clamp, expand, huber_loss, mul, sum per term plus the visibility mask
// (unique -> gather -> boolean mask in the reference): ~35 launches forward, ~35 backward on a 2.3 M-element array.
__device__ __forceinline__ float huber1(float d) { return fabsf(d) < 1.0f ? 0.5f * d * d : fabsf(d) - 0.5f; }
__device__ __forceinline__ float huber1_grad(float d) { return fabsf(d) < 1.0f ? d : (d > 0.f ? 1.0f : -1.0f); }

// w[k] = 1 for every boundary vertex (mesh vertex nwt + k) of a flagged triangle: plain stores of the same value, no atomics
__global__ void __launch_bounds__(256) k_boundary_weight(const int32_t* __restrict__ tri, const uint8_t* __restrict__ flags, int64_t T, int64_t nwt,
                                                         int64_t nb, float* __restrict__ w) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= T || !flags[t]) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int64_t v = (int64_t)tri[3 * t + k] - nwt;
        if (v >= 0 && v < nb) w[v] = 1.0f;
    }
}

// out[0] += open_w * sum_i huber(max(m_i, -eps) + eps);  out[1] += close_w * sum_j w_j huber(min(b_j, eps) - eps)
__global__ void __launch_bounds__(256) k_msdf_reg_fwd(const float* __restrict__ m, int64_t N, const float* __restrict__ b, const float* __restrict__ w,
                                                      int64_t nb, float eps, float open_w, float close_w, float* __restrict__ out) {
    float a0 = 0.f, a1 = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (open_w != 0.f)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += stride) a0 += huber1(fmaxf(m[i], -eps) + eps);
    if (close_w != 0.f && w)
        for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < nb; j += stride) {
            const float wj = w[j];
            if (wj != 0.f) a1 += wj * huber1(fminf(b[j], eps) - eps);
        }
    for (int o = 32; o > 0; o >>= 1) {
        a0 += __shfl_xor(a0, o, 64);
        a1 += __shfl_xor(a1, o, 64);
    }
    __shared__ float s0[4], s1[4];
    if ((threadIdx.x & 63) == 0) {
        s0[threadIdx.x >> 6] = a0;
        s1[threadIdx.x >> 6] = a1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t0 = (s0[0] + s0[1]) + (s0[2] + s0[3]), t1 = (s1[0] + s1[1]) + (s1[2] + s1[3]);
        if (t0 != 0.f) atomicAdd(&out[0], open_w * t0);
        if (t1 != 0.f) atomicAdd(&out[1], close_w * t1);
    }
}

// g_m[i] = g[0] open_w huber'(.) [m_i >= -eps];  g_b[j] = g[1] close_w w_j huber'(.) [b_j <= eps]   (WRITTEN)
__global__ void __launch_bounds__(256) k_msdf_reg_bwd(const float* __restrict__ m, int64_t N, const float* __restrict__ b, const float* __restrict__ w,
                                                      int64_t nb, float eps, float open_w, float close_w, const float* __restrict__ g,
                                                      float* __restrict__ g_m, float* __restrict__ g_b) {
    const float g0 = g[0] * open_w, g1 = g[1] * close_w;
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (g_m)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += stride) {
            const float v = m[i];
            g_m[i] = v >= -eps ? g0 * huber1_grad(v + eps) : 0.0f;
        }
    if (g_b)
        for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < nb; j += stride) {
            const float v = b[j], wj = w ? w[j] : 0.0f;
            g_b[j] = (v <= eps && wj != 0.f) ? g1 * wj * huber1_grad(v - eps) : 0.0f;
        }
}

}  // namespace

extern "C" int64_t gs_sdf_reg_partials(int64_t E) { return std::min<int64_t>(std::max<int64_t>(gs::cdiv(E, 256 * 8), 1), 4096); }

extern "C" int gs_sdf_reg_fwd(const float* sdf, const int32_t* edges, int64_t E, float* part_loss, float* part_count, gs_stream_t stream) {
    GS_REQUIRE(part_loss && part_count && (E == 0 || (sdf && edges)), "gs_sdf_reg_fwd: null pointer");
    hipLaunchKernelGGL(k_sdf_reg_fwd, dim3((unsigned)gs_sdf_reg_partials(E)), dim3(256), 0, (hipStream_t)stream, sdf, (const int2*)edges, E, part_loss,
                       part_count);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_sdf_reg_bwd(const float* sdf, const int32_t* edges, int64_t E, const float* g_scalar_dev, const float* count_dev, float* g_sdf,
                              gs_stream_t stream) {
    if (E == 0) return 0;
    GS_REQUIRE(sdf && edges && g_scalar_dev && count_dev && g_sdf, "gs_sdf_reg_bwd: null pointer");
    hipLaunchKernelGGL(k_sdf_reg_bwd, dim3((unsigned)gs_sdf_reg_partials(E)), dim3(256), 0, (hipStream_t)stream, sdf, (const int2*)edges, E, g_scalar_dev,
                       count_dev, g_sdf);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_boundary_weight(const int32_t* tri, const uint8_t* flags, int64_t T, int64_t n_watertight, int64_t n_boundary, float* w,
                                  gs_stream_t stream) {
    if (n_boundary == 0) return 0;
    GS_REQUIRE(w && (T == 0 || (tri && flags)), "gs_boundary_weight: null pointer");
    GS_HIP_CHECK(hipMemsetAsync(w, 0, (size_t)n_boundary * sizeof(float), (hipStream_t)stream));
    if (T == 0) return 0;
    hipLaunchKernelGGL(k_boundary_weight, dim3((unsigned)gs::cdiv(T, 256)), dim3(256), 0, (hipStream_t)stream, tri, flags, T, n_watertight, n_boundary, w);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_msdf_reg_fwd(const float* msdf, int64_t N, const float* msdf_boundary, const float* weight, int64_t n_boundary, float eps,
                               float open_w, float close_w, float* out2, gs_stream_t stream) {
    GS_REQUIRE(out2 && (N == 0 || msdf) && (n_boundary == 0 || !weight || msdf_boundary), "gs_msdf_reg_fwd: null pointer");
    GS_HIP_CHECK(hipMemsetAsync(out2, 0, 2 * sizeof(float), (hipStream_t)stream));
    const int64_t n = std::max(N, n_boundary);
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_msdf_reg_fwd, dim3((unsigned)std::min<int64_t>(gs::cdiv(n, 256), 1024)), dim3(256), 0, (hipStream_t)stream, msdf, N, msdf_boundary,
                       weight, n_boundary, eps, open_w, close_w, out2);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_msdf_reg_bwd(const float* msdf, int64_t N, const float* msdf_boundary, const float* weight, int64_t n_boundary, float eps,
                               float open_w, float close_w, const float* g_out2_dev, float* g_msdf, float* g_boundary, gs_stream_t stream) {
    GS_REQUIRE(g_out2_dev && (N == 0 || msdf) && (n_boundary == 0 || !g_boundary || msdf_boundary), "gs_msdf_reg_bwd: null pointer");
    const int64_t n = std::max(g_msdf ? N : 0, g_boundary ? n_boundary : 0);
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_msdf_reg_bwd, dim3((unsigned)std::min<int64_t>(gs::cdiv(n, 256), 2048)), dim3(256), 0, (hipStream_t)stream, msdf, N, msdf_boundary,
                       weight, n_boundary, eps, open_w, close_w, g_out2_dev, g_msdf, g_boundary);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_depth_zgrad(const float* clip_pos, const float* clip_pos_deriv, int64_t n, float eps, float* out, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_REQUIRE(clip_pos && clip_pos_deriv && out, "gs_depth_zgrad: null pointer");
    hipLaunchKernelGGL(k_depth_zgrad, dim3((unsigned)gs::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)clip_pos, (const float4*)clip_pos_deriv,
                       n, eps, (float2*)out);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_tri_area(const float* v_pos, const int32_t* tri, int64_t T, float* area, gs_stream_t stream) {
    if (T == 0) return 0;
    GS_REQUIRE(v_pos && tri && area, "gs_tri_area: null pointer");
    hipLaunchKernelGGL(k_tri_area, dim3((unsigned)gs::cdiv(T, 256)), dim3(256), 0, (hipStream_t)stream, v_pos, tri, T, area);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_surface_points_cdf(const float* v_pos, const int32_t* tri, const float* area_cdf, int64_t T, const float* r01x3, int64_t n, float* out,
                                     int64_t* face_id, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_REQUIRE(v_pos && tri && area_cdf && r01x3 && out && T > 0, "gs_surface_points_cdf: null pointer / empty mesh");
    hipLaunchKernelGGL(k_surface_points_cdf, dim3((unsigned)gs::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, v_pos, tri, area_cdf, T, r01x3, n, out, face_id);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_light_tables(const float* base, int64_t H, int64_t W, float* pdf, float* rows, float* cols, float* row_mass_scratch,
                               gs_stream_t stream) {
    GS_REQUIRE(base && pdf && rows && cols && row_mass_scratch, "gs_light_tables: null pointer");
    GS_REQUIRE(H >= 1 && H <= 1024 && W >= 1, "gs_light_tables: probe height must be in 1..1024");
    hipLaunchKernelGGL(k_light_rows, dim3((unsigned)gs::cdiv(H, 4)), dim3(256), 0, (hipStream_t)stream, base, (int)H, (int)W, pdf, cols, row_mass_scratch);
    hipLaunchKernelGGL(k_light_norm, dim3((unsigned)std::min<int64_t>(gs::cdiv(H * W, 256), 64)), dim3(256), 0, (hipStream_t)stream, row_mass_scratch, (int)H,
                       (int)W, pdf, rows);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_surface_points(const float* v_pos, const int32_t* tri, const int64_t* face_id, const float* r01, int64_t n, float* out,
                                 gs_stream_t stream) {
    if (n == 0) return 0;
    GS_REQUIRE(v_pos && tri && face_id && r01 && out, "gs_surface_points: null pointer");
    hipLaunchKernelGGL(k_surface_points, dim3((unsigned)gs::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, v_pos, tri, face_id, r01, n, out);
    GS_LAUNCH_CHECK();
    return 0;
}

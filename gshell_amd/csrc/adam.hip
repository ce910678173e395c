// Adam step over a list of parameter tensors in ONE launch (gfx950).
//
// The reference trains with torch.optim.Adam (train_gshelltet_deepfashion.py:372-383: geometry, material and light optimisers; 22 M
// parameters at tet-res 256: the 12.6 M-entry texture table, 6.8 M deformations, 2.3 M mSDF values, the two MLPs, the probe).  ATen's
// multi-tensor Adam moves its 28 bytes per parameter at 1.6 TB/s on MI355X (0.38 ms per iteration in 5 launches + 5 step-counter
// launches); this kernel reads and writes the same bytes with 16-byte accesses from 4096-element chunks: HBM bound.
//
// Arithmetic = at::native's fused Adam (ATen/native/cuda/fused_adam_utils.cuh:61-76, amsgrad off, weight_decay 0, maximize off), which
// mixes double scalars with float operands:
//     m  = float( beta1 * double(m) + (1 - beta1) * double(g) )
//     v  = float( beta2 * double(v) + (1 - beta2) * double(g) * double(g) )
//     bc1 = float(1 - pow(beta1, step)),  bc2s = float(sqrt(1 - pow(beta2, step)))          (double pow / sqrt on the device, :127-133)
//     step_size = float( lr / double(bc1) );   denom = float( double(sqrtf(v) / bc2s) + eps );   p -= step_size * m / denom
// `contract` selects how the two moment updates round, because ATen is built with hipcc's default -ffp-contract=fast and this library
// with -ffp-contract=off: 0 every product and sum rounded, 1 fma(beta, old, (1-beta) g [g]), 2 fma((1-beta) g [g], ... , beta old).
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/gshell_hip.h"
#include "common.hpp"

namespace {

constexpr int ADAM_MAX_TENSORS = 40;
constexpr int ADAM_CHUNK = 4096;          // elements per workgroup: 256 lanes x 4 x float4

struct AdamTable {
    float* p[ADAM_MAX_TENSORS];
    const float* g[ADAM_MAX_TENSORS];
    float* m[ADAM_MAX_TENSORS];
    float* v[ADAM_MAX_TENSORS];
    float* step[ADAM_MAX_TENSORS];        // device float32 step counters (torch's fused optimiser state), WRITTEN = step_value
    int64_t numel[ADAM_MAX_TENSORS];
    uint32_t first_chunk[ADAM_MAX_TENSORS + 1];
    double lr_d[ADAM_MAX_TENSORS];        // per tensor: parameter groups differ in lr only
    int n;
};

template <int CONTRACT>
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, double lr, double beta1, double beta2, double eps, float bc1, float bc2s) {
    const double gd = (double)g, omb1 = 1.0 - beta1, omb2 = 1.0 - beta2;
    double md, vd;
    if (CONTRACT == 0) {
        md = beta1 * (double)m + omb1 * gd;
        vd = beta2 * (double)v + (omb2 * gd) * gd;
    } else if (CONTRACT == 1) {
        md = __builtin_fma(beta1, (double)m, omb1 * gd);
        vd = __builtin_fma(beta2, (double)v, (omb2 * gd) * gd);
    } else {
        md = __builtin_fma(omb1, gd, beta1 * (double)m);
        vd = __builtin_fma(omb2 * gd, gd, beta2 * (double)v);
    }
    m = (float)md;
    v = (float)vd;
    const float step_size = (float)(lr / (double)bc1);
    const float denom = (float)((double)(sqrtf(v) / bc2s) + eps);
    p -= step_size * m / denom;
}

template <int CONTRACT>
__global__ void __launch_bounds__(256) k_adam(AdamTable T, double beta1, double beta2, double eps, float step_value) {
    // which tensor this chunk belongs to: the table is in scalar registers, the search is uniform
    int t = 0;
    for (int k = 1; k < T.n; ++k)
        if (blockIdx.x >= T.first_chunk[k]) t = k;
    const int64_t base = (int64_t)(blockIdx.x - T.first_chunk[t]) * ADAM_CHUNK;
    const int64_t n = T.numel[t] - base;
    const float bc1 = (float)(1.0 - pow(beta1, (double)step_value));
    const float bc2s = (float)sqrt(1.0 - pow(beta2, (double)step_value));
    const double lr = T.lr_d[t];
    float* p = T.p[t] + base;
    const float* g = T.g[t] + base;
    float* m = T.m[t] + base;
    float* v = T.v[t] + base;
    if (blockIdx.x == T.first_chunk[t] && threadIdx.x == 0 && T.step[t]) *T.step[t] = step_value;
    const bool vec = n >= ADAM_CHUNK && ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    if (vec) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = (u * 256 + threadIdx.x) * 4;
            float4 P = *reinterpret_cast<float4*>(p + i), M = *reinterpret_cast<float4*>(m + i), V = *reinterpret_cast<float4*>(v + i);
            const float4 G = *reinterpret_cast<const float4*>(g + i);
            adam_one<CONTRACT>(P.x, G.x, M.x, V.x, lr, beta1, beta2, eps, bc1, bc2s);
            adam_one<CONTRACT>(P.y, G.y, M.y, V.y, lr, beta1, beta2, eps, bc1, bc2s);
            adam_one<CONTRACT>(P.z, G.z, M.z, V.z, lr, beta1, beta2, eps, bc1, bc2s);
            adam_one<CONTRACT>(P.w, G.w, M.w, V.w, lr, beta1, beta2, eps, bc1, bc2s);
            *reinterpret_cast<float4*>(p + i) = P;
            *reinterpret_cast<float4*>(m + i) = M;
            *reinterpret_cast<float4*>(v + i) = V;
        }
    } else {
        for (int i = threadIdx.x; i < ADAM_CHUNK && i < n; i += 256) {
            float P = p[i], M = m[i], V = v[i];
            adam_one<CONTRACT>(P, g[i], M, V, lr, beta1, beta2, eps, bc1, bc2s);
            p[i] = P;
            m[i] = M;
            v[i] = V;
        }
    }
}

}  // namespace

extern "C" int gs_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                            float* const* step_tensors, const int64_t* numel, const double* lr, double beta1, double beta2, double eps,
                            double step_value, int contract, gs_stream_t stream) {
    GS_REQUIRE(n_tensors >= 0 && params && grads && exp_avg && exp_avg_sq && numel && lr, "gs_adam_step: null argument");
    GS_REQUIRE(step_value >= 1.0 && contract >= 0 && contract <= 2, "gs_adam_step: step_value counts from 1; contract is 0, 1 or 2");
    int done = 0;
    while (done < n_tensors) {          // more tensors than one table holds: several launches
        AdamTable T{};
        uint64_t chunks = 0;
        while (done < n_tensors && T.n < ADAM_MAX_TENSORS) {
            const int k = done++;
            if (numel[k] == 0) continue;
            GS_REQUIRE(params[k] && grads[k] && exp_avg[k] && exp_avg_sq[k] && numel[k] > 0, "gs_adam_step: null tensor");
            T.p[T.n] = params[k]; T.g[T.n] = grads[k]; T.m[T.n] = exp_avg[k]; T.v[T.n] = exp_avg_sq[k];
            T.step[T.n] = step_tensors ? step_tensors[k] : nullptr;
            T.numel[T.n] = numel[k];
            T.lr_d[T.n] = lr[k];
            T.first_chunk[T.n] = (uint32_t)chunks;
            chunks += (uint64_t)gs::cdiv(numel[k], ADAM_CHUNK);
            GS_REQUIRE(chunks < (1ull << 31), "gs_adam_step: too many parameters for one launch");
            ++T.n;
        }
        if (T.n == 0) break;
        T.first_chunk[T.n] = (uint32_t)chunks;
        const dim3 grid((unsigned)chunks), block(256);
        if (contract == 0) hipLaunchKernelGGL(k_adam<0>, grid, block, 0, (hipStream_t)stream, T, beta1, beta2, eps, (float)step_value);
        else if (contract == 1) hipLaunchKernelGGL(k_adam<1>, grid, block, 0, (hipStream_t)stream, T, beta1, beta2, eps, (float)step_value);
        else hipLaunchKernelGGL(k_adam<2>, grid, block, 0, (hipStream_t)stream, T, beta1, beta2, eps, (float)step_value);
    }
    GS_LAUNCH_CHECK();
    return 0;
}

// Multiresolution hash-grid encoding on gfx950 (fwd + bwd).
//
// Replaces the third-party tiny-cuda-nn `tcnn.Encoding(3, {"otype": "HashGrid", ...})` used by the reference's
// MLPTexture3D (render/mlptexture.py:57-73, sampled twice per iteration at render/render.py:68,70).  tiny-cuda-nn is
// not in the reference tree; semantics restated from its public description (SURVEY.md 8c [3P-memory]):
//   level l: scale = 2^(l log2(per_level_scale)) * base - 1, res = ceil(scale) + 1,
//   pos = x * scale + 0.5, trilinear over the 8 corners of floor(pos),
//   corner index = dense (x + y res + z res^2) while the level fits its table, else
//   (x ^ y*2654435761 ^ z*805459861) mod table_size; table_size = min(next_multiple(res^3, 8), 2^log2_T);
//   F features per level, levels concatenated -> [N, L*F].
// Precision: fp32 parameters and outputs (tiny-cuda-nn defaults to half; fp32 >= the reference's precision).
//
// MI355X mapping: one lane per (point, level) with the level index fastest across blockIdx.y so that the 16 levels of
// a point run concurrently and each level's table (<= 4 MiB = one XCD L2) is walked by its own workgroups; gathers are
// 8-byte (float2, F = 2) and L2/Infinity-Cache resident (all tables together <= 48 MiB).  Rows with mask == 0
// (background pixels) are skipped.  Backward: float2 atomics into the table + dL/dx (the reference's g-buffer
// positions carry gradient into the texture lookup).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "../../include/gshell_hip.h"
#include "common.hpp"

namespace {

constexpr int MAX_LEVELS = 32;

struct GridMeta {
    int n_levels, F;
    uint32_t offset[MAX_LEVELS + 1];  // in entries (F floats each)
    uint32_t res[MAX_LEVELS];
    float scale[MAX_LEVELS];
};

__device__ __forceinline__ uint32_t grid_index(uint32_t x, uint32_t y, uint32_t z, uint32_t res, uint32_t size) {
    // dense while res^3 fits, else spatial hash (coherent prime hash of tiny-cuda-nn)
    uint64_t dense = (uint64_t)res * res * res;
    uint32_t idx;
    if (dense <= (uint64_t)size)
        idx = x + y * res + z * res * res;
    else
        idx = x ^ (y * 2654435761u) ^ (z * 805459861u);
    return idx % size;
}

template <bool BWD>
__global__ void __launch_bounds__(256) k_hashgrid(GridMeta M, const float* __restrict__ x, const float* __restrict__ mask, int64_t N,
                                                  const float* __restrict__ params, float* __restrict__ out, const float* __restrict__ g_out,
                                                  float* __restrict__ g_params, float* __restrict__ g_x_lvl) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int l = blockIdx.y;
    const int F = M.F;
    const int C = M.n_levels * F;
    const bool in_range = i < N;
    const bool active = in_range && !(mask && !(mask[i] > 0.0f));
    if (in_range && !active) {
        if (!BWD)
            for (int f = 0; f < F; ++f) out[i * C + l * F + f] = 0.f;
        else if (g_x_lvl)
            for (int k = 0; k < 3; ++k) g_x_lvl[((int64_t)l * N + i) * 3 + k] = 0.f;
    }
    // forward: inactive lanes are done.  backward: they stay for the wave-level run combining below (as empty lanes),
    // unless the whole wave is empty.
    if (!BWD && !active) return;
    if (BWD && __ballot(active) == 0ull) return;
    const int lane = threadIdx.x & 63;
    float scale = M.scale[l];
    uint32_t res = M.res[l];
    uint32_t size = M.offset[l + 1] - M.offset[l];
    float p[3], w[3];
    uint32_t g0[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p[k] = (active ? x[3 * i + k] : 0.0f) * scale + 0.5f;
        float fl = floorf(p[k]);
        w[k] = p[k] - fl;
        g0[k] = (uint32_t)(int)fl;
    }
    const float* tab = params + (int64_t)M.offset[l] * F;
    float acc[8];
    for (int f = 0; f < F && f < 8; ++f) acc[f] = 0.f;
    float gx[3] = {0.f, 0.f, 0.f};
    float go[8];
    if (BWD)
        for (int f = 0; f < F && f < 8; ++f) go[f] = active ? g_out[i * C + l * F + f] : 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        uint32_t cx = g0[0] + (c & 1), cy = g0[1] + ((c >> 1) & 1), cz = g0[2] + ((c >> 2) & 1);
        float wx = (c & 1) ? w[0] : 1.0f - w[0];
        float wy = (c & 2) ? w[1] : 1.0f - w[1];
        float wz = (c & 4) ? w[2] : 1.0f - w[2];
        float wgt = wx * wy * wz;
        uint32_t idx = grid_index(cx, cy, cz, res, size);
        const float* e = tab + (int64_t)idx * F;
        if (!BWD) {
            for (int f = 0; f < F && f < 8; ++f) acc[f] += wgt * e[f];
        } else {
            // Lanes are consecutive pixels of a scan line, so up to the mid levels neighbouring lanes fall into the same
            // cell and would each fire an atomic at the same table entry (the backward is bound by the float-atomic rate:
            // 38 M atomics per call).  Runs of equal entries along the wave are summed by a segmented shuffle reduction
            // (suffix sums inside a run: no cancellation) and only the first lane of a run issues the atomic.
            const uint32_t key = active ? idx : (0x80000000u | (uint32_t)lane);          // empty lanes never merge
            const uint32_t prev = __shfl_up(key, 1, 64);
            const bool head = lane == 0 || key != prev;
            const uint64_t heads = __ballot(head);
            const uint64_t after = lane == 63 ? 0ull : (heads & ~((2ull << lane) - 1ull));
            const int end = after ? (__ffsll((long long)after) - 2) : 63;                 // last lane of this lane's run
            float dotp = 0.f;
            for (int f = 0; f < F && f < 8; ++f) {
                if (active) dotp += go[f] * e[f];
                float v = active ? wgt * go[f] : 0.0f;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const float ov = __shfl_down(v, d, 64);
                    if (lane + d <= end) v += ov;
                }
                if (g_params && head && active && v != 0.f) atomicAdd(&g_params[((int64_t)M.offset[l] + idx) * F + f], v);
            }
            gx[0] += ((c & 1) ? 1.f : -1.f) * wy * wz * dotp;
            gx[1] += ((c & 2) ? 1.f : -1.f) * wx * wz * dotp;
            gx[2] += ((c & 4) ? 1.f : -1.f) * wx * wy * dotp;
        }
    }
    if (!active) return;
    if (!BWD) {
        for (int f = 0; f < F && f < 8; ++f) out[i * C + l * F + f] = acc[f];
    } else if (g_x_lvl) {
        for (int k = 0; k < 3; ++k) g_x_lvl[((int64_t)l * N + i) * 3 + k] = gx[k] * scale;
    }
}

}  // namespace

static int make_meta(GridMeta& M, int n_levels, int F, int log2_T, int base_res, float per_level_scale) {
    GS_REQUIRE(n_levels >= 1 && n_levels <= MAX_LEVELS && F >= 1 && F <= 8 && log2_T >= 1 && log2_T <= 30 && base_res >= 1,
               "hashgrid: unsupported configuration");
    M.n_levels = n_levels;
    M.F = F;
    uint64_t off = 0;
    for (int l = 0; l < n_levels; ++l) {
        float scale = (float)(std::exp2((double)l * std::log2((double)per_level_scale)) * (double)base_res - 1.0);  // double: bit-stable
        uint32_t res = (uint32_t)std::ceil(scale) + 1u;
        uint64_t n = (uint64_t)res * res * res;
        n = (n + 7) / 8 * 8;
        n = std::min<uint64_t>(n, 1ull << log2_T);
        M.scale[l] = scale;
        M.res[l] = res;
        M.offset[l] = (uint32_t)off;
        off += n;
        GS_REQUIRE(off < (1ull << 32), "hashgrid: table too large");
    }
    M.offset[n_levels] = (uint32_t)off;
    return 0;
}

extern "C" int64_t gs_hashgrid_num_params(int n_levels, int F, int log2_T, int base_res, float per_level_scale) {
    GridMeta M;
    if (make_meta(M, n_levels, F, log2_T, base_res, per_level_scale)) return -1;
    return (int64_t)M.offset[n_levels] * F;
}

extern "C" int gs_hashgrid_fwd(int n_levels, int F, int log2_T, int base_res, float per_level_scale, const float* x, const float* mask, int64_t N,
                               const float* params, float* out, gs_stream_t stream) {
    GridMeta M;
    int rc = make_meta(M, n_levels, F, log2_T, base_res, per_level_scale);
    if (rc) return rc;
    if (N == 0) return 0;
    GS_REQUIRE(x && params && out, "gs_hashgrid_fwd: null pointer");
    dim3 grid((unsigned)gs::cdiv(N, 256), (unsigned)n_levels);
    hipLaunchKernelGGL(k_hashgrid<false>, grid, dim3(256), 0, (hipStream_t)stream, M, x, mask, N, params, out, (const float*)nullptr, (float*)nullptr,
                       (float*)nullptr);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_hashgrid_bwd(int n_levels, int F, int log2_T, int base_res, float per_level_scale, const float* x, const float* mask, int64_t N,
                               const float* params, const float* g_out, float* g_params, float* g_x_levels, gs_stream_t stream) {
    GridMeta M;
    int rc = make_meta(M, n_levels, F, log2_T, base_res, per_level_scale);
    if (rc) return rc;
    if (N == 0) return 0;
    GS_REQUIRE(x && params && g_out, "gs_hashgrid_bwd: null pointer");
    dim3 grid((unsigned)gs::cdiv(N, 256), (unsigned)n_levels);
    hipLaunchKernelGGL(k_hashgrid<true>, grid, dim3(256), 0, (hipStream_t)stream, M, x, mask, N, params, (float*)nullptr, g_out, g_params, g_x_levels);
    GS_LAUNCH_CHECK();
    return 0;
}

// Texture-field MLP on gfx950: 32 -> 32 -> 32 -> C bias-free ReLU network + sigmoid range mapping, fwd + bwd.
//
// Replaces the torch `_MLP` (three bias-free nn.Linear + ReLU) and the sigmoid rescale that follow the hash-grid
// encoding in MLPTexture3D.sample (reference render/mlptexture.py:18-44, :87-99).  As torch ops this network is
// pathological on any GPU: ~1 M rows x 32 features is no work for the matrix cores, but every layer round-trips a
// 128 MB activation through HBM and the weight-gradient GEMMs (32 x 32 outputs, K = 1 M) run on ONE workgroup each
// -- 10 ms per iteration for 2 GFLOP.  Here:
//   * one lane per row; a row's vectors live in a padded LDS tile ([row][33], conflict free) and every matrix-vector
//     product is a RUNTIME loop over the input index that reads one tile value and one contiguous weight row
//     (broadcast ds_read_b128) and updates 32 register accumulators.  (A fully unrolled product is one 1000-FMA basic
//     block: the scheduler hoists all 256 weight loads above it, runs out of registers and parks them in scratch --
//     measured 512 VGPRs + 1988 spills.)  The row's 32 features are read once and its C outputs written once;
//   * rows with mask <= 0 (background pixels) are skipped; whole waves of them retire immediately;
//   * backward recomputes the forward pass, back-propagates through the same loops with the untransposed weights and
//     reduces the weight gradients over the 64 rows of a wave on the matrix core: dW[i][j] += sum_rows g[row][i] *
//     h[row][j] is exactly v_mfma_f32_32x32x2_f32 with the rows as the K dimension (two rows per instruction, operands
//     read straight from the tiles), accumulated in registers over a persistent grid-stride loop and flushed with
//     one atomic per weight per block.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/gshell_hip.h"
#include "common.hpp"

namespace {

constexpr int D = 32;         // input / hidden width
constexpr int CMAX = 8;       // output channels <= 8
constexpr int TP = D + 1;     // padded row stride of the LDS tiles
constexpr int TILE = 64 * TP;

typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef GS_TEX_ABL
#define GS_TEX_ABL 0      // experiments: 1 = no weight-gradient MFMAs, 2 = no matrix-vector products, 4 = no final atomics
#endif
#ifndef GS_TEX_MFMA
#define GS_TEX_MFMA 1     // layer products of a 64-row tile on v_mfma_f32_32x32x2_f32 (0: one lane per row, 1024 FMAs per layer and lane)
#endif

struct TexArgs {
    const float* x; const float* mask; int64_t N;
    const float* w1; const float* w2; const float* w3; int C;
    const float* lo; const float* hi;
    float* out;
    const float* g_out; float* g_x; float* g_w1; float* g_w2; float* g_w3;
    int level_major;   // x / g_x are [16][N][2] (the hash-grid encoding's level-major feature tensor, hashgrid.hip) instead of [N][32]
    // optional compact list of the rows to evaluate (ascending, count on the device): lane i takes row rows[i].  Image rows in scan
    // order put a 64-row chunk on every crossing of the silhouette: a third of the lanes of the "active" chunks did work.
    const int32_t* rows;
    const int64_t* count_dev;
};

// weights into LDS: w1t / w2t are the transposes (forward products walk rows of W^T, backward products rows of W)
__device__ __forceinline__ void load_weights(const TexArgs& A, float* s_w1, float* s_w2, float* s_w1t, float* s_w2t, float* s_w3, float* s_lo,
                                             float* s_hi) {
    for (int i = threadIdx.x; i < D * D; i += blockDim.x) {
        const float a = A.w1[i], b = A.w2[i];
        const int t = (i & (D - 1)) * D + (i >> 5);
        s_w1t[t] = a; s_w2t[t] = b;
        if (s_w1) { s_w1[i] = a; s_w2[i] = b; }
    }
    for (int i = threadIdx.x; i < CMAX * D; i += blockDim.x) s_w3[i] = i < A.C * D ? A.w3[i] : 0.0f;
    if (threadIdx.x < CMAX) {
        s_lo[threadIdx.x] = threadIdx.x < A.C ? A.lo[threadIdx.x] : 0.0f;
        s_hi[threadIdx.x] = threadIdx.x < A.C ? A.hi[threadIdx.x] : 0.0f;
    }
    __syncthreads();
}

// y[0..31] = sum_{k < n} rows[k][0..31] * vec[k], vec = this lane's row of an LDS tile
__device__ __forceinline__ void matvec(const float* __restrict__ rows, const float* __restrict__ vec, int n, float (&y)[D]) {
#pragma unroll
    for (int j = 0; j < D; ++j) y[j] = 0.0f;
    if (GS_TEX_ABL & 2) {
        for (int j = 0; j < D; ++j) y[j] = vec[j & 7] * rows[j];
        return;
    }
#pragma unroll 2
    for (int k = 0; k < n; ++k) {
        const float vk = vec[k];
        const float4* r = reinterpret_cast<const float4*>(rows + k * D);
#pragma unroll
        for (int j = 0; j < D / 4; ++j) {
            const float4 w = r[j];
            y[4 * j] = __builtin_fmaf(w.x, vk, y[4 * j]);
            y[4 * j + 1] = __builtin_fmaf(w.y, vk, y[4 * j + 1]);
            y[4 * j + 2] = __builtin_fmaf(w.z, vk, y[4 * j + 2]);
            y[4 * j + 3] = __builtin_fmaf(w.w, vk, y[4 * j + 3]);
        }
    }
}

__device__ __forceinline__ void put_row(float* __restrict__ vec, const float (&v)[D]) {
#pragma unroll
    for (int j = 0; j < D; ++j) vec[j] = v[j];
}

__device__ __forceinline__ void load_row_lm(const float* __restrict__ x, int64_t N, int64_t r, bool on, float* __restrict__ vec) {
    const float2* p = reinterpret_cast<const float2*>(x) + r;
#pragma unroll
    for (int j = 0; j < D / 2; ++j) {
        const float2 q = on ? p[(int64_t)j * N] : make_float2(0.f, 0.f);
        vec[2 * j] = q.x; vec[2 * j + 1] = q.y;
    }
}

__device__ __forceinline__ void load_row(const float* __restrict__ x, int64_t r, bool on, float* __restrict__ vec) {
    const float4* p = reinterpret_cast<const float4*>(x + r * D);
#pragma unroll
    for (int j = 0; j < D / 4; ++j) {
        const float4 q = on ? p[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        vec[4 * j] = q.x; vec[4 * j + 1] = q.y; vec[4 * j + 2] = q.z; vec[4 * j + 3] = q.w;
    }
}

__device__ __forceinline__ uint32_t relu_inplace(float (&y)[D]) {
    uint32_t m = 0u;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        m |= y[j] > 0.0f ? (1u << j) : 0u;
        y[j] = fmaxf(y[j], 0.0f);
    }
    return m;
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Y[64][32] = X[64][0..K) . B[K][32] on the matrix core, tiles [64][TP] in LDS, B row-major [k][32] in LDS:
// v_mfma_f32_32x32x2_f32 with the tile rows as M (two 32-row halves), operands read straight from LDS
// (A: lane = row, k = 2 s + (lane >> 5); B: lane = column).  Both halves are accumulated before anything is written, so `out` may
// alias `in`.  MODE 0: plain, 1: ReLU, 2: masked by gate > 0 (the ReLU derivative of the layer whose output tile is `gate`).
// The one-lane-per-row product it replaces cost 0.27 of the backward kernel's 0.54 ms (tools/build_variant.sh -DGS_TEX_ABL=2).
template <int MODE>
__device__ __forceinline__ void mfma_layer(const float* __restrict__ in, const float* __restrict__ B, int K, float* __restrict__ out,
                                           const float* __restrict__ gate, int lane) {
    const int m = lane & 31, kh = lane >> 5;
    f32x16 acc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][r] = 0.0f;
        const float* ap = in + (32 * h + m) * TP + kh;
        const float* bp = B + kh * D + m;
#pragma unroll 4
        for (int s = 0; s < K / 2; ++s) acc[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s], bp[2 * s * D], acc[h], 0, 0, 0);
    }
    wave_sync();
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pos = (32 * h + (r & 3) + 8 * (r >> 2) + 4 * kh) * TP + m;
            float v = acc[h][r];
            if (MODE == 1) v = fmaxf(v, 0.0f);
            if (MODE == 2) v = gate[pos] > 0.0f ? v : 0.0f;
            out[pos] = v;
        }
    wave_sync();
}

__global__ void __launch_bounds__(256) k_texmlp_fwd(TexArgs A) {
    __shared__ __attribute__((aligned(16))) float s_w1t[D * D], s_w2t[D * D], s_w3[CMAX * D];
    __shared__ float s_lo[CMAX], s_hi[CMAX];
    __shared__ float s_tile[4][TILE];
    load_weights(A, nullptr, nullptr, s_w1t, s_w2t, s_w3, s_lo, s_hi);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t r = gi;
    bool on, valid;
    if (A.rows) {                 // compact rows: rows outside the list are the caller's (pre-filled with the zero-feature value)
        on = valid = gi < A.count_dev[0];
        r = on ? A.rows[gi] : 0;
    } else {
        valid = r < A.N;
        on = valid && (!A.mask || A.mask[r] > 0.0f);
    }
    if (__ballot(on) == 0ull) {   // zero features -> zero logits -> sigmoid = 1/2
        if (valid)
            for (int c = 0; c < A.C; ++c) A.out[r * A.C + c] = 0.5f * (s_hi[c] - s_lo[c]) + s_lo[c];
        return;
    }
    float* vec = s_tile[wave] + lane * TP;
    float y[D];
    if (A.level_major) load_row_lm(A.x, A.N, r, on, vec);
    else load_row(A.x, r, on, vec);
    if (GS_TEX_MFMA) {
        wave_sync();
        mfma_layer<1>(s_tile[wave], s_w1t, D, s_tile[wave], nullptr, lane);
        mfma_layer<1>(s_tile[wave], s_w2t, D, s_tile[wave], nullptr, lane);
#pragma unroll
        for (int j = 0; j < D; ++j) y[j] = vec[j];
    } else {
        matvec(s_w1t, vec, D, y);
        relu_inplace(y);
        put_row(vec, y);
        matvec(s_w2t, vec, D, y);
        relu_inplace(y);
    }
    if (!valid) return;
    for (int c = 0; c < A.C; ++c) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < D; ++j) acc = __builtin_fmaf(s_w3[c * D + j], y[j], acc);
        const float sg = 1.0f / (1.0f + __expf(-acc));
        A.out[r * A.C + c] = (on ? sg : 0.5f) * (s_hi[c] - s_lo[c]) + s_lo[c];
    }
}

// acc[32x32] += sum over the wave's 64 rows of G[row][i] * H[row][j]; tiles are [64][TP] in LDS
__device__ __forceinline__ void outer_accumulate(const float* __restrict__ tg, const float* __restrict__ th, int lane, f32x16& acc) {
    if (GS_TEX_ABL & 1) return;
    const int k = lane >> 5, c = lane & 31;
#pragma unroll 8
    for (int s = 0; s < 32; ++s) {
        const float a = tg[(2 * s + k) * TP + c];
        const float b = th[(2 * s + k) * TP + c];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
}

// one wave per block: 4 tiles (x, h1, h2, gradient) = 33.8 KB + 17 KB of weights -> 3 blocks per CU
__global__ void __launch_bounds__(64) k_texmlp_bwd(TexArgs A, int64_t n_chunks_in) {
    int64_t n_chunks = n_chunks_in;
    __shared__ __attribute__((aligned(16))) float s_w1[D * D], s_w2[D * D], s_w1t[D * D], s_w2t[D * D], s_w3[CMAX * D];
    __shared__ float s_lo[CMAX], s_hi[CMAX];
    __shared__ float s_tile[4][TILE];
    load_weights(A, s_w1, s_w2, s_w1t, s_w2t, s_w3, s_lo, s_hi);
    const int lane = threadIdx.x;
    float* tx = s_tile[0];
    float* t1 = s_tile[1];
    float* t2 = s_tile[2];
    float* tg = s_tile[3];
    f32x16 acc1 = {0}, acc2 = {0}, acc3 = {0};
    if (A.rows) n_chunks = (A.count_dev[0] + 63) / 64;
    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        int64_t r = chunk * 64 + lane;
        bool on;
        if (A.rows) {
            on = r < A.count_dev[0];
            r = on ? A.rows[r] : A.N;            // A.N = "no row": nothing below stores for it
        } else {
            on = r < A.N && (!A.mask || A.mask[r] > 0.0f);
        }
        if (__ballot(on) == 0ull) {
            if (r < A.N && A.g_x && !A.level_major) {   // level major: rows with mask <= 0 are never read by the encoding's backward
                float4* gx = reinterpret_cast<float4*>(A.g_x + r * D);
                for (int j = 0; j < D / 4; ++j) gx[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            continue;
        }
        float y[D];
        // forward, keeping x, h1, h2 in the tiles
        if (A.level_major) load_row_lm(A.x, A.N, r, on, tx + lane * TP);
        else load_row(A.x, r, on, tx + lane * TP);
        uint32_t m1 = 0u, m2 = 0u;
        if (GS_TEX_MFMA) {
            wave_sync();
            mfma_layer<1>(tx, s_w1t, D, t1, nullptr, lane);
            mfma_layer<1>(t1, s_w2t, D, t2, nullptr, lane);
#pragma unroll
            for (int j = 0; j < D; ++j) y[j] = t2[lane * TP + j];
        } else {
            matvec(s_w1t, tx + lane * TP, D, y);
            m1 = relu_inplace(y);
            put_row(t1 + lane * TP, y);
            matvec(s_w2t, t1 + lane * TP, D, y);
            m2 = relu_inplace(y);
            put_row(t2 + lane * TP, y);
        }
        // d loss / d logits
        for (int c = 0; c < CMAX; ++c) {
            float a = 0.0f;
#pragma unroll
            for (int j = 0; j < D; ++j) a = __builtin_fmaf(s_w3[c * D + j], y[j], a);
            const float sg = 1.0f / (1.0f + __expf(-a));
            const float g = (on && c < A.C) ? A.g_out[r * A.C + c] : 0.0f;
            tg[lane * TP + c] = g * (s_hi[c] - s_lo[c]) * sg * (1.0f - sg);
        }
        for (int c = CMAX; c < D; ++c) tg[lane * TP + c] = 0.0f;
        wave_sync();
        // layer 3:  dW3 += go (x) h2 ;  g2 = relu'(h2) * W3^T go
        outer_accumulate(tg, t2, lane, acc3);
        if (GS_TEX_MFMA) {
            mfma_layer<2>(tg, s_w3, CMAX, tg, t2, lane);           // in place: both halves are accumulated before the tile is rewritten
        } else {
            matvec(s_w3, tg + lane * TP, CMAX, y);
#pragma unroll
            for (int j = 0; j < D; ++j) y[j] = (m2 >> j) & 1u ? y[j] : 0.0f;
            wave_sync();
            put_row(tg + lane * TP, y);
            wave_sync();
        }
        // layer 2:  dW2 += g2 (x) h1 ;  g1 = relu'(h1) * W2^T g2
        outer_accumulate(tg, t1, lane, acc2);
        if (GS_TEX_MFMA) {
            mfma_layer<2>(tg, s_w2, D, tg, t1, lane);
        } else {
            matvec(s_w2, tg + lane * TP, D, y);
#pragma unroll
            for (int j = 0; j < D; ++j) y[j] = (m1 >> j) & 1u ? y[j] : 0.0f;
            wave_sync();
            put_row(tg + lane * TP, y);
            wave_sync();
        }
        // layer 1:  dW1 += g1 (x) x ;  g_x = W1^T g1
        outer_accumulate(tg, tx, lane, acc1);
        if (A.g_x) {
            if (GS_TEX_MFMA) {
                mfma_layer<0>(tg, s_w1, D, tx, nullptr, lane);     // x is dead after dW1: its tile takes g_x
#pragma unroll
                for (int j = 0; j < D; ++j) y[j] = tx[lane * TP + j];
            } else {
                matvec(s_w1, tg + lane * TP, D, y);
            }
            if (r < A.N && A.level_major) {
                if (on) {
                    float2* gx = reinterpret_cast<float2*>(A.g_x) + r;
#pragma unroll
                    for (int j = 0; j < D / 2; ++j) gx[(int64_t)j * A.N] = make_float2(y[2 * j], y[2 * j + 1]);
                }
            } else if (r < A.N) {
                float4* gx = reinterpret_cast<float4*>(A.g_x + r * D);
#pragma unroll
                for (int j = 0; j < D / 4; ++j) gx[j] = make_float4(y[4 * j], y[4 * j + 1], y[4 * j + 2], y[4 * j + 3]);
            }
        }
        wave_sync();
    }
    // flush: D[row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = lane&31]
    if (GS_TEX_ABL & 4) return;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), col = lane & 31;
        if (A.g_w1 && acc1[reg] != 0.0f) atomicAdd(&A.g_w1[row * D + col], acc1[reg]);
        if (A.g_w2 && acc2[reg] != 0.0f) atomicAdd(&A.g_w2[row * D + col], acc2[reg]);
        if (A.g_w3 && row < A.C && acc3[reg] != 0.0f) atomicAdd(&A.g_w3[row * D + col], acc3[reg]);
    }
}

}  // namespace

static int texmlp_fwd(const float* x, int level_major, const int32_t* rows, const int64_t* count_dev, int64_t cap, const float* mask, int64_t N, const float* w1, const float* w2, const float* w3, int C,
                      const float* lo, const float* hi, float* out, gs_stream_t stream) {
    if (N == 0) return 0;
    GS_REQUIRE(x && w1 && w2 && w3 && lo && hi && out, "gs_texmlp_fwd: null pointer");
    GS_REQUIRE(C >= 1 && C <= CMAX, "gs_texmlp_fwd: 1..8 output channels");
    TexArgs A{};
    A.x = x; A.mask = mask; A.N = N; A.w1 = w1; A.w2 = w2; A.w3 = w3; A.C = C; A.lo = lo; A.hi = hi; A.out = out;
    A.level_major = level_major; A.rows = rows; A.count_dev = count_dev;
    hipLaunchKernelGGL(k_texmlp_fwd, dim3((unsigned)gs::cdiv(rows ? cap : N, 256)), dim3(256), 0, (hipStream_t)stream, A);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_texmlp_fwd(const float* x, const float* mask, int64_t N, const float* w1, const float* w2, const float* w3, int C,
                             const float* lo, const float* hi, float* out, gs_stream_t stream) {
    return texmlp_fwd(x, 0, nullptr, nullptr, 0, mask, N, w1, w2, w3, C, lo, hi, out, stream);
}

extern "C" int gs_texmlp_fwd_level_major(const float* x, const float* mask, int64_t N, const float* w1, const float* w2, const float* w3, int C,
                                         const float* lo, const float* hi, float* out, gs_stream_t stream) {
    return texmlp_fwd(x, 1, nullptr, nullptr, 0, mask, N, w1, w2, w3, C, lo, hi, out, stream);
}

extern "C" int gs_texmlp_fwd_rows(const float* x_level_major, const int32_t* rows, const int64_t* count_dev, int64_t cap, int64_t N, const float* w1,
                                  const float* w2, const float* w3, int C, const float* lo, const float* hi, float* out, gs_stream_t stream) {
    GS_REQUIRE(rows && count_dev && cap >= 0 && cap <= N, "gs_texmlp_fwd_rows: row list missing");
    if (cap == 0) return 0;
    return texmlp_fwd(x_level_major, 1, rows, count_dev, cap, nullptr, N, w1, w2, w3, C, lo, hi, out, stream);
}

static int texmlp_bwd(const float* x, int level_major, const int32_t* rows, const int64_t* count_dev, int64_t cap, const float* mask, int64_t N, const float* w1, const float* w2, const float* w3, int C,
                      const float* lo, const float* hi, const float* g_out, float* g_x, float* g_w1, float* g_w2, float* g_w3, gs_stream_t stream) {
    if (N == 0) return 0;
    GS_REQUIRE(x && w1 && w2 && w3 && lo && hi && g_out, "gs_texmlp_bwd: null pointer");
    GS_REQUIRE(C >= 1 && C <= CMAX, "gs_texmlp_bwd: 1..8 output channels");
    TexArgs A{};
    A.x = x; A.mask = mask; A.N = N; A.w1 = w1; A.w2 = w2; A.w3 = w3; A.C = C; A.lo = lo; A.hi = hi;
    A.g_out = g_out; A.g_x = g_x; A.g_w1 = g_w1; A.g_w2 = g_w2; A.g_w3 = g_w3;
    A.level_major = level_major; A.rows = rows; A.count_dev = count_dev;
    const int64_t n_chunks = gs::cdiv(rows ? cap : N, 64);
    const int64_t blocks = std::min<int64_t>(n_chunks, 768);
    hipLaunchKernelGGL(k_texmlp_bwd, dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, A, n_chunks);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_texmlp_bwd(const float* x, const float* mask, int64_t N, const float* w1, const float* w2, const float* w3, int C,
                             const float* lo, const float* hi, const float* g_out, float* g_x, float* g_w1, float* g_w2, float* g_w3,
                             gs_stream_t stream) {
    return texmlp_bwd(x, 0, nullptr, nullptr, 0, mask, N, w1, w2, w3, C, lo, hi, g_out, g_x, g_w1, g_w2, g_w3, stream);
}

extern "C" int gs_texmlp_bwd_level_major(const float* x, const float* mask, int64_t N, const float* w1, const float* w2, const float* w3, int C,
                                         const float* lo, const float* hi, const float* g_out, float* g_x, float* g_w1, float* g_w2, float* g_w3,
                                         gs_stream_t stream) {
    return texmlp_bwd(x, 1, nullptr, nullptr, 0, mask, N, w1, w2, w3, C, lo, hi, g_out, g_x, g_w1, g_w2, g_w3, stream);
}

extern "C" int gs_texmlp_bwd_rows(const float* x_level_major, const int32_t* rows, const int64_t* count_dev, int64_t cap, int64_t N, const float* w1,
                                  const float* w2, const float* w3, int C, const float* lo, const float* hi, const float* g_out, float* g_x_level_major,
                                  float* g_w1, float* g_w2, float* g_w3, gs_stream_t stream) {
    GS_REQUIRE(rows && count_dev && cap >= 0 && cap <= N, "gs_texmlp_bwd_rows: row list missing");
    if (cap == 0) return 0;
    return texmlp_bwd(x_level_major, 1, rows, count_dev, cap, nullptr, N, w1, w2, w3, C, lo, hi, g_out, g_x_level_major, g_w1, g_w2, g_w3, stream);
}

// ---- compile-time variants of this file (common.hpp): non-default values announce themselves through gs_build_flags(); switches that give
// wrong results (timing-only ablations) compile only under -DGS_EXPERIMENT
GS_TUNABLE(GS_TEX_ABL, 0)
GS_TUNABLE(GS_TEX_MFMA, 1)
#if GS_TEX_ABL != 0
GS_EXPERIMENT_ONLY(GS_TEX_ABL)
#endif

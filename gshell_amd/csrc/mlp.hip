// Fused forward pass of the G-Shell SDF network on gfx950 matrix cores (fp32-in / fp32-accumulate MFMA).
//
// Replaces, for the full-grid forward evaluation (geometry/gshell_tets_geometry.py:194 `sdf = self.sdf_net(v_deformed)`,
// N = 2.28 M grid vertices at tet-res 256), the reference's module geometry/mlp.py:7-40 + geometry/embedding.py:22-39:
//   emb = (x, sin(2^k x), cos(2^k x))_{k < n_freq}                      [3 (2 n_freq + 1)]
//   h   = softplus_100(W0 emb + b0);  h = softplus_100(W_i [h | emb if i == skip] + b_i)  for the hidden layers;
//   sdf = w_out . h + b_out
// The reference (and a plain torch port) runs 15 GEMM + ~25 elementwise launches per pass and streams every
// [N,256] fp32 activation through HBM (2.3 GB each at res 256: ~65 GB of traffic for ~1.9 TFLOP of work).  Here ONE
// workgroup carries a 64-row tile through ALL layers:
//   * activations ping-pong between two LDS tiles [64][257] fp32 (row stride 257 -> conflict-free column reads),
//     the positional encoding stays in a third LDS tile for the skip connection; nothing but x[N,3] is read from
//     and sdf[N] written to HBM (16 B / vertex);
//   * weights (1.6 MB, L2-resident, pre-transposed k-major by the caller) are read as MFMA B-fragments directly from
//     global memory with a register double buffer (no LDS stage, no barrier inside the k-loop); 8 waves each own a
//     64x32 output block = 2 accumulators of v_mfma_f32_32x32x2_f32 (exact fp32, bitwise a k-ordered fmaf chain);
//     two waves per SIMD cover each other's LDS / L1 latencies (MI355X_MICROARCH: 64-cycle issue = latency);
//   * bias + Softplus(beta = 100, threshold 20) are applied on the accumulators in registers.
// Roofline: MFMA fp32, 826 880 flop / vertex (d_hidden 256, 6 hidden layers, skip at 3) against 157 TFLOP/s.
// The backward pass is row-sparse and lives in gshell_amd/geometry/mlp.py (only ~10 % of the rows carry gradient).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/gshell_hip.h"
#include "common.hpp"

#if GS_ORACLE_KERNELS
namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

#ifndef GS_MLP_SUB
#define GS_MLP_SUB 2
#endif
constexpr int SUB = GS_MLP_SUB; // 32-row sub-tiles per workgroup = accumulators per wave
constexpr int TM = 32 * SUB;    // rows per workgroup
constexpr int NT = 512;         // threads per workgroup: 8 waves, each owns TM rows x 32 columns (SUB accumulators)
constexpr int D = 256;          // hidden width (fixed by the kernel)
constexpr int LDX = D + 1;      // activation tile row stride (floats)
constexpr int EMAX = 40;        // max embedding width handled (3 (2*6 + 1) = 39, padded to even)
constexpr int LDE = EMAX + 1;
constexpr int KC = 8;           // weight rows per register-prefetch chunk
constexpr int MAX_LAYERS = 16;
#ifndef GS_MLP_INPLACE
#define GS_MLP_INPLACE 1
#endif
#define GS_MLP_INPLACE_TILES (GS_MLP_INPLACE ? 1 : 2)

struct MlpArgs {
    const float* x;       // [N,3]
    float* out;           // [N]
    int64_t N;
    int n_freq, E, Epad;  // E = 3 (2 n_freq + 1), Epad = E rounded up to a multiple of KC
    int n_layers;         // hidden-producing layers (first + n_hidden)
    int skip_layer;       // index (>= 1) of the layer whose input is [h | emb], or -1
    const float* wt[MAX_LAYERS];    // k-major weights [K_l (padded), 256]; K_0 = Epad, K_l = 256 (+ Epad for the skip layer)
    const float* bias[MAX_LAYERS];  // [256]
    const float* w_out;             // [256] followed by the output bias
};

__device__ __forceinline__ float softplus100(float x) {
    // hardware exp/log (v_exp_f32 / v_log_f32): absolute error <= 1e-9 on the softplus value, ~10 VALU ops instead of ~80
    float bx = x * 100.0f;
    return bx > 20.0f ? x : __logf(1.0f + __expf(bx)) * 0.01f;
}

// acc += A[64 x K] (LDS, row stride lda) * Wt[K x 256] (global, k-major) restricted to this wave's 64 columns.
// B fragments come straight from global memory (the 1.6 MB of weights are L2-resident and every wave of every CU walks the
// same rows, so the per-CU L1 serves most of them): no LDS staging and therefore NO barrier inside the k-loop -- the
// waves of a workgroup drift freely and keep the matrix pipes busy.  Loads for the next 8 k-rows are issued
// before the MFMAs of the current 8 (register double buffer, 16 VGPRs); sched_barrier pins that order.
__device__ __forceinline__ void gemm_segment(v16f (&acc)[SUB], const float* __restrict__ a_lds, int lda, int K, const float* __restrict__ wt, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int arow = lane & 31, ak = lane >> 5;
    const int nchunks = K / KC;
    const float* wp = wt + (int64_t)ak * D + wave * 32 + arow;     // B[k = 2s + ak][n = wave*32 + arow]
    float b[KC / 2], nb[KC / 2];
#pragma unroll
    for (int s = 0; s < KC / 2; ++s) b[s] = wp[(2 * s) * D];
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        const float* wn = wp + (int64_t)(c + 1) * KC * D;
        if (more) {
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) nb[s] = wn[(2 * s) * D];
        }
        __builtin_amdgcn_sched_barrier(0);
        const float* acur = a_lds + arow * lda + c * KC + ak;
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {
#pragma unroll
            for (int i = 0; i < SUB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[i * 32 * lda + 2 * s], b[s], acc[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) b[s] = nb[s];
        }
    }
}

__global__ void __launch_bounds__(NT, 2) k_sdf_mlp_fwd(MlpArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xa = smem;                       // [TM][LDX]
    float* xb = xa + TM * LDX;              // [TM][LDX]   (absent with GS_MLP_INPLACE)
    float* emb = xa + (GS_MLP_INPLACE_TILES) * TM * LDX;   // [TM][LDE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * TM;

    // positional encoding of the tile -> emb (zero padded to Epad columns, zero rows past N)
    for (int idx = tid; idx < TM * A.Epad; idx += NT) {
        int row = idx / A.Epad, f = idx - row * A.Epad;
        int64_t r = r0 + row;
        float v = 0.f;
        if (r < A.N && f < A.E) {
            if (f < 3)
                v = A.x[3 * r + f];
            else {
                int g = f - 3, k = g / 6, sc = (g % 6) / 3, c = g % 3;
                float arg = (float)(1 << k) * A.x[3 * r + c];
                v = sc ? cosf(arg) : sinf(arg);
            }
        }
        emb[row * LDE + f] = v;
    }
    __syncthreads();

#ifndef GS_MLP_INPLACE
#define GS_MLP_INPLACE 1
#endif
    // GS_MLP_INPLACE: ONE activation tile, overwritten in place after a barrier (k-loop of layer l done by all waves ->
    // epilogue writes layer l's outputs over its inputs).  76 KB of LDS instead of 142 KB, so TWO workgroups share a CU
    // (4 waves per SIMD) and the epilogue of one overlaps the k-loop of the other: in-phase execution of two co-resident
    // workgroups is unstable (whoever leaves the k-loop first lets the other run it at full rate), so they drift apart.
    float* xin = xa;
    float* xout = GS_MLP_INPLACE ? xa : xb;
    for (int l = 0; l < A.n_layers; ++l) {
        v16f acc[SUB];
#pragma unroll
        for (int i = 0; i < SUB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        if (l == 0) {
            gemm_segment(acc, emb, LDE, A.Epad, A.wt[0], tid);
        } else {
            gemm_segment(acc, xin, LDX, D, A.wt[l], tid);
            if (l == A.skip_layer) gemm_segment(acc, emb, LDE, A.Epad, A.wt[l] + (int64_t)D * D, tid);
        }
        if (GS_MLP_INPLACE) __syncthreads();     // every wave is done reading the tile
        // epilogue: bias + softplus, accumulator layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        int col = wave * 32 + (lane & 31);
        float bj = A.bias[l][col];
#pragma unroll
        for (int i = 0; i < SUB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                xout[row * LDX + col] = softplus100(acc[i][r] + bj);
            }
        __syncthreads();
        float* t = xin;
        xin = xout;
        xout = t;
    }
    // output layer: 8 lanes per row, 32 columns each
    if (tid < TM * 8) {
        int row = tid >> 3, q = tid & 7;
        float s = 0.f;
        const float* h = xin + row * LDX + q * 32;
        const float* w = A.w_out + q * 32;
        for (int c = 0; c < 32; ++c) s += h[c] * w[c];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        int64_t r = r0 + row;
        if (q == 0 && r < A.N) A.out[r] = s + A.w_out[D];
    }
}

constexpr size_t SMEM_BYTES = (size_t)((GS_MLP_INPLACE ? 1 : 2) * TM * LDX + TM * LDE) * sizeof(float);

}  // namespace

#endif  // GS_ORACLE_KERNELS

extern "C" int64_t gs_sdf_mlp_packed_floats(int n_freq, int n_hidden, int skip_layer) {
    constexpr int D = 256, KC = 8;      // the packed layout (python packs it; the kernel that reads it is an oracle kernel)
    int E = 3 * (2 * n_freq + 1);
    int Epad = (E + KC - 1) / KC * KC;
    int64_t n = (int64_t)Epad * D + D;                       // layer 0 weights + bias
    for (int i = 1; i <= n_hidden; ++i) n += (int64_t)(D + (i == skip_layer ? Epad : 0)) * D + D;
    return n + D + 1;                                        // output weights + bias
}

// packed = [ Wt_0 (Epad x 256, k-major, zero padded) | b_0 | Wt_1 | b_1 | ... | w_out (256) | b_out ]
// with Wt_l = transpose of torch's Linear.weight [256, K_l]; for the skip layer the K axis is [h (256) | emb (Epad)].
#if GS_ORACLE_KERNELS
extern "C" int gs_sdf_mlp_fwd(const float* x, int64_t N, const float* packed, int n_freq, int n_hidden, int skip_layer, float* out,
                              gs_stream_t stream) {
    if (N == 0) return 0;
    GS_REQUIRE(x && packed && out, "gs_sdf_mlp_fwd: null pointer");
    int E = 3 * (2 * n_freq + 1);
    int Epad = (E + KC - 1) / KC * KC;
    GS_REQUIRE(n_freq >= 0 && n_freq <= 10 && Epad <= EMAX, "gs_sdf_mlp_fwd: positional encoding wider than 40 is not supported");
    GS_REQUIRE(n_hidden >= 0 && n_hidden + 1 <= MAX_LAYERS, "gs_sdf_mlp_fwd: too many layers");
    GS_REQUIRE(skip_layer == -1 || (skip_layer >= 1 && skip_layer <= n_hidden), "gs_sdf_mlp_fwd: bad skip layer");
    MlpArgs A{};
    A.x = x; A.out = out; A.N = N; A.n_freq = n_freq; A.E = E; A.Epad = Epad; A.n_layers = n_hidden + 1; A.skip_layer = skip_layer;
    const float* p = packed;
    A.wt[0] = p; p += (int64_t)Epad * D;
    A.bias[0] = p; p += D;
    for (int i = 1; i <= n_hidden; ++i) {
        A.wt[i] = p; p += (int64_t)(D + (i == skip_layer ? Epad : 0)) * D;
        A.bias[i] = p; p += D;
    }
    A.w_out = p;
    // per launch: the attribute is per device, and a process-wide "done" flag would be neither per-device nor thread-safe
    GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sdf_mlp_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_BYTES));
    hipLaunchKernelGGL(k_sdf_mlp_fwd, dim3((unsigned)gs::cdiv(N, TM)), dim3(NT), SMEM_BYTES, (hipStream_t)stream, A);
    GS_LAUNCH_CHECK();
    return 0;
}
#else
extern "C" int gs_sdf_mlp_fwd(const float* x, int64_t N, const float* packed, int n_freq, int n_hidden, int skip_layer, float* out,
                              gs_stream_t stream) {
    GS_ORACLE_ONLY("gs_sdf_mlp_fwd");
}
#endif

// ---- compile-time variants of this file (common.hpp): non-default values announce themselves through gs_build_flags(); switches that give
// wrong results (timing-only ablations) compile only under -DGS_EXPERIMENT
#if GS_ORACLE_KERNELS
GS_TUNABLE(GS_MLP_SUB, 2)
GS_TUNABLE(GS_MLP_INPLACE, 1)
#endif

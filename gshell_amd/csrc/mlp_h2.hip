// SDF network of G-Shell on the gfx950 f16 matrix path with fp32-class accuracy ("h2": every fp32 operand is carried
// as a PAIR of fp16 pieces, v = hi + lo / 2048, and every product as three MFMAs).
//
// Why: the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, csrc/mlp.hip) runs at the fp32 VECTOR rate, 157 TFLOP/s -- 1/16 of
// the f16 / bf16 matrix rate (MI355X_MICROARCH.md "Matrix cores").  The full-grid forward pass of geometry/mlp.py:32-40
// (2.28 M rows x 826 880 flop at tet-res 256) is therefore a 12 ms floor on that instruction and was 31 % of the training
// iteration.  Splitting  a = a1 + a2/2048,  w = w1 + w2/2048  with a1 = fp16(a), a2 = fp16((a - a1) 2048) represents both
// operands to 2^-22 relative (fp32: 2^-24) and
//        a w  =  a1 w1  +  (a1 w2 + a2 w1) / 2048  +  O(2^-22 |a w|)
// costs THREE v_mfma_f32_32x32x16_f16 (fp32 accumulate; fp16 x fp16 products are exact in fp32) = 3/16 of the fp32
// instruction's time.  The 1/2048 keeps the low pieces in the normal fp16 range; they get their own accumulator, folded
// into the high one in the epilogue.  Values below the fp16 normal range go entirely into the scaled low piece.
// Measured error against the exact-fp32 kernel and float64: see DESIGN.md (tools/mlp_precision.py).
//
// Shape of the kernel (same dataflow as csrc/mlp.hip): one 512-thread workgroup carries a 64-row tile through ALL layers;
// activations live in LDS as two fp16 planes [64][256 (+8 pad)], overwritten in place between two barriers; 80 KB of LDS ->
// two workgroups per CU, whose phases drift apart so that one's softplus epilogue (VALU) overlaps the other's k-loop
// (matrix pipe).  The GEMMs are computed TRANSPOSED, D[feature][row] = W[feature][k] . H[row][k]^T:
//   * A operand = weights, pre-packed once per call into "fragment-major" order (the 16 bytes lane l of wave w needs
//     at k-step s are contiguous with its neighbours': one fully coalesced 1 KB global_load_b128 per piece per k-step),
//   * B operand = activations, ds_read_b128 of 8 consecutive k of one row (row stride 33 x 16 B: conflict-free),
//   * each lane of the accumulator then holds 16 features of ONE row in groups of four consecutive features -> the
//     epilogue writes the next layer's input with ds_write_b64 (the untransposed form would need 2-byte scatter writes).
// Wave w owns features [32 w, 32 w + 32) x 64 rows = 2 (row halves) x 2 (hi, lo) accumulators.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/gshell_hip.h"
#include "common.hpp"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int TM = 64;            // rows per workgroup
constexpr int NT = 512;           // 8 waves
constexpr int D = 256;            // hidden width
constexpr int LDH = D + 8;        // activation plane row stride (halfs): 528 B = 33 sixteen-byte slots
constexpr int EK = 48;            // embedding width padded to a multiple of the MFMA K (39 -> 48)
constexpr int LDEH = EK + 8;      // 112 B = 7 slots
constexpr int MAX_LAYERS = 16;
constexpr float LO_SCALE = 2048.0f, LO_INV = 1.0f / 2048.0f;
constexpr float H_MIN_NORMAL = 6.103515625e-05f;   // 2^-14

struct H2Args {
    const float* x;       // [N,3]
    float* out;           // [N]
    int64_t N;
    int n_freq, E;
    int n_layers;         // hidden-producing layers (first + n_hidden)
    int skip_layer;       // index (>= 1) of the layer whose input is [h | emb], or -1
    const h8* wfrag[MAX_LAYERS];    // fragment-major weights: [k-step][wave 8][piece 2][lane 64] x 16 B
    const float* bias[MAX_LAYERS];  // [256] fp32
    const float* w_out;             // [256] fp32 followed by the output bias
};

__device__ __forceinline__ void split_h2(float v, _Float16& hi, _Float16& lo) {
    v = fminf(fmaxf(v, -60000.0f), 60000.0f);
    _Float16 h = (_Float16)v;                              // round to nearest even
    if (fabsf(v) < H_MIN_NORMAL) h = (_Float16)0.0f;       // no reliance on fp16 denormals in the matrix pipe
    hi = h;
    lo = (_Float16)((v - (float)h) * LO_SCALE);
}

__device__ __forceinline__ float softplus100(float x) {
    // hardware exp/log (v_exp_f32 / v_log_f32): absolute error <= 1e-9 on the softplus value (same formula as csrc/mlp.hip)
    float bx = x * 100.0f;
    return bx > 20.0f ? x : __logf(1.0f + __expf(bx)) * 0.01f;
}

// hi/lo += W[32 features of this wave][K] . P[64 rows][K]^T over `nsteps` k-steps of 16.
// Weight fragments for step s+1 are in flight while the six MFMAs of step s issue (register double buffer); the LDS
// fragments of a step are read just before its MFMAs -- the other three waves of the SIMD cover that latency.
template <int STRIDE>
__device__ __forceinline__ void gemm_seg(v16f (&hi)[2], v16f (&lo)[2], const _Float16* __restrict__ P1, const _Float16* __restrict__ P2,
                                         int nsteps, const h8* __restrict__ wf, int wave, int lane) {
    const int row = lane & 31, kq = lane >> 5;
    const _Float16* b1p = P1 + row * STRIDE + kq * 8;
    const _Float16* b2p = P2 + row * STRIDE + kq * 8;
    const h8* wp = wf + wave * 128 + lane;       // + step * 1024 (+64 for the low piece)
    h8 a1 = wp[0], a2 = wp[64];
    for (int st = 0; st < nsteps; ++st) {
        const h8 b10 = *reinterpret_cast<const h8*>(b1p + st * 16);
        const h8 b11 = *reinterpret_cast<const h8*>(b1p + 32 * STRIDE + st * 16);
        const h8 b20 = *reinterpret_cast<const h8*>(b2p + st * 16);
        const h8 b21 = *reinterpret_cast<const h8*>(b2p + 32 * STRIDE + st * 16);
        h8 n1 = a1, n2 = a2;
        if (st + 1 < nsteps) {
            n1 = wp[(st + 1) * 1024];
            n2 = wp[(st + 1) * 1024 + 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        hi[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b10, hi[0], 0, 0, 0);
        hi[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b11, hi[1], 0, 0, 0);
        lo[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b20, lo[0], 0, 0, 0);
        lo[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b21, lo[1], 0, 0, 0);
        lo[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b10, lo[0], 0, 0, 0);
        lo[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b11, lo[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a1 = n1;
        a2 = n2;
    }
}

__global__ void __launch_bounds__(NT, 4) k_sdf_mlp_fwd_h2(H2Args A) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    _Float16* H1 = smem_h;                   // [TM][LDH]
    _Float16* H2 = H1 + TM * LDH;            // [TM][LDH]
    _Float16* E1 = H2 + TM * LDH;            // [TM][LDEH]
    _Float16* E2 = E1 + TM * LDEH;           // [TM][LDEH]   (the output reduction scratch is overlaid on E1/E2 at the end)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * TM;

    // positional encoding of the tile (x, sin(2^k x), cos(2^k x))_k, zero padded to EK columns, zero rows past N
    for (int idx = tid; idx < TM * EK; idx += NT) {
        int row = idx / EK, f = idx - row * EK;
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
        _Float16 hi, lo;
        split_h2(v, hi, lo);
        E1[row * LDEH + f] = hi;
        E2[row * LDEH + f] = lo;
    }
    __syncthreads();

    const int n_base = wave * 32 + 4 * (lane >> 5);      // + 8 g + j  (g = reg >> 2, j = reg & 3)
    const int m_lane = lane & 31;                        // + 32 s
    for (int l = 0; l < A.n_layers; ++l) {
        v16f hi[2], lo[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) hi[s][r] = lo[s][r] = 0.f;
        if (l == 0) {
            gemm_seg<LDEH>(hi, lo, E1, E2, EK / 16, A.wfrag[0], wave, lane);
        } else {
            gemm_seg<LDH>(hi, lo, H1, H2, D / 16, A.wfrag[l], wave, lane);
            if (l == A.skip_layer) gemm_seg<LDEH>(hi, lo, E1, E2, EK / 16, A.wfrag[l] + (D / 16) * 1024, wave, lane);
        }
        __syncthreads();     // every wave is done reading the planes: they are overwritten in place
        const float* bl = A.bias[l] + n_base;
        const bool last = l + 1 == A.n_layers;
        float part[2] = {0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4*>(bl + 8 * g);
            const float bj[4] = {b4.x, b4.y, b4.z, b4.w};
            float wj[4] = {0.f, 0.f, 0.f, 0.f};
            if (last) {
                const float4 w4 = *reinterpret_cast<const float4*>(A.w_out + n_base + 8 * g);
                wj[0] = w4.x; wj[1] = w4.y; wj[2] = w4.z; wj[3] = w4.w;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                h4 o1, o2;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g + j;
                    float v = softplus100(__builtin_fmaf(lo[s][r], LO_INV, hi[s][r]) + bj[j]);
                    if (last) {
                        part[s] = __builtin_fmaf(v, wj[j], part[s]);
                    } else {
                        _Float16 a, b;
                        split_h2(v, a, b);
                        o1[j] = a;
                        o2[j] = b;
                    }
                }
                if (!last) {
                    const int off = (32 * s + m_lane) * LDH + n_base + 8 * g;
                    *reinterpret_cast<h4*>(H1 + off) = o1;
                    *reinterpret_cast<h4*>(H2 + off) = o2;
                }
            }
        }
        if (last) {
            // output layer: this lane holds sum over its 16 features; add the other 16 of the wave's 32 (lane ^ 32), then
            // the 8 waves through LDS in a fixed order (deterministic)
            float* red = reinterpret_cast<float*>(E1);       // [8 waves][64 rows] fp32 = 2 KB (the embedding planes are dead now)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float p = part[s] + __shfl_xor(part[s], 32, 64);
                if (lane < 32) red[wave * TM + 32 * s + lane] = p;
            }
        }
        __syncthreads();
    }
    if (tid < TM) {
        const float* red = reinterpret_cast<const float*>(E1);
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += red[w * TM + tid];
        int64_t r = r0 + tid;
        if (r < A.N) A.out[r] = s + A.w_out[D];
    }
}

constexpr size_t SMEM_BYTES = (size_t)(2 * TM * LDH + 2 * TM * LDEH) * sizeof(_Float16);
static_assert(SMEM_BYTES <= 80 * 1024, "two workgroups per CU");

// ---- weight packing ------------------------------------------------------------------------------------------------
struct PackArgs {
    const float* w[MAX_LAYERS + 1];   // torch Linear.weight [out, in] of every layer, output layer last
    const float* b[MAX_LAYERS + 1];
    int n_layers, skip_layer, E;
    int64_t frag_off[MAX_LAYERS];     // in h8 units
    int nsteps[MAX_LAYERS];
    int64_t total_frags;              // h8 entries
    h8* frags;
    float* tail;                      // biases [n_layers][256], w_out [256], b_out
};

__global__ void __launch_bounds__(256) k_h2_pack(PackArgs P) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < P.total_frags) {
        int l = 0;
        while (l + 1 < P.n_layers && i >= P.frag_off[l + 1]) ++l;
        int64_t j = i - P.frag_off[l];
        const int lane = (int)(j & 63), piece = (int)((j >> 6) & 1), wave = (int)((j >> 7) & 7), step = (int)(j >> 10);
        const int n = wave * 32 + (lane & 31);
        const int k0 = step * 16 + 8 * (lane >> 5);
        const int Kin = l == 0 ? P.E : (l == P.skip_layer ? D + P.E : D);      // torch row length
        h8 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int k = k0 + q, src = -1;
            if (l == 0) src = k < P.E ? k : -1;
            else if (k < D) src = k;
            else if (l == P.skip_layer && k - D < P.E) src = k;            // [h | emb] order of geometry/mlp.py:37
            float v = src >= 0 ? P.w[l][(int64_t)n * Kin + src] : 0.f;
            _Float16 hi, lo;
            split_h2(v, hi, lo);
            o[q] = piece ? lo : hi;
        }
        P.frags[i] = o;
    }
    const int n_tail = P.n_layers * D + D + 1;
    if (i < n_tail) {
        float v;
        if (i < (int64_t)P.n_layers * D) v = P.b[i / D][i % D];
        else if (i < (int64_t)P.n_layers * D + D) v = P.w[P.n_layers][i - (int64_t)P.n_layers * D];
        else v = P.b[P.n_layers][0];
        P.tail[i] = v;
    }
}

int layer_steps(int l, int skip_layer) { return l == 0 ? EK / 16 : D / 16 + (l == skip_layer ? EK / 16 : 0); }

int check_shape(const char* who, int n_freq, int n_hidden, int skip_layer) {
    int E = 3 * (2 * n_freq + 1);
    GS_REQUIRE(n_freq >= 0 && E <= EK, "sdf_mlp_h2: positional encoding wider than 48 is not supported");
    GS_REQUIRE(n_hidden >= 0 && n_hidden + 1 <= MAX_LAYERS, "sdf_mlp_h2: too many layers");
    GS_REQUIRE(skip_layer == -1 || (skip_layer >= 1 && skip_layer <= n_hidden), "sdf_mlp_h2: bad skip layer");
    (void)who;
    return 0;
}

}  // namespace

extern "C" int64_t gs_sdf_mlp_h2_packed_bytes(int n_freq, int n_hidden, int skip_layer) {
    (void)n_freq;
    int64_t frags = 0;
    for (int l = 0; l <= n_hidden; ++l) frags += (int64_t)layer_steps(l, skip_layer) * 1024;
    return frags * 16 + ((int64_t)(n_hidden + 1) * D + D + 1) * 4;
}

// weights / biases: HOST arrays of n_hidden + 2 DEVICE pointers (torch layout, Linear.weight [out, in] row-major; the
// output layer last).  packed: gs_sdf_mlp_h2_packed_bytes bytes, 16-byte aligned, WRITTEN.
extern "C" int gs_sdf_mlp_h2_pack(const float* const* weights, const float* const* biases, int n_freq, int n_hidden, int skip_layer, void* packed,
                                  gs_stream_t stream) {
    GS_REQUIRE(weights && biases && packed, "gs_sdf_mlp_h2_pack: null pointer");
    GS_REQUIRE(((uintptr_t)packed & 15) == 0, "gs_sdf_mlp_h2_pack: packed buffer must be 16-byte aligned");
    if (int rc = check_shape("pack", n_freq, n_hidden, skip_layer)) return rc;
    PackArgs P{};
    P.n_layers = n_hidden + 1; P.skip_layer = skip_layer; P.E = 3 * (2 * n_freq + 1);
    int64_t off = 0;
    for (int l = 0; l < P.n_layers; ++l) {
        P.frag_off[l] = off;
        P.nsteps[l] = layer_steps(l, skip_layer);
        off += (int64_t)P.nsteps[l] * 1024;
    }
    P.total_frags = off;
    for (int l = 0; l <= P.n_layers; ++l) {
        GS_REQUIRE(weights[l] && biases[l], "gs_sdf_mlp_h2_pack: null layer pointer");
        P.w[l] = weights[l];
        P.b[l] = biases[l];
    }
    P.frags = (h8*)packed;
    P.tail = (float*)((char*)packed + off * 16);
    hipLaunchKernelGGL(k_h2_pack, dim3((unsigned)gs::cdiv(off, 256)), dim3(256), 0, (hipStream_t)stream, P);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_sdf_mlp_fwd_h2(const float* x, int64_t N, const void* packed, int n_freq, int n_hidden, int skip_layer, float* out,
                                 gs_stream_t stream) {
    if (N == 0) return 0;
    GS_REQUIRE(x && packed && out, "gs_sdf_mlp_fwd_h2: null pointer");
    if (int rc = check_shape("fwd", n_freq, n_hidden, skip_layer)) return rc;
    H2Args A{};
    A.x = x; A.out = out; A.N = N; A.n_freq = n_freq; A.E = 3 * (2 * n_freq + 1); A.n_layers = n_hidden + 1; A.skip_layer = skip_layer;
    int64_t off = 0;
    for (int l = 0; l < A.n_layers; ++l) {
        A.wfrag[l] = (const h8*)packed + off;
        off += (int64_t)layer_steps(l, skip_layer) * 1024;
    }
    const float* tail = (const float*)((const char*)packed + off * 16);
    for (int l = 0; l < A.n_layers; ++l) A.bias[l] = tail + (int64_t)l * D;
    A.w_out = tail + (int64_t)A.n_layers * D;
    // per launch (cheap, and correct per device / per thread, unlike a process-wide "done" flag)
    GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sdf_mlp_fwd_h2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_BYTES));
    hipLaunchKernelGGL(k_sdf_mlp_fwd_h2, dim3((unsigned)gs::cdiv(N, TM)), dim3(NT), SMEM_BYTES, (hipStream_t)stream, A);
    GS_LAUNCH_CHECK();
    return 0;
}

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

// Saved planes (layer outputs A, layer adjoints D) of the gradient passes: [layer][32-row slab][256 features][32 rows] fp32.
// Lanes of the MFMA accumulator layout hold (row m, a few features): with the ROW index fastest a store / load of one feature is
// 32 lanes x 4 B = one full 128-byte line (the row-major [row][256] layout made every access 16 bytes at a 1 KB stride: 0.3 of the
// reverse chain's 0.8 ms went into its stores), and the weight-gradient kernel reads 4 consecutive rows of a feature -- exactly
// the 8-byte piece of its transposed LDS image -- with one 16-byte load, 1 KB per wave instruction.
__device__ __forceinline__ int64_t plane_idx(int64_t Rpad, int l, int64_t r, int f) {
    return ((int64_t)l * Rpad + (r & ~(int64_t)31)) * 256 + (int64_t)f * 32 + (r & 31);
}
constexpr int MAX_LAYERS = 16;
constexpr float LO_SCALE = 2048.0f, LO_INV = 1.0f / 2048.0f;
constexpr float H_MIN_NORMAL = 6.103515625e-05f;   // 2^-14

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#ifndef GS_H2_FLUSH
#define GS_H2_FLUSH 0   // 1: values below the fp16 normal range go entirely into the scaled low piece.  Not needed: v_mfma_f32_32x32x16_f16 honours fp16 denormal inputs on gfx950 (tools/micro/mfma_f16_denorm.hip: exact products down to 6e-8)
#endif

// Row sources of the chain kernels.
//   GRID: row r of x [N,3]                              (full-grid forward, nothing saved)
//   ROWS: row rows[r] of x, r < R                       (row-sparse backward: recompute + save)
//   EIK : 16 samples per tile, 4 VIRTUAL rows each: tile row 16 c + i = (value | d/dx | d/dy | d/dz) of sample 16 tile + i.
//         The tangent rows carry the forward-mode derivative of the network w.r.t. the input point through the same GEMMs
//         (no bias; activation a' = sigma'(z) z'), so that |grad f| of the eikonal term costs one pass over 4 n rows.
//   FIX : row rows[r] of x, r < R, nothing saved: the second pass of the two-pass forward (k_h1_fwd first): out[rows[r]] is
//         overwritten with the three-product value, the sign bit of the row is corrected, max |new - old| is recorded
//   RR  : the eikonal term by REVERSE over reverse (round 4; what the reference's autograd does, gshell_tets_geometry.py:302-324), half the
//         row passes of EIK.  With a_l = sigma(z_l), s_l = sigma'(z_l), in_l = the layer's input ([a_{l-1} | e] at the skip layer):
//           1. ROWS forward over the samples (saves a_l, e)             2. ROWS reverse chain with g_out = 1: delta_l = s_l (W_{l+1}^T delta_{l+1}),
//              grad f = J_enc^T (adjoint of e)  ->  loss = sum (|grad f| - 1)^2, gbar = d loss / d grad f
//           3. k_h2_fwd<RR>: the adjoint of pass 2 is a TANGENT pass in direction gbar:  dt_l = W_l uin_l  (uin_0 = J_enc gbar, no bias),
//              u_l = s_l dt_l (saved: the X operand of delta_l in dW_l), and the second-order source  S_l = 100 (1 - s_l) delta_l dt_l
//           4. k_h2_bwd<RR>: the adjoint of pass 1 with that source:  zbar_l = s_l (W_{l+1}^T zbar_{l+1}) + S_l   (top: 0)
//           5. dW_l = sum_rows zbar_l (x) in_l + delta_l (x) uin_l,  db_l = sum zbar_l,  dw_out = sum u_{L-1}: ONE weight-gradient launch over
//              2 Rpad rows -- the planes are laid out [a ; u], [zbar ; delta], [e ; J gbar] with the second operand pair Rpad rows below the first.
enum { MODE_GRID = 0, MODE_ROWS = 1, MODE_EIK = 2, MODE_FIX = 3, MODE_RR = 4 };
// status words of the forward kernels (device, zeroed by the caller): [0] != 0: a non-finite value left the network (an activation
// or weight beyond the fp16 range -- the caller re-runs the exact-fp32 kernel); [1]: bits of max |three-product - one-product| over
// the rows the second pass recomputed (the a-posteriori check of the one-product pass's error bound)
enum { ST_NONFINITE = 0, ST_MAXDEV = 1, ST_MAXREL = 2 };

struct H2Args {
    const float* x;       // [N,3] points
    const int32_t* rows;  // MODE_ROWS: [R]
    uint64_t* occ;        // MODE_GRID: optional sign bits, word t = ballot(sdf[64 t + i] > 0);  MODE_FIX: the same words, corrected per row
    uint32_t* status;     // optional status words (ST_*)
    float* out;           // MODE_GRID / MODE_FIX: [N] sdf;  MODE_EIK: [tiles*64] per virtual row (value rows: f - b_out; tangent rows: df/dx_d)
    float* A;             // saved activations [n_layers][Rpad][256] fp32 (value rows a_l, tangent rows a'_l)   (ROWS / EIK)
    float* EMB;           // saved encoding [Rpad][EK] fp32 (tangent rows: d enc / dx_d)                            (ROWS / EIK)
    int64_t N;            // GRID: rows; ROWS: active rows R (capacity when n_dev is given); EIK: samples
    const int64_t* n_dev; // ROWS: optional DEVICE-resident row count (<= N): tiles past it exit; lets the caller size by a bound, no host sync
    int64_t Rpad;         // tiles * 64
    int n_freq, E;
    int n_layers;         // hidden-producing layers (first + n_hidden)
    int skip_layer;       // index (>= 1) of the layer whose input is [h | emb], or -1
    const h8* wfrag[MAX_LAYERS];    // fragment-major weights: [k-step][wave 8][piece 2][lane 64] x 16 B
    const float* tailR;             // k_h1r_fwd: [n_layers][256] biases x c, [256] output weights / c   (c = 100 log2(e), see r1_act_a)
    const h8* wfragR[MAX_LAYERS];   // k_h1r_fwd: [feature block 8][k-step][lane 64] x 16 B, high pieces, K in the register-resident order
    const float* bias[MAX_LAYERS];  // [256] fp32
    const float* w_out;             // [256] fp32 followed by the output bias
    float tau;                      // MODE_FIX: > 0 = also record status[ST_MAXREL] = max |new - old| / max(tau, |new|)  (status then has 3 words)
    int64_t Pstride;                // rows per layer of the saved planes; 0 = Rpad (RR: 2 Rpad, the two operand pairs share one allocation)
    // MODE_RR (tangent pass): A = u planes WRITTEN, EMB = J_enc gbar WRITTEN; read: the value pass's planes, the unit direction and its scalar
    const float* A_in;              // a_l planes of the value pass
    const float* EMB_in;            // its saved encoding [Rpad][EK]
    const float* D_in;              // delta_l planes (reverse chain with g_out = 1)
    float* S_out;                   // source planes S_l WRITTEN
    const float* gbar;              // [N][3] d loss / d grad f for a unit upstream gradient
    const float* gmul;              // device scalar: the upstream gradient of the loss
};

__device__ __forceinline__ void split_h2(float v, _Float16& hi, _Float16& lo) {
    v = fminf(fmaxf(v, -60000.0f), 60000.0f);
    _Float16 h = (_Float16)v;                              // round to nearest even
    if (GS_H2_FLUSH && fabsf(v) < H_MIN_NORMAL) h = (_Float16)0.0f;
    hi = h;
    lo = (_Float16)((v - (float)h) * LO_SCALE);
#ifdef GS_H2_EMU1          // numerics experiment: drop the low pieces = the arithmetic of a ONE-product fp16 kernel
    lo = (_Float16)0.0f;
#endif
}

// two values at a time: v_pk_* fp32 ops and v_cvt_pk_f16_f32 (the epilogue is VALU bound: see DESIGN.md)
__device__ __forceinline__ void split_h2_pair(f2 v, h2& hi, h2& lo) {
    f2 vh = v;
    if (GS_H2_FLUSH) {
        vh.x = fabsf(v.x) < H_MIN_NORMAL ? 0.0f : v.x;
        vh.y = fabsf(v.y) < H_MIN_NORMAL ? 0.0f : v.y;
    }
    hi = __builtin_convertvector(vh, h2);
    const f2 hf = __builtin_convertvector(hi, f2);
    lo = __builtin_convertvector((v - hf) * LO_SCALE, h2);
#ifdef GS_H2_EMU1
    lo = h2{(_Float16)0.0f, (_Float16)0.0f};
#endif
}

// Softplus(beta = 100, threshold 20) on the raw v_exp_f32 / v_log_f32 (base 2, constants folded):
// absolute error <= 1e-9 on the value (log2(1 + e) flushes e < 6e-8, i.e. softplus < 6e-10, to 0)
constexpr float SP_C1 = 144.26950408889634f;      // 100 log2(e)
constexpr float SP_C2 = 0.006931471805599453f;    // ln(2) / 100
constexpr float SP_T = 28.853900817779268f;       // 20 log2(e)
__device__ __forceinline__ f2 softplus100_pair(f2 z) {
    const f2 t = z * SP_C1;
    f2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    e = e + 1.0f;
    f2 l = {__builtin_amdgcn_logf(e.x), __builtin_amdgcn_logf(e.y)};
    l = l * SP_C2;
    return f2{t.x > SP_T ? z.x : l.x, t.y > SP_T ? z.y : l.y};
}
// sigma'(z) of that softplus = logistic(100 z)
__device__ __forceinline__ f2 logistic100_pair(f2 z) {
    const f2 t = z * (-SP_C1);
    f2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    e = e + 1.0f;
    return f2{__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
}
// the same derivative from the SAVED softplus value a = softplus(z) >= 0:  sigma' = 1 - exp(-100 a)   (exact identity)
__device__ __forceinline__ float slope_from_value(float a) {
    // both forms evaluated, one selected: as a branch this split every epilogue that uses it into ~30 basic blocks (one per element), each
    // waiting for its own plane load
    const float x = 100.0f * a;
    const float small = x * (1.0f - 0.5f * x + 0.16666667f * x * x), big = 1.0f - __builtin_amdgcn_exp2f(-SP_C1 * a);
    return x < 0.01f ? small : big;
}

#ifndef GS_H2_BPF
#define GS_H2_BPF 0     // 1: double-buffer the LDS (activation) fragments as well
#endif
#ifndef GS_H2_ORDER
#define GS_H2_ORDER 0   // 1: pin the MFMA order so that MFMAs on one accumulator are three issue slots apart
#endif
#ifndef GS_H2_PRIO
#define GS_H2_PRIO 0    // 1: raise the wave priority around the MFMA cluster
#endif
#ifndef GS_H2_PD
#define GS_H2_PD 1      // weight-fragment prefetch distance in k-steps (register ring of PD + 1 stages x 8 VGPRs); measured flat 1..4
#endif

// hi/lo += W[32 output features of block `blk`][K] . P[64 rows][K]^T over NSTEPS k-steps of 16 (compile-time: fully unrolled,
// the register ring rotates for free).  `wf` is fragment-major with `nblk` 32-feature blocks per k-step.
template <int STRIDE, int NSTEPS>
__device__ __forceinline__ void gemm_seg(v16f (&hi)[2], v16f (&lo)[2], const _Float16* __restrict__ P1, const _Float16* __restrict__ P2,
                                         const h8* __restrict__ wf, int blk, int nblk, int lane) {
    constexpr int PD = GS_H2_PD < NSTEPS ? GS_H2_PD : NSTEPS - 1;
    const int row = lane & 31, kq = lane >> 5;
    const _Float16* b1p = P1 + row * STRIDE + kq * 8;
    const _Float16* b2p = P2 + row * STRIDE + kq * 8;
    const h8* wp = wf + blk * 128 + lane;       // + step * nblk * 128 (+64 for the low piece)
    const int sstride = nblk * 128;
    h8 a1[PD + 1], a2[PD + 1];
#pragma unroll
    for (int i = 0; i < PD; ++i) {
        a1[i] = wp[i * sstride];
        a2[i] = wp[i * sstride + 64];
    }
#if GS_H2_BPF
    h8 c10 = *reinterpret_cast<const h8*>(b1p), c11 = *reinterpret_cast<const h8*>(b1p + 32 * STRIDE);
    h8 c20 = *reinterpret_cast<const h8*>(b2p), c21 = *reinterpret_cast<const h8*>(b2p + 32 * STRIDE);
#endif
#pragma unroll
    for (int st = 0; st < NSTEPS; ++st) {
#ifndef GS_H2_NOW          // timing experiment: GS_H2_NOW keeps re-using the first weight fragments (wrong results)
        if (st + PD < NSTEPS) {
            a1[(st + PD) % (PD + 1)] = wp[(st + PD) * sstride];
            a2[(st + PD) % (PD + 1)] = wp[(st + PD) * sstride + 64];
        }
#endif
#if GS_H2_BPF
        const h8 b10 = c10, b11 = c11, b20 = c20, b21 = c21;
        if (st + 1 < NSTEPS) {
            c10 = *reinterpret_cast<const h8*>(b1p + (st + 1) * 16);
            c11 = *reinterpret_cast<const h8*>(b1p + 32 * STRIDE + (st + 1) * 16);
            c20 = *reinterpret_cast<const h8*>(b2p + (st + 1) * 16);
            c21 = *reinterpret_cast<const h8*>(b2p + 32 * STRIDE + (st + 1) * 16);
        }
#else
#ifdef GS_H2_NOLDS        // timing experiment: re-use the first activation fragments (wrong results)
        const int so = 0;
#else
        const int so = st * 16;
#endif
        const h8 b10 = *reinterpret_cast<const h8*>(b1p + so);
        const h8 b11 = *reinterpret_cast<const h8*>(b1p + 32 * STRIDE + so);
        const h8 b20 = *reinterpret_cast<const h8*>(b2p + so);
        const h8 b21 = *reinterpret_cast<const h8*>(b2p + 32 * STRIDE + so);
#endif
#ifdef GS_H2_NOW
        const h8 w1 = a1[0], w2 = a2[0];
#else
        const h8 w1 = a1[st % (PD + 1)], w2 = a2[st % (PD + 1)];
#endif
        __builtin_amdgcn_sched_barrier(0);
#if GS_H2_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#if GS_H2_ORDER
        // same-accumulator MFMAs three issue slots apart in every step and across steps (lo0 . lo1 . hi0 . lo0 . lo1 . hi1)
        lo[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b20, lo[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        lo[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b21, lo[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        hi[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b10, hi[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        lo[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, b10, lo[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        lo[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, b11, lo[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        hi[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b11, hi[1], 0, 0, 0);
#else
        hi[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b10, hi[0], 0, 0, 0);
        hi[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b11, hi[1], 0, 0, 0);
        lo[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b20, lo[0], 0, 0, 0);
        lo[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b21, lo[1], 0, 0, 0);
        lo[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, b10, lo[0], 0, 0, 0);
        lo[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, b11, lo[1], 0, 0, 0);
#endif
#if GS_H2_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void zero_acc(v16f (&hi)[2], v16f (&lo)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) hi[s][r] = lo[s][r] = 0.f;
}

__device__ __forceinline__ float pow2_scale_for(float gmax) {      // 2^-floor(log2 gmax), 1 for zero / tiny rows
    const int e = (__float_as_int(gmax) >> 23) & 0xff;
    return (e < 32 || e > 222) ? 1.0f : __int_as_float((254 - e) << 23);
}

// Point of tile row `row` (and, for EIK, which virtual row it is: c = 0 value, 1..3 tangent d = c - 1).
template <int MODE>
__device__ __forceinline__ bool tile_point(const H2Args& A, int64_t n_act, int64_t tile, int row, float (&p)[3], int& c) {
    int64_t src;
    c = 0;
    if (MODE == MODE_EIK) {
        src = tile * 16 + (row & 15);
        c = row >> 4;
        if (src >= n_act) return false;
    } else {
        src = tile * TM + row;
        if (src >= n_act) return false;
        if ((MODE == MODE_ROWS || MODE == MODE_FIX) && A.rows) src = A.rows[src];       // rows == nullptr: the rows are x[0 .. N)
    }
    p[0] = A.x[3 * src]; p[1] = A.x[3 * src + 1]; p[2] = A.x[3 * src + 2];
    return true;
}

static_assert(EK == 48, "the encoding stage lays out 39 + 9 columns");

// DUAL: one 1024-thread workgroup carries TWO 64-row tiles (waves 0-7 and 8-15, all 160 KB of LDS) through the layers ONE
// BARRIER STEP APART: while one half runs the k-loop of layer l (matrix pipe) the other runs the epilogue of its layer l - 1 /
// l (VALU), then they swap -- the two halves share every s_barrier, so the anti-phase is deterministic.  Two independent
// 512-thread workgroups per CU (DUAL = false) run the same code but settle IN phase (both in the k-loop, then both in the
// epilogue): matrix pipe 49 % busy; starting the second one late changes nothing (measured, GS_H2_STAGGER experiment).
template <int MODE, bool DUAL>
__global__ void __launch_bounds__(DUAL ? 2 * NT : NT, 4) k_h2_fwd(H2Args A) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    const int half = DUAL ? (int)(threadIdx.x >> 9) : 0;
    _Float16* H1 = smem_h + half * (2 * TM * LDH + 2 * TM * LDEH);   // [TM][LDH]
    _Float16* H2 = H1 + TM * LDH;            // [TM][LDH]
    _Float16* E1 = H2 + TM * LDH;            // [TM][LDEH]
    _Float16* E2 = E1 + TM * LDEH;           // [TM][LDEH]   (the output reduction scratch is overlaid on E1/E2 at the end)
    const int tid = threadIdx.x & (NT - 1), lane = tid & 63, wave = tid >> 6;
    const int64_t tile = DUAL ? 2 * (int64_t)blockIdx.x + half : (int64_t)blockIdx.x, r0 = tile * TM;
    const int64_t n_act = ((MODE == MODE_ROWS || MODE == MODE_FIX) && A.n_dev) ? min(*A.n_dev, A.N) : A.N;
    if ((MODE == MODE_ROWS || MODE == MODE_FIX) && (DUAL ? 2 * (int64_t)blockIdx.x : tile) * TM >= n_act) return;      // whole workgroup past the device-side count
    const int64_t PS = A.Pstride ? A.Pstride : A.Rpad;

    // encoding of the tile, zero padded to EK columns, zero rows past the end.  24 work items per row: 18 (frequency, axis)
    // pairs -- ONE sincosf serves the sin and the cos column (and, on tangent rows, both derivatives) --, the 3 coordinates,
    // and 3 x 3 padding columns
    auto put = [&](int row, int f, float v) {
        _Float16 hi, lo;
        split_h2(v, hi, lo);
        E1[row * LDEH + f] = hi;
        E2[row * LDEH + f] = lo;
        if (MODE == MODE_ROWS || MODE == MODE_EIK) A.EMB[(r0 + row) * EK + f] = v;
    };
    if (MODE == MODE_RR) {
        // tangent pass: the input is J_enc(x) gbar, from the SAVED sin / cos of the value pass (no sincosf), every row scaled by a power of
        // two so that the chain runs at O(1) whatever the upstream scalar is (it is linear in gbar); the planes are written unscaled
        const float gm = *A.gmul;
        for (int idx = tid; idx < TM * 24; idx += NT) {
            const int row = idx / 24, slot = idx - row * 24;
            const int64_t src = r0 + row;
            const bool valid = src < n_act;
            float gb[3] = {0.f, 0.f, 0.f};
            if (valid) { gb[0] = A.gbar[3 * src] * gm; gb[1] = A.gbar[3 * src + 1] * gm; gb[2] = A.gbar[3 * src + 2] * gm; }
            const float rsc = pow2_scale_for(fmaxf(fmaxf(fabsf(gb[0]), fabsf(gb[1])), fabsf(gb[2])));
            const float* em = A.EMB_in + src * EK;
            auto put_t = [&](int f, float v) {
                _Float16 hi, lo;
                split_h2(v * rsc, hi, lo);
                E1[row * LDEH + f] = hi;
                E2[row * LDEH + f] = lo;
                A.EMB[src * EK + f] = v;
            };
            if (slot < 18) {
                const int k = slot / 3, ax = slot - 3 * k;
                float vs = 0.f, vc = 0.f;
                if (valid && k < A.n_freq) {
                    const float fr = (float)(1 << k);
                    vs = fr * em[3 + 6 * k + 3 + ax] * gb[ax];          // d sin(2^k x) = 2^k cos(2^k x) dx
                    vc = -fr * em[3 + 6 * k + ax] * gb[ax];
                }
                put_t(3 + 6 * k + ax, vs);
                put_t(3 + 6 * k + 3 + ax, vc);
            } else if (slot < 21) {
                put_t(slot - 18, gb[slot - 18]);
            } else {
                for (int j = 0; j < 3; ++j) put_t(39 + 3 * (slot - 21) + j, 0.0f);
            }
        }
    }
    for (int idx = tid; MODE != MODE_RR && idx < TM * 24; idx += NT) {
        const int row = idx / 24, slot = idx - row * 24;
        float p[3] = {0.f, 0.f, 0.f};
        int c = 0;
        const bool valid = tile_point<MODE>(A, n_act, tile, row, p, c);
        if (slot < 18) {
            const int k = slot / 3, ax = slot - 3 * k;
            float vs = 0.f, vc = 0.f;
            if (valid && k < A.n_freq) {
                const float fr = (float)(1 << k);
                float sn, cs;
                sincosf(fr * p[ax], &sn, &cs);
                if (c == 0) { vs = sn; vc = cs; }
                else if (ax == c - 1) { vs = fr * cs; vc = -fr * sn; }      // d/dx_ax of (sin, cos)(2^k x_ax)
            }
            put(row, 3 + 6 * k + ax, vs);
            put(row, 3 + 6 * k + 3 + ax, vc);
        } else if (slot < 21) {
            const int f = slot - 18;
            put(row, f, valid ? (c == 0 ? p[f] : (f == c - 1 ? 1.0f : 0.0f)) : 0.0f);
        } else {
            for (int j = 0; j < 3; ++j) put(row, 39 + 3 * (slot - 21) + j, 0.0f);
        }
    }
    __syncthreads();
    if (DUAL && half == 1) __syncthreads();              // the second tile runs one barrier step behind the first

    const int n_base = wave * 32 + 4 * (lane >> 5);      // + 8 g + j  (g = reg >> 2, j = reg & 3)
    const int m_lane = lane & 31;                        // + 32 s
    const bool low16 = (lane & 16) == 0;                 // EIK: this lane holds (value, d/dy) rows; the other half (d/dx, d/dz)
    float isc[2] = {1.0f, 1.0f};                         // RR: 1 / row scale of rows m_lane, 32 + m_lane (exact: powers of two)
    if (MODE == MODE_RR) {
        const float gm = *A.gmul;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int64_t r = r0 + 32 * s + m_lane;
            if (r < n_act) isc[s] = 1.0f / pow2_scale_for(fmaxf(fmaxf(fabsf(A.gbar[3 * r] * gm), fabsf(A.gbar[3 * r + 1] * gm)), fabsf(A.gbar[3 * r + 2] * gm)));
        }
    }
    for (int l = 0; l < A.n_layers; ++l) {
        v16f hi[2], lo[2];
        zero_acc(hi, lo);
        if (l == 0) {
            gemm_seg<LDEH, EK / 16>(hi, lo, E1, E2, A.wfrag[0], wave, 8, lane);
        } else {
            gemm_seg<LDH, D / 16>(hi, lo, H1, H2, A.wfrag[l], wave, 8, lane);
            if (l == A.skip_layer) gemm_seg<LDEH, EK / 16>(hi, lo, E1, E2, A.wfrag[l] + (D / 16) * 1024, wave, 8, lane);
        }
        // bias (and output weights) of this lane's 16 features: requested BEFORE the barrier, so that the L2 round trip is
        // covered by the wait for the slowest wave instead of stalling the epilogue four times per layer (measured: 5.8 -> ? ms)
        const float* bl = A.bias[l] + n_base;
        const bool last = l + 1 == A.n_layers;
        float4 b4v[4], w4v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            b4v[g] = MODE == MODE_RR ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(bl + 8 * g);
            w4v[g] = (last && MODE != MODE_RR) ? *reinterpret_cast<const float4*>(A.w_out + n_base + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // RR: the saved a_l and delta_l of this lane's elements, one group of four features AHEAD of its use (the first group before the
        // barrier): issued in front of the previous group's plane stores, which the compiler may not move them across (same allocations)
        float avn[2][4], dvn[2][4];
        auto load_group = [&](int g) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int64_t pi = plane_idx(PS, l, r0 + 32 * s + m_lane, n_base + 8 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) { avn[s][j] = A.A_in[pi + 32 * j]; dvn[s][j] = A.D_in[pi + 32 * j]; }
            }
        };
        if (MODE == MODE_RR) load_group(0);
#ifndef GS_H2_NOBAR      // timing experiment: no barriers inside the layer loop (wrong results)
        __syncthreads();     // every wave is done reading the planes: they are overwritten in place
#endif
        f2 part[2] = {f2{0.f, 0.f}, f2{0.f, 0.f}};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b4 = b4v[g];
            const f2 bj[2] = {f2{b4.x, b4.y}, f2{b4.z, b4.w}};
            const f2 wj[2] = {f2{w4v[g].x, w4v[g].y}, f2{w4v[g].z, w4v[g].w}};
            f2 v[2][2];          // [row half s][pair]
            f2 src_t[2][2];      // RR: the second-order source S_l of the same elements
            float av[2][4], dv[2][4];
            if (MODE == MODE_RR) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { av[s][j] = avn[s][j]; dv[s][j] = dvn[s][j]; }
                if (g + 1 < 4) load_group(g + 1);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = 4 * g + 2 * q;
                const f2 z0 = f2{hi[0][r], hi[0][r + 1]} + f2{lo[0][r], lo[0][r + 1]} * LO_INV;
                const f2 z1 = f2{hi[1][r], hi[1][r + 1]} + f2{lo[1][r], lo[1][r + 1]} * LO_INV;
                if (MODE == MODE_RR) {
                    // u = s dt (the next layer's input, still scaled);  S = 100 (1 - s) delta dt, unscaled
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const f2 z = s == 0 ? z0 : z1;
                        const f2 sl = f2{slope_from_value(av[s][2 * q]), slope_from_value(av[s][2 * q + 1])};
                        v[s][q] = sl * z;
                        src_t[s][q] = (1.0f - sl) * f2{dv[s][2 * q], dv[s][2 * q + 1]} * z * (100.0f * isc[s]);
                    }
                    continue;
                }
#ifdef GS_H2_NOEPI       // timing experiment only: no softplus (wrong results)
                if (true) {
                    v[0][q] = z0 + bj[q];
                    v[1][q] = z1 + bj[q];
                } else
#endif
                if (MODE != MODE_EIK) {
                    v[0][q] = softplus100_pair(z0 + bj[q]);
                    v[1][q] = softplus100_pair(z1 + bj[q]);
                } else {
                    // lanes with (lane & 16) == 0: acc 0 = value row, acc 1 = d/dy row of the SAME sample; the lanes 16 above
                    // hold that sample's d/dx (acc 0) and d/dz (acc 1) rows: a' = sigma'(z) z' with sigma' from the value row
                    const f2 zb = z0 + bj[q];
                    f2 sl = logistic100_pair(zb);
                    f2 so = f2{__shfl_xor(sl.x, 16, 64), __shfl_xor(sl.y, 16, 64)};
                    const f2 s = low16 ? sl : so;
                    v[0][q] = low16 ? softplus100_pair(zb) : s * z0;
                    v[1][q] = s * z1;
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (MODE == MODE_ROWS || MODE == MODE_EIK) {
                    float* ap = A.A + plane_idx(PS, l, r0 + 32 * s + m_lane, n_base + 8 * g);
                    ap[0] = v[s][0].x; ap[32] = v[s][0].y; ap[64] = v[s][1].x; ap[96] = v[s][1].y;
                }
                if (MODE == MODE_RR) {
                    const int64_t pi = plane_idx(PS, l, r0 + 32 * s + m_lane, n_base + 8 * g);
                    float* ap = A.A + pi;
                    float* sp = A.S_out + pi;
                    ap[0] = v[s][0].x * isc[s]; ap[32] = v[s][0].y * isc[s]; ap[64] = v[s][1].x * isc[s]; ap[96] = v[s][1].y * isc[s];
                    sp[0] = src_t[s][0].x; sp[32] = src_t[s][0].y; sp[64] = src_t[s][1].x; sp[96] = src_t[s][1].y;
                }
                if (last) {
                    if (MODE != MODE_RR) part[s] = part[s] + v[s][0] * wj[0] + v[s][1] * wj[1];
                } else {
                    h2 a0, b0, a1, b1;
                    split_h2_pair(v[s][0], a0, b0);
                    split_h2_pair(v[s][1], a1, b1);
                    const int off = (32 * s + m_lane) * LDH + n_base + 8 * g;
                    *reinterpret_cast<h4*>(H1 + off) = h4{a0.x, a0.y, a1.x, a1.y};
                    *reinterpret_cast<h4*>(H2 + off) = h4{b0.x, b0.y, b1.x, b1.y};
                }
            }
        }
        if (last && MODE != MODE_ROWS && MODE != MODE_RR) {
            // output layer: this lane holds the sum over its 16 features; add the other 16 of the wave's 32 (lane ^ 32), then
            // the 8 waves through LDS in a fixed order (deterministic)
            float* red = reinterpret_cast<float*>(E1);       // [8 waves][64 rows] fp32 = 2 KB (the encoding planes are dead now)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float p = part[s].x + part[s].y;
                p += __shfl_xor(p, 32, 64);
                if (lane < 32) red[wave * TM + 32 * s + lane] = p;
            }
        }
#ifndef GS_H2_NOBAR
        __syncthreads();
#else
        if (last) __syncthreads();
#endif
    }
    if (DUAL && half == 0) __syncthreads();              // same number of barriers in both halves
    if (MODE != MODE_ROWS && MODE != MODE_RR && tid < TM) {
        const float* red = reinterpret_cast<const float*>(E1);
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += red[w * TM + tid];
        int64_t r = r0 + tid;
        if (MODE == MODE_GRID) {
            s += A.w_out[D];
            if (r < A.N) A.out[r] = s;
            // fused geometry front end: the tile is one 64-bit word of the extraction's occupancy bits (strict > 0, ref gshell_tets.py:250)
            const uint64_t m = __ballot(r < A.N && s > 0.0f);
            if (A.occ && tid == 0 && r0 < A.N) A.occ[tile] = m;
            // an activation beyond the fp16 range turns into inf in the pair split and reaches the output as inf / NaN
            if (A.status && __ballot(r < A.N && !(fabsf(s) < 3.0e38f)) != 0ull && tid == 0) atomicOr(&A.status[ST_NONFINITE], 1u);
        } else if (MODE == MODE_FIX) {
            s += A.w_out[D];
            float dev = 0.f, rel = 0.f;
            if (r < n_act) {
                const int64_t g = A.rows[r];
                const float old = A.out[g];
                dev = fabsf(s - old);
                // the one-product value decides a SIGN correctly iff its error is below |value|; below tau the row is re-evaluated anyway,
                // so the margin that matters is max(tau, |value|): rows far from the surface (the audit sample) may be off by more than tau
                rel = dev / fmaxf(A.tau, fabsf(s));
                A.out[g] = s;
                if (A.occ && ((s > 0.0f) != (old > 0.0f))) atomicXor((unsigned long long*)&A.occ[g >> 6], 1ull << (g & 63));
                if (!(fabsf(s) < 3.0e38f)) { dev = 0.f; rel = 0.f; if (A.status) atomicOr(&A.status[ST_NONFINITE], 1u); }
            }
            for (int o = 32; o > 0; o >>= 1) { dev = fmaxf(dev, __shfl_xor(dev, o, 64)); rel = fmaxf(rel, __shfl_xor(rel, o, 64)); }
            if (A.status && tid == 0 && dev > 0.f) {
                atomicMax(&A.status[ST_MAXDEV], __float_as_uint(dev));
                if (A.tau > 0.f) atomicMax(&A.status[ST_MAXREL], __float_as_uint(rel));
            }
        } else {
            A.out[r] = s;                                    // virtual rows: value rows lack b_out (unused), tangent rows = df/dx_d
        }
    }
}

// ---- one-product forward: the FIRST pass of the two-pass full-grid evaluation --------------------------------------------------
// Far from the surface the signed distance is consumed only through its sign (extraction: occ = sdf > 0, ref gshell_tets.py:250;
// sdf regulariser and every gradient: end points of sign-crossing edges only, gshell_tets_geometry.py:33-39).  This kernel
// evaluates the network with ONE fp16 product per algorithmic product (operands rounded to fp16, 2^-11; fp32 accumulate): a third
// of the matrix work of k_h2_fwd, one activation plane (41 KB of LDS: three workgroups per CU), no pair split in the epilogue.
// Its values carry an error of ~1e-4 (measured: max 2.3e-4, rms 3e-5 on the fitted bench network); every row that can matter --
// |sdf| below a threshold tau >> that error, or an end point of an edge with such a row or with a sign change (gs_mtets_flag_refine_rows)
// -- is then recomputed by k_h2_fwd<MODE_FIX>, which also MEASURES the error on those rows (ST_MAXDEV).  If the one-product error
// is below tau everywhere, signs at all vertices and values at all crossing-edge end points equal the one-pass h2 result bit for bit
// (argument in DESIGN.md 2.1; tests/test_fullsize_parity_gpu.py checks it on the res-256 grid).
#ifndef GS_H1_PD
#define GS_H1_PD 2
#endif
#ifndef GS_H1_ASM
#define GS_H1_ASM 0       // 1: activation fragments requested two k-steps ahead with volatile-asm ds_reads and counted waits.  Bit-identical results;
                           // measured 5.85 ms against 2.66 with 8 waves per workgroup, 2.67 (no change) with 4: more LDS requests in flight per wave HURT
#endif
#ifndef GS_H1_PRE
#define GS_H1_PRE 0       // 1: the next layer's first weight fragments requested before the epilogue -- measured SLOWER (2.74 vs 2.59 ms: the 8 live VGPRs cost a wave of occupancy)
#endif
// Operand traffic decides this kernel, not the epilogue (profiles/r03_h1_dissection.txt): a 32x32x16 MFMA takes 8 cycles of a CU's
// four matrix pipes, in which LDS delivers 1 KB and the vector-memory path 0.5 KB.  A wave that owns NB feature blocks x RM row blocks
// reads RM activation fragments (LDS) and NB weight fragments (L1 / L2) of 1 KB each per k-step for NB x RM MFMAs:
//   NB 1, RM 2 (8 waves x 64 rows)   1 KB LDS + 0.5 KB L1 per MFMA: both pipes at their limit, the matrix pipe reaches ~50 %
//   NB 2, RM 4 (4 waves x 128 rows)  0.5 KB LDS + 0.25 KB L1 per MFMA
template <int NB>
struct H1Pre { h8 w[NB][GS_H1_PD]; };        // the first GS_H1_PD weight fragments of a layer, requested one epilogue ahead (GS_H1_PRE)
template <int NB>
__device__ __forceinline__ void h1_preload(H1Pre<NB>& pre, const h8* __restrict__ wf, int blk0, int nblk, int lane) {
    const h8* wp = wf + blk0 * 128 + lane;
#pragma unroll
    for (int i = 0; i < GS_H1_PD; ++i)
#pragma unroll
        for (int q = 0; q < NB; ++q) pre.w[q][i] = wp[i * nblk * 128 + q * 128];
}
template <int STRIDE, int NSTEPS, int NB, int RM>
__device__ __forceinline__ void gemm_seg1(v16f (&acc)[NB][RM], const _Float16* __restrict__ P1, const h8* __restrict__ wf, int blk0, int nblk, int lane,
                                          const H1Pre<NB>* pre = nullptr) {
    constexpr int PD = GS_H1_PD < NSTEPS ? GS_H1_PD : NSTEPS - 1;      // weight fragments in flight ahead of the MFMAs
    const int row = lane & 31, kq = lane >> 5;
    const _Float16* b1p = P1 + row * STRIDE + kq * 8;
    const h8* wp = wf + blk0 * 128 + lane;       // the high pieces of the h2 fragment set: + step * nblk * 128
    const int sstride = nblk * 128;
    h8 a1[NB][PD + 1];
#pragma unroll
    for (int i = 0; i < PD; ++i)
#pragma unroll
        for (int q = 0; q < NB; ++q) a1[q][i] = pre ? pre->w[q][i] : wp[i * sstride + q * 128];
#if GS_H1_ASM
    // The activation fragments of step st + 2 are requested BEFORE the MFMAs of step st, and stay there: the ds_reads and the counted
    // s_waitcnt are volatile asm (hipcc sinks plain loads back to their uses and then waits out the LDS latency in front of every MFMA
    // pair).  The wait statement names the fragments it releases as in / out operands, so no MFMA can be scheduled above it.
    static_assert(RM == 2 || RM == 4, "counted waits are written out for 2 and 4 row blocks");
    constexpr int LA = 2 < NSTEPS ? 2 : NSTEPS - 1;
    const uint32_t la = (uint32_t)(uintptr_t)b1p;
    h8 bq[LA + 1][RM];
    auto request = [&](int st) {
#pragma unroll
        for (int r = 0; r < RM; ++r) asm volatile("ds_read_b128 %0, %1" : "=v"(bq[st % (LA + 1)][r]) : "v"(la + 2u * (uint32_t)(32 * r * STRIDE + st * 16)) : "memory");
    };
#pragma unroll
    for (int i = 0; i < LA; ++i) request(i);
#pragma unroll
    for (int st = 0; st < NSTEPS; ++st) {
        if (st + PD < NSTEPS)
#pragma unroll
            for (int q = 0; q < NB; ++q) a1[q][(st + PD) % (PD + 1)] = wp[(st + PD) * sstride + q * 128];
        if (st + LA < NSTEPS) request(st + LA);
        const int ahead = (NSTEPS - 1 - st) < LA ? (NSTEPS - 1 - st) : LA;          // steps requested beyond this one
        h8(&cur)[RM] = bq[st % (LA + 1)];
        if (RM == 2) {
            if (ahead == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]));
            else if (ahead == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(cur[0]), "+v"(cur[1]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]));
        } else {
            if (ahead == 2) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2 % RM]), "+v"(cur[3 % RM]));
            else if (ahead == 1) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2 % RM]), "+v"(cur[3 % RM]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2 % RM]), "+v"(cur[3 % RM]));
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const h8 w1 = a1[q][st % (PD + 1)];
#pragma unroll
            for (int r = 0; r < RM; ++r) acc[q][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, cur[r], acc[q][r], 0, 0, 0);
        }
    }
    return;
#endif
#pragma unroll
    for (int st = 0; st < NSTEPS; ++st) {
        if (st + PD < NSTEPS)
#pragma unroll
            for (int q = 0; q < NB; ++q) a1[q][(st + PD) % (PD + 1)] = wp[(st + PD) * sstride + q * 128];
        h8 bf[RM];
#pragma unroll
        for (int r = 0; r < RM; ++r) bf[r] = *reinterpret_cast<const h8*>(b1p + 32 * r * STRIDE + st * 16);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const h8 w1 = a1[q][st % (PD + 1)];
#pragma unroll
            for (int r = 0; r < RM; ++r) acc[q][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, bf[r], acc[q][r], 0, 0, 0);
        }
    }
}

// Softplus(beta = 100) of the one-product pass WITHOUT transcendentals.  v_exp_f32 / v_log_f32 issue at a quarter of the VALU rate:
// the exact softplus is 4 of them + 9 full-rate instructions per PAIR of elements = 100 cycles, 1600 per layer and wave against
// 1024 cycles of MFMA work -- the epilogue, not the matrix pipe, bounded the kernel.  With t = 100 log2(e) z:
//     softplus(z) = max(z, 0) + (ln 2 / 100) g(|t|),   g(u) = log2(1 + 2^-u)  in (0, 1],
// g is replaced by a degree-8 polynomial in v = u / 5 - 1 on u in [0, 10] (max error 6.2e-5, i.e. 4e-7 on the activation -- the
// fp16 rounding of the activation that follows is 2^-11 relative) and held at its end value beyond (g(10) = 1.4e-3: 1e-5 on the
// activation; the exact kernel switches to the identity at t > 28.9).  All of it packed fp32 FMAs: 17 full-rate instructions per pair.
#ifndef GS_H1_ASMMAX
#define GS_H1_ASMMAX 1
#endif
#ifndef GS_H1_POLY
#define GS_H1_POLY 0       // 0: the exact softplus100_pair (default: measured FASTER -- 2.45 ms against 2.74 (degree 8) / 2.68 (degree 7): the
                           // transcendental unit runs beside the VALU, the 17 packed instructions do not)
#endif
// this file is compiled with -ffp-contract=off: the fused form has to be asked for
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, float c) { return __builtin_elementwise_fma(a, b, f2{c, c}); }
__device__ __forceinline__ f2 softplus100_pair_poly(f2 z) {
#if GS_H1_POLY == 0
    // max(z, 0) + (ln 2 / 100) log2(1 + 2^-|t|): no overflow for any t, so no threshold select (2 v_cmp + 2 v_cndmask per pair less than
    // softplus100_pair; differs from torch's thresholded form by <= log(1 + e^-20) / 100 = 2e-11)
    const f2 t = z * SP_C1;
    f2 e = {__builtin_amdgcn_exp2f(-fabsf(t.x)), __builtin_amdgcn_exp2f(-fabsf(t.y))};
    e = e + 1.0f;
    const f2 l = {__builtin_amdgcn_logf(e.x), __builtin_amdgcn_logf(e.y)};
#if GS_H1_ASMMAX
    // max(z, 0) as ONE v_max_f32: fmaxf, and fmed3(z, 0, inf) which the compiler folds back into it, both come with a canonicalising v_max(z, z) in
    // front -- 4 of the epilogue's 12 instructions per pair.  (The unused second input keeps the asm behind the logarithm it is added to: hoisted, a
    // tile's worth of results would sit in registers.)
    f2 m;
    asm("v_max_f32 %0, 0, %1" : "=v"(m.x) : "v"(z.x), "v"(l.x));
    asm("v_max_f32 %0, 0, %1" : "=v"(m.y) : "v"(z.y), "v"(l.y));
    return pk_fma(l, f2{SP_C2, SP_C2}, m);
#else
    return pk_fma(l, f2{SP_C2, SP_C2}, f2{__builtin_amdgcn_fmed3f(z.x, 0.0f, __builtin_inff()), __builtin_amdgcn_fmed3f(z.y, 0.0f, __builtin_inff())});
#endif
#else
    const f2 t = z * SP_C1;
    const f2 u = f2{fminf(fabsf(t.x), 10.0f), fminf(fabsf(t.y), 10.0f)};
    const f2 v = pk_fma(u, f2{0.2f, 0.2f}, -1.0f);
#if GS_H1_POLY == 8
    f2 g = pk_fma(v, f2{-0.0240361113f, -0.0240361113f}, 0.0448770784f);
    g = pk_fma(g, v, 0.0118750501f);
    g = pk_fma(g, v, -0.118264653f);
    g = pk_fma(g, v, 0.214377925f);
    g = pk_fma(g, v, -0.274279803f);
    g = pk_fma(g, v, 0.254028171f);
    g = pk_fma(g, v, -0.151620954f);
    g = pk_fma(g, v, 0.0444051363f);
#else
    f2 g = pk_fma(v, f2{0.0448770784f, 0.0448770784f}, -0.03299281f);
    g = pk_fma(g, v, -0.118264653f);
    g = pk_fma(g, v, 0.240263477f);
    g = pk_fma(g, v, -0.274279803f);
    g = pk_fma(g, v, 0.24932164f);
    g = pk_fma(g, v, -0.151620954f);
    g = pk_fma(g, v, 0.0445358753f);
#endif
    return pk_fma(g, f2{SP_C2, SP_C2}, f2{fmaxf(z.x, 0.0f), fmaxf(z.y, 0.0f)});
#endif
}

#ifndef GS_H1_WAVES
#define GS_H1_WAVES 6     // NW = 8, RM = 2: waves per SIMD the register budget is sized for (3 workgroups of 8 waves per CU)
#endif
#ifndef GS_H1_NW
#define GS_H1_NW 8        // waves per workgroup: 8 (one 32-feature block per wave) or 4 (two blocks per wave)
#endif
#ifndef GS_H1_RM
#define GS_H1_RM 2        // 32-row blocks per workgroup tile: 2 (64 rows) or 4 (128 rows: 80 KB of LDS, two workgroups fill a CU's 160 KB)
#endif
#ifndef GS_H1_WAVES4
#define GS_H1_WAVES4 (GS_H1_RM == 4 ? 2 : 3)    // NW = 4: waves per SIMD
#endif
template <int RM> constexpr size_t smem_h1_bytes() { return (size_t)(32 * RM * LDH + 32 * RM * LDEH) * sizeof(_Float16); }
template <int NW, int RM>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? GS_H1_WAVES : GS_H1_WAVES4) k_h1_fwd(H2Args A) {
    constexpr int NTW = 64 * NW, NB = 8 / NW, TMT = 32 * RM;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    _Float16* H1 = smem_h;                   // [TMT][LDH]
    _Float16* E1 = H1 + TMT * LDH;           // [TMT][LDEH]   (the output reduction scratch is overlaid at the end)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t tile = blockIdx.x, r0 = tile * TMT;
    for (int idx = tid; idx < TMT * 24; idx += NTW) {
        const int row = idx / 24, slot = idx - row * 24;
        const int64_t src = r0 + row;
        const bool valid = src < A.N;
        float p[3] = {0.f, 0.f, 0.f};
        if (valid) { p[0] = A.x[3 * src]; p[1] = A.x[3 * src + 1]; p[2] = A.x[3 * src + 2]; }
        _Float16* e = E1 + row * LDEH;
        if (slot < 18) {
            const int k = slot / 3, ax = slot - 3 * k;
            float sn = 0.f, cs = 0.f;
            if (valid && k < A.n_freq) sincosf((float)(1 << k) * p[ax], &sn, &cs);
            e[3 + 6 * k + ax] = (_Float16)sn;
            e[3 + 6 * k + 3 + ax] = (_Float16)cs;
        } else if (slot < 21) {
            e[slot - 18] = (_Float16)fminf(fmaxf(p[slot - 18], -60000.0f), 60000.0f);
        } else {
            for (int j = 0; j < 3; ++j) e[39 + 3 * (slot - 21) + j] = (_Float16)0.0f;
        }
    }
    const int blk0 = wave * NB;
    const int m_lane = lane & 31;
    // the bias is the accumulators' initial value (no add in the epilogue); the NEXT layer's is requested before this layer's epilogue,
    // so that its L2 round trip (one per layer otherwise: 0.18 ms per launch) hides behind the activation function
    float4 bnext[NB][4];
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) bnext[q][g] = *reinterpret_cast<const float4*>(A.bias[0] + (blk0 + q) * 32 + 4 * (lane >> 5) + 8 * g);
    H1Pre<NB> wpre;
    __syncthreads();
    for (int l = 0; l < A.n_layers; ++l) {
        const bool last = l + 1 == A.n_layers;
        v16f acc[NB][RM];
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = bnext[q][g];
#pragma unroll
                for (int r = 0; r < RM; ++r) {
                    acc[q][r][4 * g] = b4.x; acc[q][r][4 * g + 1] = b4.y; acc[q][r][4 * g + 2] = b4.z; acc[q][r][4 * g + 3] = b4.w;
                }
            }
        {   // next layer's bias, or the output weights after the last hidden layer (consumed by the epilogue of the last layer)
            const float* nb = last ? A.w_out : A.bias[l + 1];
#pragma unroll
            for (int q = 0; q < NB; ++q)
#pragma unroll
                for (int g = 0; g < 4; ++g) bnext[q][g] = *reinterpret_cast<const float4*>(nb + (blk0 + q) * 32 + 4 * (lane >> 5) + 8 * g);
        }
        if (l == 0) {
            gemm_seg1<LDEH, EK / 16, NB, RM>(acc, E1, A.wfrag[0], blk0, 8, lane);
        } else {
            gemm_seg1<LDH, D / 16, NB, RM>(acc, H1, A.wfrag[l], blk0, 8, lane, GS_H1_PRE ? &wpre : nullptr);
            if (l == A.skip_layer) gemm_seg1<LDEH, EK / 16, NB, RM>(acc, E1, A.wfrag[l] + (D / 16) * 1024, blk0, 8, lane);
        }
        // the next layer's first weight fragments: requested here, they land while the activation function runs (each layer's GEMM
        // otherwise opens with an exposed L2 round trip)
        if (GS_H1_PRE && !last) h1_preload<NB>(wpre, A.wfrag[l + 1], blk0, 8, lane);
        __syncthreads();     // every wave is done reading the plane: it is overwritten in place
        f2 part[RM];
#pragma unroll
        for (int r = 0; r < RM; ++r) part[r] = f2{0.f, 0.f};
        // (`last` is tested ONCE per layer, not per element: as a branch inside the loops it cut the epilogue into one basic block per two pairs,
        //  and nothing could be scheduled across them)
        if (last) {
#pragma unroll
            for (int q = 0; q < NB; ++q)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f2 wj[2] = {f2{bnext[q][g].x, bnext[q][g].y}, f2{bnext[q][g].z, bnext[q][g].w}};       // the output weights
#pragma unroll
                    for (int r = 0; r < RM; ++r) {
                        const f2 v0 = softplus100_pair_poly(f2{acc[q][r][4 * g], acc[q][r][4 * g + 1]});
                        const f2 v1 = softplus100_pair_poly(f2{acc[q][r][4 * g + 2], acc[q][r][4 * g + 3]});
                        part[r] = part[r] + v0 * wj[0] + v1 * wj[1];
                    }
                }
        } else {
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int n_base = (blk0 + q) * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < RM; ++r) {
                        const f2 v0 = softplus100_pair_poly(f2{acc[q][r][4 * g], acc[q][r][4 * g + 1]});
                        const f2 v1 = softplus100_pair_poly(f2{acc[q][r][4 * g + 2], acc[q][r][4 * g + 3]});
                        const h2 a0 = __builtin_convertvector(v0, h2), a1 = __builtin_convertvector(v1, h2);
                        *reinterpret_cast<h4*>(H1 + (32 * r + m_lane) * LDH + n_base + 8 * g) = h4{a0.x, a0.y, a1.x, a1.y};
                    }
            }
        }
        if (last) {
            float* red = reinterpret_cast<float*>(E1);       // [NW waves][TMT rows] fp32 (the encoding plane is dead now)
#pragma unroll
            for (int r = 0; r < RM; ++r) {
                float p = part[r].x + part[r].y;
                p += __shfl_xor(p, 32, 64);
                if (lane < 32) red[wave * TMT + 32 * r + lane] = p;
            }
        }
        __syncthreads();
    }
    if (tid < TMT) {
        const float* red = reinterpret_cast<const float*>(E1);
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[w * TMT + tid];
        s += A.w_out[D];
        const int64_t r = r0 + tid;
        if (r < A.N) A.out[r] = s;
        const uint64_t m = __ballot(r < A.N && s > 0.0f);                       // one 64-row sign word per wave
        if (A.occ && lane == 0 && r < A.N) A.occ[tile * (TMT / 64) + wave] = m;
        if (A.status && __ballot(r < A.N && !(fabsf(s) < 3.0e38f)) != 0ull && lane == 0) atomicOr(&A.status[ST_NONFINITE], 1u);
    }
}

// ---- one-product forward, REGISTER-RESIDENT activations (round 5; selectable with gs_sdf_mlp_h1_impl(1), NOT the default) ---------------------------
// k_h1_fwd above is operand-delivery bound: per MFMA it reads 1 KB of activations from LDS and 0.5 KB of weights through L1, and every 64-row tile
// re-reads the 0.65 MB weight image from L2.  This kernel changes the ratio instead of the schedule:
//   * a wave carries a 32-row block through ALL layers and keeps its activations IN REGISTERS.  The transposed product D[feature][row] leaves lane
//     (row n, half h) with features 32 mb + 4 h + 8 g + j (g, j < 4) of ITS row -- which is already a valid B operand of the next layer if k-slot
//     (step s', half h, element i) of that layer is DEFINED as feature F(s', h, i) = 32 (s' / 2) + 4 h + 8 (2 (s' % 2) + i / 4) + i % 4: the K order
//     of a GEMM is free, so the weights are packed in that order (k_h2_pack, third section) and no lane ever exchanges a value.  bias -> accumulator
//     init, softplus in scaled variables on the 16 accumulator values of a feature block, v_cvt_pk_f16_f32 into the next layer's fragments.
//   * weights are the A operand, staged through LDS once per WORKGROUP (8 waves = 256 rows) by LDS-DMA in chunks of one 32-feature block (triple
//     buffered, one barrier per chunk): no activation traffic at all, a quarter of the L2 weight traffic.
// What was measured on MI355X (tools/h1r_check.py, tools/micro/mfma_*.hip; wall clock -- clock64() ticks do NOT advance with wall time when several
// waves share a SIMD):
//   * first form, ONE wave per SIMD with 64 rows (~450 registers): 5.6 ms.  A wave's VALU instructions are not hidden behind its own MFMAs (6 v_fma
//     between two MFMAs: 16.6 -> 25.5 ns per MFMA; a second wave on the SIMD hides about a third of them, four waves half), and hipcc needed 1 500
//     spilled registers for the 400 live ones;
//   * this form, TWO waves per SIMD with 32 rows each (256 registers, ~70 spilled): 2.38 ms against k_h1_fwd's 2.34 -- on par, not better.  Timing-only
//     ablations: without the weight DMA 2.07 (three 8 KB LDS-DMA pieces per block and wave cost their issue slots), without the activation function
//     2.10, without both 1.70; MFMAs alone would take 0.65.  With ~7 VALU instructions per MFMA of this network, MFMA + 0.5 x VALU puts the floor of
//     ANY schedule at ~1.1 ms; neither kernel is near it, and the remaining distance is issue contention, barriers and operand delivery, not a
//     missing trick.  Kept selectable as the measured record of the design; same arithmetic class as k_h1_fwd (different K order: values differ in
//     the last bits of fp32 sums; positional encoding by v_sin_f32 / v_cos_f32).
constexpr int R1_NW = 8;                                       // waves per workgroup = two per SIMD; each carries ONE 32-row block through all layers
constexpr int R1_NT = 64 * R1_NW;
constexpr int R1_STEPS_MAX = D / 16 + EK / 16;                 // k-steps of a chunk at the skip layer
constexpr int R1_STAGE = (R1_STEPS_MAX * 64 + R1_NT - 1) / R1_NT;      // 16-byte pieces per thread and chunk (3)
constexpr int R1_CHUNK = R1_STAGE * R1_NT;                     // h8 per LDS buffer (24 KB): EVERY chunk is staged as 1536 fragments -- a shorter chunk
                                                               // drags the head of the next one (or of the fp32 tail) along, unread: no predicates
constexpr int R1_BUFS = 3;                                     // ring: chunk c is read while c + 1 is already visible (its first fragments are
                                                               // requested BEFORE the barrier that ends c) and c + 2 is landing
constexpr int R1_ENC = R1_NW * (EK / 16) * 64;                 // h8: the waves' encoding fragments [wave][k-step 3][lane 64] (24 KB)
constexpr size_t SMEM_H1R_BYTES = (size_t)(R1_BUFS * R1_CHUNK + R1_ENC) * sizeof(h8) + (size_t)(MAX_LAYERS + 1) * D * sizeof(float);

// global -> LDS without staging registers: global_load_lds_dwordx4, lane i of the wave writes its 16 bytes at M0 + 16 i.  As ONE asm statement: a
// DMA issued through the builtin is a pending LDS write on the compiler's books, so it waits `vmcnt(0)` in front of the next ds_read -- the whole
// L2 round trip, every block.  An asm DMA is invisible to that bookkeeping; the kernel waits for it itself (r1_dma_wait) before the barrier that
// publishes the chunk, one block of MFMAs later.  M0 is saved and restored inside the statement (it is compiler-reserved).
__device__ __forceinline__ void r1_dma_chunk(const h8* __restrict__ src, h8* dst_buf, int tid) {
    static_assert(R1_STAGE == 3 && R1_NT * 16 == 0x2000, "three 8 KB pieces per chunk are written out below");
    const uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) h8*)(dst_buf + (tid & ~63)));
    uint32_t off = (uint32_t)tid * 16u, keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_add_u32 m0, m0, 0x2000\n\t"
        "v_add_u32 %1, 0x2000, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_add_u32 m0, m0, 0x2000\n\t"
        "v_add_u32 %1, 0x2000, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "+v"(off)
        : "s"(src), "s"(lds)
        : "memory", "scc");
}
__device__ __forceinline__ void r1_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ int r1_steps(int l, int skip_layer) { return l == 0 ? EK / 16 : D / 16 + (l == skip_layer ? EK / 16 : 0); }

#ifndef GS_H1R_ABL
#define GS_H1R_ABL 0         // timing-only ablations, compiled only under GS_EXPERIMENT: 1 = no activation function, 2 = no MFMAs, 4 = no weight DMA
#endif
#if defined(GS_EXPERIMENT) && (GS_H1R_ABL & 2)
#define R1_MFMA(a, b, c) (c)
#else
#define R1_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif
#ifndef GS_H1R_PD
#define GS_H1R_PD 1          // weight fragments (LDS -> registers) in flight ahead of the MFMAs that use them (measured 1 / 2 / 3: 2.38 / 2.44 / 2.52 ms:
                             // the second wave of the SIMD covers the LDS latency, the ring's registers cost spills)
#endif
#ifndef GS_H1R_FENCE
#define GS_H1R_FENCE 1       // 1: scheduling fences pin [fragment read, MFMA, half a pair's activation function] per k-step
#endif
#if GS_H1R_FENCE
#define R1_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define R1_FENCE()
#endif

// The activation function in the kernel's SCALED variables.  With c = 100 log2(e), softplus_100(z) = max(z, 0) + log2(1 + 2^-|c z|) / c, so in
// z' = c z and a' = c a it reads  a' = max(z', 0) + log2(1 + 2^-|z'|): no multiplications.  The scaling is free: z' = W a' + c b for the ORIGINAL
// W of every layer fed by activations (k_h2_pack scales the biases, the weights that multiply the encoding, and the output weights by 1 / c).
// Two halves, one behind each of two consecutive MFMAs.
struct R1Half { float z0, z1, l0, l1; };
__device__ __forceinline__ R1Half r1_act_a(float z0, float z1) {
    R1Half q;
    q.z0 = z0; q.z1 = z1;
#if defined(GS_EXPERIMENT) && (GS_H1R_ABL & 1)      // timing only (WRONG results): no activation function
    q.l0 = 0.f; q.l1 = 0.f;
#else
    const float e0 = __builtin_amdgcn_exp2f(-fabsf(z0)) + 1.0f, e1 = __builtin_amdgcn_exp2f(-fabsf(z1)) + 1.0f;
    q.l0 = __builtin_amdgcn_logf(e0);
    q.l1 = __builtin_amdgcn_logf(e1);
#endif
    return q;
}
__device__ __forceinline__ f2 r1_act_b(const R1Half& q) {
    float m0, m1;
    // plain v_max_f32: fmaxf / fmed3 cost a canonicalising v_max on top (the unused second input keeps the asm behind the logarithm)
    asm("v_max_f32 %0, 0, %1" : "=v"(m0) : "v"(q.z0), "v"(q.l0));
    asm("v_max_f32 %0, 0, %1" : "=v"(m1) : "v"(q.z1), "v"(q.l1));
    return f2{m0 + q.l0, m1 + q.l1};
}

struct R1State {
    int c, n_chunks;          // next chunk to be COMPUTED, chunks in total
    int cur, nxt, nx2;        // LDS buffer (h8 offset) of chunk c, c + 1, c + 2
};

// One layer of the register-resident forward for the wave's 32 rows.  FIRST: K = the encoding (3 k-steps); SKIP: 16 + 3 k-steps; LAST: the activation is
// multiplied by the output weights and summed instead of becoming the next layer's fragments.  Everything is unrolled and BRANCH-FREE -- one basic
// block per layer: a branch anywhere lets the compiler sink every block's activation function into the layer's last block, where its results are
// first used.  The accumulator of feature block mb - 1 goes through the activation function while block mb's MFMAs run: one pair of values per two
// k-steps.  A wave's own VALU work is NOT hidden behind its own MFMAs on this hardware (tools/micro/mfma_fillers.hip: 6 v_fma between two MFMAs cost
// 20 of their own cycles) -- the second wave of the SIMD is what fills the matrix pipe meanwhile.
template <bool FIRST, bool SKIP, bool LAST>
__device__ __forceinline__ void r1_layer(const H2Args& A, int l, R1State& S, h8* smem_w, const float* tailL, int tid, int lane, int h, const h8* Benc,
                                         const h8 (&Bin)[D / 16], h8 (&Bout)[D / 16], f2& part) {
    constexpr int NS = FIRST ? EK / 16 : D / 16 + (SKIP ? EK / 16 : 0);
    constexpr int PD = GS_H1R_PD < NS ? GS_H1R_PD : NS - 1;
    const float* bl = tailL + l * D + 4 * h;                    // this lane's bias entries: + 32 mb + 8 g
    const float* wo = tailL + A.n_layers * D + 4 * h;           // (LAST) output weights / c
    v16f accs[2];                     // block mb accumulates in accs[mb & 1] while the activation function reads accs[(mb - 1) & 1]
    float4 wqs[2][4];                 // (LAST) the blocks' output weights, same parity
    uint32_t pend[4];                 // packed activations of the fragment being assembled
    R1Half half;                      // a pair between its two halves
    h8 fr[PD + 1];                    // fragment ring; the first PD fragments of a chunk are requested at the end of the previous one
    {
        const h8* w0 = smem_w + S.cur + lane;
#pragma unroll
        for (int i = 0; i < PD; ++i) fr[i] = w0[i * 64];
    }
    R1_FENCE();
#pragma unroll
    for (int mb = 0; mb <= 8; ++mb) {
        v16f& acc = accs[mb & 1];
        v16f& accP = accs[(mb + 1) & 1];
        float4 (&wq)[4] = wqs[mb & 1];
        float4 (&wP)[4] = wqs[(mb + 1) & 1];
        if (mb < 8) {
            // chunk c + 2: L2 -> its buffer (last read two barriers ago); it lands during this block and is published by the barrier that ends it
            // (unconditional: past the end the last chunk is fetched again into a buffer nobody reads)
#if !(defined(GS_EXPERIMENT) && (GS_H1R_ABL & 4))
            {
                const int c2 = min(S.c + 2, S.n_chunks - 1), l2 = c2 >> 3;
                r1_dma_chunk(A.wfragR[l2] + (int64_t)(c2 & 7) * (r1_steps(l2, A.skip_layer) * 64), smem_w + S.nx2, tid);
            }
#endif
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = *reinterpret_cast<const float4*>(bl + mb * 32 + 8 * g);          // LDS (broadcast reads)
                if (LAST) wq[g] = *reinterpret_cast<const float4*>(wo + mb * 32 + 8 * g);
                acc[4 * g] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
            }
            R1_FENCE();
        }
        // pair pp (0..7) of the previous block: accumulator elements 2 pp, 2 pp + 1
        auto pair_a = [&](int pp) { half = r1_act_a(accP[2 * pp], accP[2 * pp + 1]); };
        auto pair_b = [&](int pp) {
            const int v = 2 * pp, pb = mb > 0 ? mb - 1 : 0;
            const f2 a = r1_act_b(half);
            if (LAST) {
                const float4 w4 = wP[v >> 2];
                part.x = __builtin_fmaf(a.x, (v & 2) ? w4.z : w4.x, part.x);
                part.y = __builtin_fmaf(a.y, (v & 2) ? w4.w : w4.y, part.y);
            } else {
                // four consecutive pairs are one fragment of the next layer: elements 8 s .. 8 s + 7 of feature block pb = k-step 2 pb + s.  The
                // fragment is assembled from its four packed words and assigned ONCE: element-wise inserts into the fragment array made every
                // insert a read-modify-write of a 128-bit tuple for the register allocator (1 500 spilled registers)
                const h2 q2 = __builtin_convertvector(a, h2);
                pend[pp & 3] = __builtin_bit_cast(uint32_t, q2);
                if ((pp & 3) == 3) {
                    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                    Bout[2 * pb + (pp >> 2)] = __builtin_bit_cast(h8, u4{pend[0], pend[1], pend[2], pend[3]});
                }
            }
        };
        if (mb < 8) {
            const h8* wl = smem_w + S.cur + lane;
            const h8* wn = smem_w + S.nxt + lane;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                // request: this chunk's fragment s + PD, or (its last PD steps) the NEXT chunk's first fragments -- visible since the last barrier
                // (fragment s of block mb lives in ring slot (mb NS + s) % (PD + 1): the ring runs on across the blocks of a layer)
                const int q0 = mb * NS + s;
                if (s + PD < NS) fr[(q0 + PD) % (PD + 1)] = wl[(s + PD) * 64];
                else if (mb < 7) fr[(q0 + PD) % (PD + 1)] = wn[(s + PD - NS) * 64];
                const h8 a = fr[q0 % (PD + 1)];
                // (the encoding's fragments come from LDS: two layers read them, 12 registers would hold them for all seven)
                const h8 b = (FIRST || s >= D / 16) ? Benc[(FIRST ? s : s - D / 16) * 64] : Bin[s < D / 16 ? s : 0];
                acc = R1_MFMA(a, b, acc);
                R1_FENCE();
                if (mb > 0 && NS >= 16 && s < 16) {        // the previous block's 8 pairs ride on this block's first 16 k-steps
                    if (s & 1) pair_b(s >> 1); else pair_a(s >> 1);
                    R1_FENCE();
                }
            }
        }
        if (mb > 0 && (mb == 8 || NS < 16)) {          // not interleaved: the first layer's short k-loop, and every layer's last block
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                pair_a(pp);
                pair_b(pp);
                R1_FENCE();
            }
        }
        if (mb < 8) {
            r1_dma_wait();        // this wave's pieces of chunk c + 2 have landed (requested a block of MFMAs ago)
            __syncthreads();      // every wave is done with chunk c's buffer; chunk c + 2 is visible
            ++S.c;
            const int t = S.cur;
            S.cur = S.nxt; S.nxt = S.nx2; S.nx2 = t;
        }
    }
}

#if GS_ORACLE_KERNELS      // (common.hpp: alternate design, built into lib/variants/oracles.so only)
__global__ void __launch_bounds__(R1_NT, 2) k_h1r_fwd(H2Args A) {
    extern __shared__ __attribute__((aligned(16))) h8 smem_w[];           // [R1_BUFS][R1_CHUNK] | encoding fragments | scaled biases + output weights
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * R1_NW + wave, r0 = tile * 32;

    // ---- weight chunks c = 8 l + mb: chunks 0, 1 -> LDS now, chunk c + 2 during chunk c
    R1State S{0, A.n_layers * 8, 0, R1_CHUNK, 2 * R1_CHUNK};
    auto chunk_src = [&](int c) { const int l = c >> 3; return A.wfragR[l] + (int64_t)(c & 7) * (r1_steps(l, A.skip_layer) * 64); };
    r1_dma_chunk(chunk_src(0), smem_w, tid);            // n_chunks >= 8
    r1_dma_chunk(chunk_src(1), smem_w + R1_CHUNK, tid);
    float* tailL = reinterpret_cast<float*>(smem_w + R1_BUFS * R1_CHUNK + R1_ENC);
    for (int i = tid; i < (A.n_layers + 1) * D; i += R1_NT) tailL[i] = A.tailR[i];
    r1_dma_wait();

    // ---- positional encoding of the wave's 32 rows, as B fragments: k-slot (s', h, i) = encoding entry 16 s' + 8 h + i
    h8* Benc = smem_w + R1_BUFS * R1_CHUNK + wave * ((EK / 16) * 64) + lane;        // [k-step][lane]
    {
        const int64_t row = r0 + n;
        float p[3] = {0.f, 0.f, 0.f};
        if (row < A.N) { p[0] = A.x[3 * row]; p[1] = A.x[3 * row + 1]; p[2] = A.x[3 * row + 2]; }
        float e[EK];
#pragma unroll
        for (int i = 0; i < EK; ++i) e[i] = 0.f;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) e[ax] = fminf(fmaxf(p[ax], -60000.0f), 60000.0f);
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const float rev = p[ax] * ((float)(1 << k) * 0.15915494309189535f);          // v_sin / v_cos take revolutions
                const bool on = k < A.n_freq && row < A.N;
                e[3 + 6 * k + ax] = on ? __builtin_amdgcn_sinf(rev) : 0.f;
                e[3 + 6 * k + 3 + ax] = on ? __builtin_amdgcn_cosf(rev) : 0.f;
            }
#pragma unroll
        for (int s = 0; s < EK / 16; ++s) {
            h8 f;
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = (_Float16)(h ? e[16 * s + 8 + i] : e[16 * s + i]);
            Benc[s * 64] = f;
        }
    }
    __syncthreads();

    h8 Ba[D / 16], Bb[D / 16];          // the layers' input / output fragments, roles alternating (no copy between layers)
    f2 part = f2{0.f, 0.f};
    for (int l = 0; l < A.n_layers; ++l) {
        const bool last = l + 1 == A.n_layers, skip = l == A.skip_layer;
#define R1_CALL(F, K, L_)                                                                                      \
        do {                                                                                                   \
            if (l & 1) r1_layer<F, K, L_>(A, l, S, smem_w, tailL, tid, lane, h, Benc, Bb, Ba, part);           \
            else r1_layer<F, K, L_>(A, l, S, smem_w, tailL, tid, lane, h, Benc, Ba, Bb, part);                 \
        } while (0)
        if (l == 0) {
            if (last) R1_CALL(true, false, true);
            else R1_CALL(true, false, false);
        } else if (last) {
            if (skip) R1_CALL(false, true, true);
            else R1_CALL(false, false, true);
        } else {
            if (skip) R1_CALL(false, true, false);
            else R1_CALL(false, false, false);
        }
#undef R1_CALL
    }
    // ---- output layer: the two lane halves of a row hold disjoint feature sets; lane l < 32 <-> row r0 + l
    float sv = part.x + part.y;
    sv += __shfl_xor(sv, 32, 64);
    sv += A.w_out[D];
    const int64_t row = r0 + n;
    const bool valid = row < A.N && h == 0;
    if (valid) A.out[row] = sv;
    const uint64_t m = __ballot(valid && sv > 0.0f);
    // the occupancy words are 64 rows wide: this wave owns the low (even tile) or high (odd tile) 32 bits
    if (A.occ && lane == 0 && r0 < A.N) reinterpret_cast<uint32_t*>(A.occ)[tile] = (uint32_t)m;
    if (A.status && __ballot(valid && !(fabsf(sv) < 3.0e38f)) != 0ull && lane == 0) atomicOr(&A.status[ST_NONFINITE], 1u);
}
#endif  // GS_ORACLE_KERNELS

// ---- backward chain ------------------------------------------------------------------------------------------------
// Per 64-row tile, from the top: G = g_out (x) w_out;  for l = L-1 .. 0:  D_l = (dL/dz_l) from G and the saved activations
// (elementwise, in the accumulator layout);  D_l -> HBM (fp32, for the weight-gradient kernel) and -> LDS as fp16 pairs;
// G = W_l^T D_l  (the same transposed MFMA with the transposed fragment set).  The adjoint of the encoding (layer 0 and the
// skip layer) is accumulated in LDS and turned into dL/dx at the end (ROWS).  Every row is scaled by a power of two so that
// |g_out| is in [1,2) inside the chain (the chain is linear in g_out; gradients of 1e-7 would otherwise sit in the fp16
// denormal range of the pair split) and unscaled on the way out (exact).
struct BwdArgs {
    const float* g_out;   // [Rpad] upstream gradient per (virtual) row; 0 on padding rows
    const float* A;       // planes [n_layers][Rpad / 32][256][32] (plane_idx)
    const float* EMB;     // [Rpad][EK]
    float* Dsave;         // planes [n_layers][Rpad / 32][256][32]  WRITTEN
    const int32_t* rows;  // ROWS: [R]
    float* g_x;           // ROWS: [N,3] scatter target (rows are unique), or null
    int64_t R, Rpad;
    const int64_t* n_dev; // optional device-resident row count (<= R), see H2Args
    int E, n_layers, skip_layer;
    const h8* wfragT[MAX_LAYERS];   // [n-step 16][block nbT(l)][piece 2][lane 64]
    int nblkT[MAX_LAYERS];
    const float* w_out;
    int64_t Pstride;      // rows per layer of the planes; 0 = Rpad
    // MODE_RR (adjoint of the value pass with the second-order source): Dsave holds S_l on entry and zbar_l on exit; g_out is not read
    const float* gbar;    // [R][3], gmul: device scalar -- the rows' power-of-two scales, as in the tangent pass
    const float* gmul;
};

constexpr int LDG = EK + 1;      // fp32 encoding-adjoint tile row stride

template <int MODE>
__global__ void __launch_bounds__(NT, 4) k_h2_bwd(BwdArgs B) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    _Float16* H1 = smem_h;                    // [TM][LDH]   D_l pieces
    _Float16* H2 = H1 + TM * LDH;
    float* GE = reinterpret_cast<float*>(H2 + TM * LDH);      // [TM][LDG] adjoint of the encoding (ROWS)
    float* SC = GE + TM * LDG;                                 // [TM] 1 / row scale
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * TM;
    const int64_t n_act = B.n_dev ? min(*B.n_dev, B.R) : B.R;
    if (MODE == MODE_ROWS && r0 >= n_act) return;
    const int n_base = wave * 32 + 4 * (lane >> 5);
    const int m_lane = lane & 31;
    const bool low16 = (lane & 16) == 0;
    const bool need_x = MODE == MODE_ROWS && B.g_x != nullptr;

    const int64_t PS = B.Pstride ? B.Pstride : B.Rpad;
    float go[2] = {0.f, 0.f};
    if (MODE != MODE_RR) { go[0] = B.g_out[r0 + m_lane]; go[1] = B.g_out[r0 + 32 + m_lane]; }
    float sc[2];
    if (MODE == MODE_RR) {
        const float gm = *B.gmul;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int64_t r = r0 + 32 * s + m_lane;
            sc[s] = r < n_act ? pow2_scale_for(fmaxf(fmaxf(fabsf(B.gbar[3 * r] * gm), fabsf(B.gbar[3 * r + 1] * gm)), fabsf(B.gbar[3 * r + 2] * gm))) : 1.0f;
        }
    } else if (MODE == MODE_EIK) {
        float gm = fmaxf(fabsf(go[0]), fabsf(go[1]));
        gm = fmaxf(gm, __shfl_xor(gm, 16, 64));              // the four virtual rows of a sample share one scale
        sc[0] = sc[1] = pow2_scale_for(gm);
    } else {
        sc[0] = pow2_scale_for(fabsf(go[0]));
        sc[1] = pow2_scale_for(fabsf(go[1]));
    }
    const float isc[2] = {1.0f / sc[0], 1.0f / sc[1]};       // exact (powers of two)
    go[0] *= sc[0];
    go[1] *= sc[1];
    if (need_x) {
        for (int i = tid; i < TM * LDG; i += NT) GE[i] = 0.f;
        if (wave == 0 && lane < 32) { SC[lane] = isc[0]; SC[32 + lane] = isc[1]; }
    }
    // adjoint of the last hidden activation, in the accumulator layout: G[s][4 g + j] <-> feature n_base + 8 g + j of row 32 s + m
    v16f G[2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 w4 = *reinterpret_cast<const float4*>(B.w_out + n_base + 8 * g);
        const float wj[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            G[0][4 * g + j] = go[0] * wj[j];
            G[1][4 * g + j] = go[1] * wj[j];
        }
    }
    if (need_x) __syncthreads();

    // the saved outputs of a layer are fetched one layer AHEAD (issued before the GEMM of the layer above, consumed after its
    // closing barrier): per-lane 16-byte pieces at a 1 KB row stride are latency, not bandwidth
#ifndef GS_H2_ABL
#define GS_H2_ABL 0      // experiments (tools/build_variant.sh): 1 = no D-plane stores, 2 = no A-plane loads, 4 = no dgrad GEMM
#endif
    float4 an[2][4];
    float4 sn[2][4];        // RR: the source S_l of the same elements (read where zbar_l is written)
    auto fetch_plane = [&](int l) {
        if (GS_H2_ABL & 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) an[0][g] = an[1][g] = make_float4(0.01f * (float)(l + g), 0.02f, 0.03f, 0.04f);
            return;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t i0 = plane_idx(PS, l, r0 + m_lane, n_base + 8 * g), i1 = plane_idx(PS, l, r0 + 32 + m_lane, n_base + 8 * g);
            const float *p0 = B.A + i0, *p1 = B.A + i1;
            an[0][g] = make_float4(p0[0], p0[32], p0[64], p0[96]);
            an[1][g] = make_float4(p1[0], p1[32], p1[64], p1[96]);
            if (MODE == MODE_RR) {
                const float *q0 = B.Dsave + i0, *q1 = B.Dsave + i1;
                sn[0][g] = make_float4(q0[0], q0[32], q0[64], q0[96]);
                sn[1][g] = make_float4(q1[0], q1[32], q1[64], q1[96]);
            }
        }
    };
    fetch_plane(B.n_layers - 1);
    for (int l = B.n_layers - 1; l >= 0; --l) {
        // ---- dL/dz_l from G (adjoint of the layer's OUTPUT) and the saved outputs
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 a0 = an[0][g];
            const float4 a1 = an[1][g];
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
            float d0[4], d1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * g + j;
                if (MODE == MODE_RR) {
                    const float sv0[4] = {sn[0][g].x, sn[0][g].y, sn[0][g].z, sn[0][g].w}, sv1[4] = {sn[1][g].x, sn[1][g].y, sn[1][g].z, sn[1][g].w};
                    d0[j] = __builtin_fmaf(G[0][r], slope_from_value(av0[j]), sv0[j] * sc[0]);
                    d1[j] = __builtin_fmaf(G[1][r], slope_from_value(av1[j]), sv1[j] * sc[1]);
                } else if (MODE != MODE_EIK) {
                    d0[j] = G[0][r] * slope_from_value(av0[j]);
                    d1[j] = G[1][r] * slope_from_value(av1[j]);
                } else {
                    // low16 lanes: (value a, d/dy a') of the sample; the lanes 16 above: (d/dx a', d/dz a').
                    //   D_tangent = s G_tangent;   D_value = s G_value + 100 (1 - s) sum_d G_d a'_d        (a'_d = s z'_d)
                    const float sl = slope_from_value(av0[j]);                   // meaningful on low16 lanes only
                    const float pl = G[0][r] * av0[j] + G[1][r] * av1[j];        // meaningful on the upper lanes: G_x a'_x + G_z a'_z
                    const float so = __shfl_xor(low16 ? sl : pl, 16, 64);        // low16 receives p of its partner, upper receives s
                    if (low16) {
                        d0[j] = G[0][r] * sl + 100.0f * (1.0f - sl) * (so + G[1][r] * av1[j]);
                        d1[j] = sl * G[1][r];
                    } else {
                        d0[j] = so * G[0][r];
                        d1[j] = so * G[1][r];
                    }
                }
            }
            if (!(GS_H2_ABL & 1)) {
                float* dp0 = B.Dsave + plane_idx(PS, l, r0 + m_lane, n_base + 8 * g);
                float* dp1 = B.Dsave + plane_idx(PS, l, r0 + 32 + m_lane, n_base + 8 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dp0[32 * j] = d0[j] * isc[0];
                    dp1[32 * j] = d1[j] * isc[1];
                }
            }
            if (l > 0 || need_x) {
                h2 p0, q0, p1, q1;
                split_h2_pair(f2{d0[0], d0[1]}, p0, q0);
                split_h2_pair(f2{d0[2], d0[3]}, p1, q1);
                int off = m_lane * LDH + n_base + 8 * g;
                *reinterpret_cast<h4*>(H1 + off) = h4{p0.x, p0.y, p1.x, p1.y};
                *reinterpret_cast<h4*>(H2 + off) = h4{q0.x, q0.y, q1.x, q1.y};
                split_h2_pair(f2{d1[0], d1[1]}, p0, q0);
                split_h2_pair(f2{d1[2], d1[3]}, p1, q1);
                off += 32 * LDH;
                *reinterpret_cast<h4*>(H1 + off) = h4{p0.x, p0.y, p1.x, p1.y};
                *reinterpret_cast<h4*>(H2 + off) = h4{q0.x, q0.y, q1.x, q1.y};
            }
        }
#ifndef GS_H2_BWD_PREFETCH
#define GS_H2_BWD_PREFETCH 0     // 1: issue the next plane's loads BEFORE the GEMM (spills: 71 / 22 VGPRs), 0: after it, before the closing
#endif                           //    barrier; 2: before for <EIK>, after for <ROWS>.  Measured (both calls, ms): HEAD 1.78, 1 -> 1.67, 0 -> 1.47
        constexpr bool EARLY = GS_H2_BWD_PREFETCH == 1 || (GS_H2_BWD_PREFETCH == 2 && MODE == MODE_EIK);
        if (EARLY && l > 0) fetch_plane(l - 1);
        if (l == 0 && !need_x) break;
        __syncthreads();
        // ---- G = W_l^T D_l
        v16f hi[2], lo[2];
        // the adjoint of the encoding FIRST (two 32-feature blocks, 48 used): the previous G is dead here, so the two GEMMs never hold G and a
        // second accumulator set at once (after the main GEMM they did: 62 spilled registers in <ROWS>)
        if (need_x && (l == 0 || l == B.skip_layer) && wave < 2) {
            zero_acc(hi, lo);
            gemm_seg<LDH, D / 16>(hi, lo, H1, H2, B.wfragT[l], (l == 0 ? 0 : 8) + wave, B.nblkT[l], lane);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = 32 * wave + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
                    if (f < EK) GE[(32 * s + m_lane) * LDG + f] += __builtin_fmaf(lo[s][r], LO_INV, hi[s][r]);
                }
        }
        if (l > 0) {
            zero_acc(hi, lo);
            if (!(GS_H2_ABL & 4)) gemm_seg<LDH, D / 16>(hi, lo, H1, H2, B.wfragT[l], wave, B.nblkT[l], lane);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) G[s][r] = __builtin_fmaf(lo[s][r], LO_INV, hi[s][r]);
        } else {
            // G is not read again (l == 0 is the last pass), but the compiler does not see that: without this redefinition it keeps the OLD G
            // alive across the encoding GEMM above -- 30 registers spilled and reloaded around it in every layer
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) G[s][r] = 0.f;
        }
        if (!EARLY && l > 0) fetch_plane(l - 1);
        __syncthreads();
    }
    if (need_x) {
        // dL/dx_c = G[c] + sum_k 2^k (cos(2^k x_c) G[sin_k,c] - sin(2^k x_c) G[cos_k,c]); sin / cos are the saved encoding
        if (tid < TM * 3) {
            const int row = tid / 3, c = tid - 3 * row;
            const int64_t r = r0 + row;
            if (r < n_act) {
                const float* ge = GE + row * LDG;
                const float* em = B.EMB + r * EK;
                float acc = ge[c];
                const int nf = (B.E - 3) / 6;
                for (int k = 0; k < nf; ++k) {
                    const float fr = (float)(1 << k);
                    acc += fr * (em[3 + 6 * k + 3 + c] * ge[3 + 6 * k + c] - em[3 + 6 * k + c] * ge[3 + 6 * k + 3 + c]);
                }
                B.g_x[3 * (B.rows ? (int64_t)B.rows[r] : r) + c] = acc * SC[row];        // rows == nullptr: the rows are x[0 .. R)
            }
        }
    }
}

// ---- weight gradients ----------------------------------------------------------------------------------------------
// dW_l[n][k] = sum_rows D_l[row][n] X_l[row][k],  X_l = the layer's input ([a_{l-1} | enc] for the skip layer, enc for layer 0),
// db_l[n] = sum over VALUE rows of D_l[row][n],  over ~4 10^5 (virtual) rows: a 256 x 256 output with the rows as the reduction
// dimension.  One workgroup owns the whole [256 x K] output of ONE layer for a strip of rows (grid = strips x layers), streams
// 32-row slabs of D_l and X_l through a double-buffered LDS image (registers in between) and accumulates on the matrix cores
// with the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: both fragments are single floats, so the row-major fp32 slabs are read as
// they lie -- no transposition).  Wave w owns features [32 w, 32 w + 32) x all K: up to 10 accumulator blocks.
// The bias gradient is the column sum of the D slab, taken from the staging registers (each thread always holds the same four
// features); partial results of the strips are combined with float atomics into the (zeroed) torch-layout gradient tensors.
struct WgradArgs {
    const float* A;       // planes [n_layers][Rpad / 32][256][32] (plane_idx)
    const float* EMB;     // [Rpad][EK]
    const float* D;       // planes, same layout
    const float* g_out;   // [Rpad]
    int64_t Rpad, n;      // rows of the planes; rows in use (ROWS: the count or its capacity; EIK: all of Rpad)
    const int64_t* n_dev; // optional device-resident row count: only the tiles below it are reduced
    int E, n_layers, skip_layer, mode;
    int slabs_per_strip;  // 32-row slabs per workgroup
    int only_output;      // 1: k_h2_wgrad handles the output layer only (the hidden layers run in k_h2_wgrad16)
    int wg_first[MAX_LAYERS + 1];   // k_h2_wgrad16 (1-D grid): workgroups [wg_first[l], wg_first[l + 1]) reduce layer l -- in proportion to the layer's work
    float* dW[MAX_LAYERS + 1];     // torch layout [256][K_l]; [n_layers] = output layer [1][256]
    float* db[MAX_LAYERS + 1];     // [256]; the output layer's bias gradient is the caller's
};

__device__ __forceinline__ int64_t wgrad_rows(const WgradArgs& W) {      // rows to reduce: whole 64-row tiles below the (device-side) count
    // the chain kernels skip tiles that lie entirely past the row count, so those plane rows are never written
    const int64_t n = W.n_dev ? min(*W.n_dev, W.n) : W.n;
    return min((n + TM - 1) / TM * TM, W.Rpad);
}

constexpr int WS = 32;                       // rows per slab
constexpr int WG_D = WS * D;                 // floats of the D / X_h slab
constexpr int WG_E = WS * 64;                // encoding slab, padded to 64 columns
constexpr int WG_BUF = 2 * WG_D + WG_E;      // one LDS buffer (floats)

template <int NB, bool HAS_H, bool HAS_E>
__device__ __forceinline__ void wgrad_layer(const WgradArgs& W, int l, float* smem, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    // slabs are dealt round-robin to the strips (workgroups): any prefix of valid slabs (device-side row count) stays balanced
    const int64_t nslabs_total = wgrad_rows(W) / WS, stride = gridDim.x, slab0 = blockIdx.x;
    const int64_t nslab = slab0 < nslabs_total ? (nslabs_total - slab0 + stride - 1) / stride : 0;
    if (nslab <= 0) return;
    const float* Dl = W.D + (int64_t)l * W.Rpad * D;
    const float* Xh = HAS_H ? W.A + (int64_t)(l - 1) * W.Rpad * D : nullptr;
    v16f acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    float4 dreg[4], xreg[4], ereg;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    // thread t stages float4 chunk c = t + 512 i of a [32][256] slab: row = c / 64 = t / 64 + 8 i, feature quad = t % 64 (fixed)
    auto load_slab = [&](int64_t slab) {
        const int64_t rbase = slab * WS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t row = rbase + (tid >> 6) + 8 * i;
            const int64_t pi = plane_idx(0, 0, row, 4 * (tid & 63));        // planes are [slab][feature][32 rows]
            dreg[i] = make_float4(Dl[pi], Dl[pi + 32], Dl[pi + 64], Dl[pi + 96]);
            if (HAS_H) xreg[i] = make_float4(Xh[pi], Xh[pi + 32], Xh[pi + 64], Xh[pi + 96]);
        }
        if (HAS_E && tid < WS * (EK / 4)) ereg = *reinterpret_cast<const float4*>(W.EMB + (rbase + tid / (EK / 4)) * EK + 4 * (tid % (EK / 4)));
    };
    auto store_slab = [&](float* buf, int64_t slab) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 6) + 8 * i;
            *reinterpret_cast<float4*>(buf + row * D + 4 * (tid & 63)) = dreg[i];
            if (HAS_H) *reinterpret_cast<float4*>(buf + WG_D + row * D + 4 * (tid & 63)) = xreg[i];
            // bias gradient: value rows only (EIK: tile rows 0..15 of every 64)
            const bool value_row = W.mode == MODE_RR ? 2 * slab * WS < W.Rpad : (W.mode != MODE_EIK || (((slab * WS + row) & 63) < 16));      // RR: the zbar half
            if (value_row) { bsum.x += dreg[i].x; bsum.y += dreg[i].y; bsum.z += dreg[i].z; bsum.w += dreg[i].w; }
        }
        if (HAS_E && tid < WS * (EK / 4)) *reinterpret_cast<float4*>(buf + 2 * WG_D + (tid / (EK / 4)) * 64 + 4 * (tid % (EK / 4))) = ereg;
    };
    if (HAS_E)      // columns EK..63 of both encoding slabs stay zero
        for (int i = tid; i < 2 * WS * 16; i += NT) smem[(i / (WS * 16)) * WG_BUF + 2 * WG_D + ((i / 16) % WS) * 64 + EK + (i & 15)] = 0.f;
    load_slab(slab0);
    store_slab(smem, slab0);
    __syncthreads();
    for (int64_t s = 0; s < nslab; ++s) {
        float* cur = smem + (s & 1) * WG_BUF;
        const bool more = s + 1 < nslab;
        if (more) load_slab(slab0 + (s + 1) * stride);
        const float* ap = cur + (lane >> 5) * D + wave * 32 + (lane & 31);            // D[row = 2 ks + (lane >> 5)][n]
        const float* xp = cur + WG_D + (lane >> 5) * D + (lane & 31);
        const float* ep = cur + 2 * WG_D + (lane >> 5) * 64 + (lane & 31);
#pragma unroll 4
        for (int ks = 0; ks < WS / 2; ++ks) {
            const float a = ap[ks * 2 * D];
            if (HAS_H) {
#pragma unroll
                for (int b = 0; b < 8; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xp[ks * 2 * D + 32 * b], acc[b], 0, 0, 0);
            }
            if (HAS_E) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[(HAS_H ? 8 : 0) + b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, ep[ks * 2 * 64 + 32 * b], acc[(HAS_H ? 8 : 0) + b], 0, 0, 0);
            }
        }
        if (more) store_slab(smem + ((s + 1) & 1) * WG_BUF, slab0 + (s + 1) * stride);
        __syncthreads();
    }
    // flush: acc[b][reg] = dW[n = 32 wave + (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)][k = 32 b + (lane & 31)]
    const int Kreal = (HAS_H ? D : 0) + (HAS_E ? W.E : 0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        int k = 32 * b + (lane & 31);
        bool ok = true;
        if (HAS_E && b >= (HAS_H ? 8 : 0)) {
            const int e = 32 * (b - (HAS_H ? 8 : 0)) + (lane & 31);
            ok = e < W.E;
            k = (HAS_H ? D : 0) + e;
        }
        if (ok)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                atomicAdd(W.dW[l] + (int64_t)n * Kreal + k, acc[b][r]);
            }
    }
    // bias gradient: the 8 waves hold partial sums of the same feature quads
    float4* red = reinterpret_cast<float4*>(smem);
    red[tid] = bsum;
    __syncthreads();
    if (tid < 64) {
        float4 t = red[tid];
        for (int w = 1; w < 8; ++w) { const float4 o = red[w * 64 + tid]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
        atomicAdd(W.db[l] + 4 * tid, t.x);
        atomicAdd(W.db[l] + 4 * tid + 1, t.y);
        atomicAdd(W.db[l] + 4 * tid + 2, t.z);
        atomicAdd(W.db[l] + 4 * tid + 3, t.w);
    }
}

// ---- weight gradients on the bf16 matrix path (default) -----------------------------------------------------------------
// Same job as wgrad_layer above at 3/16 of its matrix-core time: D and X are split into bf16 pairs v = hi + lo (bf16 keeps the
// fp32 exponent range, so the gradient planes need no scaling; 16 significant bits per operand, products hi hi + hi lo + lo hi
// = relative 2^-16 per term, unbiased, over >= 10^5 rows) and multiplied with v_mfma_f32_32x32x16_bf16 into ONE fp32
// accumulator.  The MFMA reduces over ROWS here, so both fragments need 8 consecutive rows per lane: the fp32 slabs are
// transposed on the way into LDS -- lane = feature, 4 consecutive rows per thread -> one ds_write_b64 per (feature, piece) into
// a [feature][32 rows + 8 pad] image whose 80-byte row stride keeps the ds_read_b128 fragment reads conflict-free.
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
constexpr int WB_RS = WS + 8;                               // image row stride (bf16 elements): 80 B = 5 sixteen-byte slots
constexpr int WB_FEATS = D + D + 64;                        // D_l | X_h | X_e (64 = encoding padded)
constexpr int WB_PLANE = WB_FEATS * WB_RS;                  // one piece
constexpr size_t SMEM_WGRAD16_BYTES = (size_t)2 * WB_PLANE * sizeof(__bf16);

__device__ __forceinline__ void split_bf16_4(const float (&v)[4], bf4& hi, bf4& lo) {
    const f2 a = {v[0], v[1]}, b = {v[2], v[3]};
    const bf2 ha = __builtin_convertvector(a, bf2), hb = __builtin_convertvector(b, bf2);
    const bf2 la = __builtin_convertvector(a - __builtin_convertvector(ha, f2), bf2);
    const bf2 lb = __builtin_convertvector(b - __builtin_convertvector(hb, f2), bf2);
    hi = bf4{ha.x, ha.y, hb.x, hb.y};
    lo = bf4{la.x, la.y, lb.x, lb.y};
}

#ifndef GS_WG_PIPE
#define GS_WG_PIPE 0     // 1: reload each register piece right after it has been staged (measured: 1.55 vs 1.46 ms without, both passes)
#endif
template <int NB, bool HAS_H, bool HAS_E>
__device__ __forceinline__ void wgrad16_layer(const WgradArgs& W, int l, __bf16* img, int tid, int strip, int nstrips) {
    const int lane = tid & 63, wave = tid >> 6;
#ifndef GS_WG_CONTIG
#define GS_WG_CONTIG 1   // 1: a workgroup reduces a CONTIGUOUS range of slabs (of the device-side row count, so any count stays balanced); 0: round-robin
#endif
    const int64_t nslabs_total = wgrad_rows(W) / WS;
    const int64_t per = (nslabs_total + nstrips - 1) / nstrips;
    const int64_t stride = GS_WG_CONTIG ? 1 : (int64_t)nstrips, slab0 = GS_WG_CONTIG ? strip * per : (int64_t)strip;
    const int64_t nslab = slab0 < nslabs_total ? (GS_WG_CONTIG ? min(per, nslabs_total - slab0) : (nslabs_total - slab0 + stride - 1) / stride) : 0;
    if (nslab <= 0) return;
    const float* Dl = W.D + (int64_t)l * W.Rpad * D;
    const float* Xh = HAS_H ? W.A + (int64_t)(l - 1) * W.Rpad * D : nullptr;
    __bf16* hi_img = img;
    __bf16* lo_img = img + WB_PLANE;
    v16f acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    // staging: the planes are [slab][feature][32 rows] (plane_idx), so FOUR consecutive rows of one feature -- the 8-byte piece of the
    // transposed image -- are one 16-byte load: this thread owns rows 4 (lane & 7) .. + 3 of features 32 wave + 8 c + (lane >> 3),
    // c = 0..3 (a wave instruction reads 8 features x 32 rows = 1 KB contiguous; the row-major planes needed 16 dword loads per
    // plane and thread: 0.64 of this kernel's 1.65 ms were those loads).  Encoding: rows 4 wave .. + 3 of feature `lane` (< EK).
    struct Staged {
        float d[4][4], x[4][4], e[4];
    };
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    const int frow = 4 * (lane & 7), fsub = 32 * wave + (lane >> 3);
    auto load_slab = [&](Staged& R, int64_t slab) {
        const float* ds = Dl + slab * (int64_t)(WS * D) + frow;
        const float* xs = HAS_H ? Xh + slab * (int64_t)(WS * D) + frow : nullptr;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 dv = *reinterpret_cast<const float4*>(ds + (fsub + 8 * c) * WS);
            R.d[c][0] = dv.x; R.d[c][1] = dv.y; R.d[c][2] = dv.z; R.d[c][3] = dv.w;
            if (HAS_H) {
                const float4 xv = *reinterpret_cast<const float4*>(xs + (fsub + 8 * c) * WS);
                R.x[c][0] = xv.x; R.x[c][1] = xv.y; R.x[c][2] = xv.z; R.x[c][3] = xv.w;
            }
        }
        if (HAS_E) {
            const int64_t r0 = slab * WS + 4 * wave;
#pragma unroll
            for (int i = 0; i < 4; ++i) R.e[i] = lane < EK ? W.EMB[(r0 + i) * EK + lane] : 0.0f;
        }
    };
    // `next` >= 0: as soon as piece c of this slab is in the image its registers are reloaded with piece c of slab `next`, so the
    // loads fly during the rest of the staging, both barriers and the MFMAs (issued after the staging they only overlapped the MFMAs:
    // the loop ran at ~2 TB/s with HBM idle half of the time)
    auto store_slab = [&](Staged& R, int64_t slab, int64_t next) {
        const float* ds = Dl + next * (int64_t)(WS * D) + frow;
        const float* xs = HAS_H ? Xh + next * (int64_t)(WS * D) + frow : nullptr;
        // bias gradient: value rows only (EIK: tile rows 0..15 of every 64 = the even slab's rows 0..15 = this thread's four rows or
        // none of them); a multiplier, not a branch: sixteen divergent branches per slab kept the scheduler from overlapping anything
        // RR: rows [0, Rpad / 2) are the zbar rows, the delta rows below them carry no bias term
        const float vsel = (W.mode == MODE_RR ? 2 * slab * WS < W.Rpad : (W.mode != MODE_EIK || ((slab & 1) == 0 && frow < 16))) ? 1.0f : 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int f = fsub + 8 * c;
            bf4 h, lw;
            split_bf16_4(R.d[c], h, lw);
            *reinterpret_cast<bf4*>(hi_img + f * WB_RS + frow) = h;
            *reinterpret_cast<bf4*>(lo_img + f * WB_RS + frow) = lw;
            bsum[c] += vsel * ((R.d[c][0] + R.d[c][1]) + (R.d[c][2] + R.d[c][3]));
            if (HAS_H) {
                split_bf16_4(R.x[c], h, lw);
                *reinterpret_cast<bf4*>(hi_img + (D + f) * WB_RS + frow) = h;
                *reinterpret_cast<bf4*>(lo_img + (D + f) * WB_RS + frow) = lw;
            }
            if (GS_WG_PIPE && next >= 0) {
                const float4 dv = *reinterpret_cast<const float4*>(ds + f * WS);
                R.d[c][0] = dv.x; R.d[c][1] = dv.y; R.d[c][2] = dv.z; R.d[c][3] = dv.w;
                if (HAS_H) {
                    const float4 xv = *reinterpret_cast<const float4*>(xs + f * WS);
                    R.x[c][0] = xv.x; R.x[c][1] = xv.y; R.x[c][2] = xv.z; R.x[c][3] = xv.w;
                }
            }
        }
        if (HAS_E) {
            const int roff = 4 * wave;
            bf4 h, lw;
            split_bf16_4(R.e, h, lw);
            *reinterpret_cast<bf4*>(hi_img + (2 * D + lane) * WB_RS + roff) = h;
            *reinterpret_cast<bf4*>(lo_img + (2 * D + lane) * WB_RS + roff) = lw;
            if (GS_WG_PIPE && next >= 0) {
                const int64_t r0 = next * WS + 4 * wave;
#pragma unroll
                for (int i = 0; i < 4; ++i) R.e[i] = lane < EK ? W.EMB[(r0 + i) * EK + lane] : 0.0f;
            }
        }
    };
#ifndef GS_WG_ABL
#define GS_WG_ABL 0      // experiments: 1 = no MFMAs, 2 = LDS image written once, 4 = slab loaded once
#endif
#ifndef GS_WG_DEPTH
#define GS_WG_DEPTH 1    // slabs in flight ahead of the one being reduced (two register sets; one workgroup per CU: 92 KB image)
#endif
    // one slab: registers -> transposed bf16-pair image, then the next-but-one slab's loads go out and fly during TWO slabs of MFMAs
    // (a load issued one slab ahead arrived after the 0.7 us of MFMAs: the loop ran at the HBM latency, 6.4 us per slab)
    auto reduce_slab = [&](Staged& R, int64_t s) {
        const bool more = s + GS_WG_DEPTH < nslab && !(GS_WG_ABL & 4);
        if (!(GS_WG_ABL & 2) || s == 0) store_slab(R, slab0 + s * stride, more ? slab0 + (s + GS_WG_DEPTH) * stride : (int64_t)-1);
        __syncthreads();
        if (!GS_WG_PIPE && more) load_slab(R, slab0 + (s + GS_WG_DEPTH) * stride);
        const int roff = 8 * (lane >> 5);
        const __bf16* ah = hi_img + (wave * 32 + (lane & 31)) * WB_RS + roff;
        const __bf16* al = lo_img + (wave * 32 + (lane & 31)) * WB_RS + roff;
#ifndef GS_WG_PAIR
#define GS_WG_PAIR 1     // 1: accumulator blocks in PAIRS, their three products interleaved (an MFMA never follows the one just issued on its own
#endif                   //    accumulator), the next pair's four fragments requested before the current pair's six MFMAs (scheduling groups pin the order)
        auto feat_of = [&](int b) { return (HAS_H ? (b < 8 ? D + 32 * b : 2 * D + 32 * (b - 8)) : 2 * D + 32 * b) + (lane & 31); };
        if constexpr (GS_WG_PAIR && NB <= 8) {
            static_assert(NB % 2 == 0, "accumulator blocks are processed in pairs");
            constexpr int PP = NB / 2, NP = (WS / 16) * PP;        // pairs per k-step, pair steps per slab
            bf8 dh[WS / 16], dl[WS / 16];
#pragma unroll
            for (int ks = 0; ks < WS / 16; ++ks) {
                dh[ks] = *reinterpret_cast<const bf8*>(ah + 16 * ks);
                dl[ks] = *reinterpret_cast<const bf8*>(al + 16 * ks);
            }
            bf8 xh[2][2], xl[2][2];
            auto request = [&](int q, int buf) {
                const int ks = q / PP, b = 2 * (q % PP);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    xh[buf][i] = *reinterpret_cast<const bf8*>(hi_img + feat_of(b + i) * WB_RS + roff + 16 * ks);
                    xl[buf][i] = *reinterpret_cast<const bf8*>(lo_img + feat_of(b + i) * WB_RS + roff + 16 * ks);
                }
            };
            request(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (WS / 16) + 4, 0);         // the D fragments and the first pair
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int ks = q / PP, b = 2 * (q % PP), buf = q & 1;
                if (q + 1 < NP) request(q + 1, buf ^ 1);
                if (GS_WG_ABL & 1) continue;
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh[ks], xh[buf][0], acc[b], 0, 0, 0);
                acc[b + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh[ks], xh[buf][1], acc[b + 1], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh[ks], xl[buf][0], acc[b], 0, 0, 0);
                acc[b + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh[ks], xl[buf][1], acc[b + 1], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dl[ks], xh[buf][0], acc[b], 0, 0, 0);
                acc[b + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dl[ks], xh[buf][1], acc[b + 1], 0, 0, 0);
                if (q + 1 < NP) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);      // 4 LDS reads (the next pair)
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                       // 6 MFMAs (this pair)
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < WS / 16; ++ks) {
                const bf8 dh = *reinterpret_cast<const bf8*>(ah + 16 * ks), dl = *reinterpret_cast<const bf8*>(al + 16 * ks);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int feat = feat_of(b);
                    const bf8 xh = *reinterpret_cast<const bf8*>(hi_img + feat * WB_RS + roff + 16 * ks);
                    const bf8 xl = *reinterpret_cast<const bf8*>(lo_img + feat * WB_RS + roff + 16 * ks);
                    if (GS_WG_ABL & 1) continue;
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh, xh, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh, xl, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dl, xh, acc[b], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    };
    Staged R0, R1;
    load_slab(R0, slab0);
    if (GS_WG_DEPTH == 2 && nslab > 1) load_slab(R1, slab0 + stride);
    if (GS_WG_DEPTH == 2) {
        for (int64_t s = 0; s < nslab; s += 2) {
            reduce_slab(R0, s);
            if (s + 1 < nslab) reduce_slab(R1, s + 1);
        }
    } else {
        for (int64_t s = 0; s < nslab; ++s) reduce_slab(R0, s);
    }
    const int Kreal = (HAS_H ? D : 0) + (HAS_E ? W.E : 0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        int k = 32 * b + (lane & 31);
        bool ok = true;
        if (HAS_E && b >= (HAS_H ? 8 : 0)) {
            const int e = 32 * (b - (HAS_H ? 8 : 0)) + (lane & 31);
            ok = e < W.E;
            k = (HAS_H ? D : 0) + e;
        }
        if (ok)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                atomicAdd(W.dW[l] + (int64_t)n * Kreal + k, acc[b][r]);
            }
    }
    // bias gradient: the 8 lanes that share lane >> 3 hold the partial column sums of feature 32 wave + 8 c + (lane >> 3)
    float* red = reinterpret_cast<float*>(img);       // [256]
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = bsum[c];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        if ((lane & 7) == 0) red[fsub + 8 * c] = v;
    }
    __syncthreads();
    if (tid < D) atomicAdd(W.db[l] + tid, red[tid]);
}

// Tried and removed (round 4): the same reduction with its phases OVERLAPPED -- the D fragments loaded straight from the planes into registers
// (8 consecutive rows of one feature are 32 contiguous bytes), an X-only image (51 KB) double-buffered so that slab s + 1 is converted and
// written BETWEEN the MFMA groups of slab s, every register piece re-requested for slab s + 2 as soon as consumed, one barrier per slab, the
// loop body branch-free (a conditional around a load makes hipcc wait for ALL outstanding loads in front of every piece: 1.59 ms).  Gradients
// identical; 1.32 ms per iteration for both passes against 1.25 (the 160 accumulators of the skip layer leave no room for the staging
// registers: 22 spills), and 1.248 against 1.259 with the skip layer left on this kernel -- no gain.  Later timing-only variants:
// the MFMAs are hidden, the fragment-read + MFMA loop alone runs at 80 % of the matrix rate, and what is left is the lock-step latency chain
// store -> barrier -> reads + MFMAs -> barrier of ONE workgroup per CU; also measured without gain and removed: a 64-feature x half-K wave tile
// (a third less LDS read traffic).
__global__ void __launch_bounds__(NT, 2) k_h2_wgrad16(WgradArgs W) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem_b[];
    const int tid = threadIdx.x;
    int l = 0;
    for (int k = 1; k < W.n_layers; ++k)
        if ((int)blockIdx.x >= W.wg_first[k]) l = k;            // uniform: the table sits in scalar registers
    const int strip = (int)blockIdx.x - W.wg_first[l], nstrips = W.wg_first[l + 1] - W.wg_first[l];
    if (l == 0) wgrad16_layer<2, false, true>(W, l, smem_b, tid, strip, nstrips);
    else if (l == W.skip_layer) wgrad16_layer<10, true, true>(W, l, smem_b, tid, strip, nstrips);
    else wgrad16_layer<8, true, false>(W, l, smem_b, tid, strip, nstrips);
}

__global__ void __launch_bounds__(NT, 2) k_h2_wgrad(WgradArgs W) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int l = W.only_output ? W.n_layers : blockIdx.y, tid = threadIdx.x;
    if (l == W.n_layers) {
        // output layer: dw_out[n] = sum_rows g_out[row] a_{L-1}[row][n]  (tangent rows included: their g_out is dL/d(df/dx_d))
        const int64_t nslabs_total = wgrad_rows(W) / WS;
        const float* X = W.A + (int64_t)(W.n_layers - 1) * W.Rpad * D;
        // planes are [slab][feature][32 rows]: thread = (feature tid >> 1, rows 16 (tid & 1) .. + 15 of the slab), four 16-byte loads
        const int f = tid >> 1, rb = 16 * (tid & 1);
        float acc = 0.f;
        for (int64_t s = blockIdx.x; s < nslabs_total; s += gridDim.x) {
            const float* xp = X + s * (int64_t)(WS * D) + f * WS + rb;
            const float* gp = W.g_out + s * WS + rb;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 x = *reinterpret_cast<const float4*>(xp + 4 * i), g = *reinterpret_cast<const float4*>(gp + 4 * i);
                acc += g.x * x.x + g.y * x.y + g.z * x.z + g.w * x.w;
            }
        }
        acc += __shfl_xor(acc, 1, 64);
        if ((tid & 1) == 0 && acc != 0.f) atomicAdd(W.dW[l] + f, acc);
        return;
    }
    if (l == 0) wgrad_layer<2, false, true>(W, l, smem_f, tid);
    else if (l == W.skip_layer) wgrad_layer<10, true, true>(W, l, smem_f, tid);
    else wgrad_layer<8, true, false>(W, l, smem_f, tid);
}


// ---- device-side compaction of the rows that carry gradient ------------------------------------------------------------
// rows[] = ascending indices i with g[i] != 0, g_rows[] = their values, count on the DEVICE: the chain kernels read it, so the
// caller can size the planes by a bound (2 x crossing edges) instead of waiting for the exact count (a host sync costs
// ~1.5 ms of launch-ahead per iteration).  count[1] is set when the bound was too small (checked by the caller off the hot path).
constexpr int CZ_TILE = 1024;
__global__ void __launch_bounds__(256) k_cnz_count(const float* __restrict__ g, int64_t stride, int64_t N, int32_t* __restrict__ partial) {
    __shared__ int lds[4];
    const int64_t base = (int64_t)blockIdx.x * CZ_TILE + threadIdx.x * 4;
    int c = 0;
    for (int j = 0; j < 4; ++j) c += (base + j < N && g[(base + j) * stride] != 0.0f) ? 1 : 0;
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}
__global__ void __launch_bounds__(256) k_cnz_scan(int32_t* __restrict__ partial, int64_t nb, int64_t cap, int64_t* __restrict__ count) {
    // exclusive scan of the per-tile counts by one workgroup (thread t owns a contiguous run of tiles); the first version walked the
    // 2048 tiles of a 2 M-row mask with ONE thread: 0.2 ms of dependent loads
    __shared__ int64_t s_sum[256];
    const int tid = threadIdx.x;
    const int64_t per = (nb + 255) / 256, lo = tid * per, hi = min(nb, lo + per);
    int64_t sum = 0;
    for (int64_t b = lo; b < hi; ++b) sum += partial[b];
    s_sum[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int64_t run = 0;
        for (int t = 0; t < 256; ++t) { const int64_t v = s_sum[t]; s_sum[t] = run; run += v; }
        count[0] = min(run, cap);
        count[1] = run > cap ? 1 : 0;
    }
    __syncthreads();
    int64_t run = s_sum[tid];
    for (int64_t b = lo; b < hi; ++b) { const int32_t v = partial[b]; partial[b] = (int32_t)min(run, (int64_t)0x7fffffff); run += v; }
}
__global__ void __launch_bounds__(256) k_cnz_write(const float* __restrict__ g, int64_t stride, int64_t N, const int32_t* __restrict__ partial, int64_t cap,
                                                   int32_t* __restrict__ rows, float* __restrict__ g_rows) {
    __shared__ int lds[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * CZ_TILE + threadIdx.x * 4;
    float v[4];
    int c = 0;
    for (int j = 0; j < 4; ++j) { v[j] = base + j < N ? g[(base + j) * stride] : 0.0f; c += v[j] != 0.0f; }
    int inc = c;
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int64_t pos = partial[blockIdx.x] + inc - c;
    for (int w = 0; w < wave; ++w) pos += lds[w];
    for (int j = 0; j < 4; ++j)
        if (v[j] != 0.0f) {
            if (pos < cap) { rows[pos] = (int32_t)(base + j); if (g_rows) g_rows[pos] = v[j]; }
            ++pos;
        }
}

constexpr size_t SMEM_BYTES = (size_t)(2 * TM * LDH + 2 * TM * LDEH) * sizeof(_Float16);
constexpr size_t SMEM_H1_BYTES = (size_t)(TM * LDH + TM * LDEH) * sizeof(_Float16);
constexpr size_t SMEM_BWD_BYTES = (size_t)(2 * TM * LDH) * sizeof(_Float16) + (size_t)(TM * LDG + TM) * sizeof(float);
constexpr size_t SMEM_WGRAD_BYTES = (size_t)2 * WG_BUF * sizeof(float);
static_assert(SMEM_BYTES <= 80 * 1024 && SMEM_BWD_BYTES <= 80 * 1024, "two workgroups per CU");
static_assert(SMEM_WGRAD_BYTES <= 160 * 1024 && SMEM_WGRAD16_BYTES <= 160 * 1024, "LDS");

// ---- weight packing ------------------------------------------------------------------------------------------------
// packed = [ forward fragments | transposed (dgrad) fragments | fp32 tail: biases [n_layers][256], w_out [256], b_out ]
struct PackLayout {
    int n_layers, skip_layer, E;
    int64_t frag_off[MAX_LAYERS];     // h8 units, forward fragments of layer l: [k-step][8 blocks][2 pieces][64 lanes]
    int64_t fragT_off[MAX_LAYERS];    // transposed fragments of layer l: [n-step 16][nblkT][2][64]
    int nblkT[MAX_LAYERS];
    int64_t fragR_off[MAX_LAYERS];    // k_h1r_fwd's fragments of layer l: [feature block 8][k-step][64 lanes], ONE piece
    int64_t total_frags;              // forward + transposed + register-resident
    int64_t tail_off_bytes;
};

int64_t tail_floats(int n_layers) { return (((int64_t)n_layers * D + D + 1) + 3) & ~(int64_t)3; }       // fp32 tail, padded to 16 bytes
__host__ __device__ inline int64_t tail_r_off(const struct PackLayout& L);
int layer_steps(int l, int skip_layer) { return l == 0 ? EK / 16 : D / 16 + (l == skip_layer ? EK / 16 : 0); }

PackLayout make_layout(int n_freq, int n_hidden, int skip_layer) {
    PackLayout L{};
    L.n_layers = n_hidden + 1; L.skip_layer = skip_layer; L.E = 3 * (2 * n_freq + 1);
    int64_t off = 0;
    for (int l = 0; l < L.n_layers; ++l) {
        L.frag_off[l] = off;
        off += (int64_t)layer_steps(l, skip_layer) * 1024;
    }
    for (int l = 0; l < L.n_layers; ++l) {
        L.nblkT[l] = l == 0 ? 2 : (l == skip_layer ? 10 : 8);
        L.fragT_off[l] = off;
        off += (int64_t)(D / 16) * L.nblkT[l] * 128;
    }
    for (int l = 0; l < L.n_layers; ++l) {
        L.fragR_off[l] = off;
        off += (int64_t)8 * layer_steps(l, skip_layer) * 64;
    }
    L.total_frags = off;
    L.tail_off_bytes = off * 16;
    return L;
}

__host__ __device__ inline int64_t tail_r_off(const PackLayout& L) { return ((((int64_t)L.n_layers * D + D + 1) + 3) & ~(int64_t)3); }   // floats from the tail's start

struct PackArgs {
    const float* w[MAX_LAYERS + 1];   // torch Linear.weight [out, in] of every layer, output layer last
    const float* b[MAX_LAYERS + 1];
    PackLayout L;
    h8* frags;
    float* tail;
    uint32_t* status;                 // optional: ST_NONFINITE is raised for a weight beyond the fp16 range (split_h2 would clamp it)
    int with_r;                       // also write k_h1r_fwd's fragment section and scaled tail
};

__global__ void __launch_bounds__(256) k_h2_pack(PackArgs P) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const PackLayout& L = P.L;
    if (i < L.total_frags) {
        h8 o;
        if (i < L.fragT_off[0]) {      // forward: rows = output features n, K = input index k
            int l = 0;
            while (l + 1 < L.n_layers && i >= L.frag_off[l + 1]) ++l;
            const int64_t j = i - L.frag_off[l];
            const int lane = (int)(j & 63), piece = (int)((j >> 6) & 1), wave = (int)((j >> 7) & 7), step = (int)(j >> 10);
            const int n = wave * 32 + (lane & 31);
            const int k0 = step * 16 + 8 * (lane >> 5);
            const int Kin = l == 0 ? L.E : (l == L.skip_layer ? D + L.E : D);      // torch row length
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                int k = k0 + q, src = -1;
                if (l == 0) src = k < L.E ? k : -1;
                else if (k < D) src = k;
                else if (l == L.skip_layer && k - D < L.E) src = k;            // [h | emb] order of geometry/mlp.py:37
                float v = src >= 0 ? P.w[l][(int64_t)n * Kin + src] : 0.f;
                if (P.status && !(fabsf(v) <= 60000.0f)) atomicOr(&P.status[ST_NONFINITE], 1u);
                _Float16 hi, lo;
                split_h2(v, hi, lo);
                o[q] = piece ? lo : hi;
            }
        } else if (i >= L.fragR_off[0]) {      // k_h1r_fwd: rows = output features, K-slot (step, half, element) -> input index in ITS order
            if (!P.with_r) return;
            int l = 0;
            while (l + 1 < L.n_layers && i >= L.fragR_off[l + 1]) ++l;
            const int64_t j = i - L.fragR_off[l];
            const int nsteps = l == 0 ? EK / 16 : D / 16 + (l == L.skip_layer ? EK / 16 : 0);
            const int lane = (int)(j & 63), step = (int)((j >> 6) % nsteps), mb = (int)((j >> 6) / nsteps);
            const int n = mb * 32 + (lane & 31), hh = lane >> 5;
            const int Kin = l == 0 ? L.E : (l == L.skip_layer ? D + L.E : D);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                int src;
                if (l == 0) { const int e = 16 * step + 8 * hh + q; src = e < L.E ? e : -1; }
                else if (step < D / 16) src = 32 * (step >> 1) + 4 * hh + 8 * (2 * (step & 1) + (q >> 2)) + (q & 3);      // the accumulator layout of the layer below
                else { const int e = 16 * (step - D / 16) + 8 * hh + q; src = e < L.E ? D + e : -1; }
                // weights that multiply the (unscaled) encoding carry the kernel's activation scale c; those fed by scaled activations do not
                const bool enc_in = l == 0 || step >= D / 16;
                const float v = src >= 0 ? P.w[l][(int64_t)n * Kin + src] * (enc_in ? SP_C1 : 1.0f) : 0.f;
                if (P.status && !(fabsf(v) <= 60000.0f)) atomicOr(&P.status[ST_NONFINITE], 1u);
                _Float16 hi, lo;
                split_h2(v, hi, lo);
                o[q] = hi;
            }
        } else {                       // transposed: rows = INPUT index of the layer (h feature, or encoding entry), K = output feature n
            int l = 0;
            while (l + 1 < L.n_layers && i >= L.fragT_off[l + 1]) ++l;
            const int64_t j = i - L.fragT_off[l];
            const int nb = L.nblkT[l];
            const int lane = (int)(j & 63), piece = (int)((j >> 6) & 1);
            const int blk = (int)((j >> 7) % nb), step = (int)((j >> 7) / nb);
            const int Kin = l == 0 ? L.E : (l == L.skip_layer ? D + L.E : D);
            int col;                  // torch column of this fragment row, -1 = padding
            if (l == 0) { int e = blk * 32 + (lane & 31); col = e < L.E ? e : -1; }
            else if (blk < 8) col = blk * 32 + (lane & 31);
            else { int e = (blk - 8) * 32 + (lane & 31); col = e < L.E ? D + e : -1; }
            const int n0 = step * 16 + 8 * (lane >> 5);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float v = col >= 0 ? P.w[l][(int64_t)(n0 + q) * Kin + col] : 0.f;
                _Float16 hi, lo;
                split_h2(v, hi, lo);
                o[q] = piece ? lo : hi;
            }
        }
        P.frags[i] = o;
    }
    const int n_tail = L.n_layers * D + D + 1;
    if (i < n_tail) {
        float v;
        if (i < (int64_t)L.n_layers * D) v = P.b[i / D][i % D];
        else if (i < (int64_t)L.n_layers * D + D) v = P.w[L.n_layers][i - (int64_t)L.n_layers * D];
        else v = P.b[L.n_layers][0];
        P.tail[i] = v;
        if (P.with_r && i < (int64_t)L.n_layers * D + D) P.tail[tail_r_off(L) + i] = i < (int64_t)L.n_layers * D ? v * SP_C1 : v * (1.0f / SP_C1);
    }
}

int check_shape(int n_freq, int n_hidden, int skip_layer) {
    int E = 3 * (2 * n_freq + 1);
    GS_REQUIRE(n_freq >= 0 && E <= EK, "sdf_mlp_h2: positional encoding wider than 48 is not supported");
    GS_REQUIRE(n_hidden >= 0 && n_hidden + 1 <= MAX_LAYERS, "sdf_mlp_h2: too many layers");
    GS_REQUIRE(skip_layer == -1 || (skip_layer >= 1 && skip_layer <= n_hidden), "sdf_mlp_h2: bad skip layer");
    return 0;
}

void fill_fwd_args(H2Args& A, const void* packed, const PackLayout& L) {
    A.n_layers = L.n_layers; A.skip_layer = L.skip_layer; A.E = L.E;
    const float* tail = (const float*)((const char*)packed + L.tail_off_bytes);
    for (int l = 0; l < L.n_layers; ++l) {
        A.wfrag[l] = (const h8*)packed + L.frag_off[l];
        A.wfragR[l] = (const h8*)packed + L.fragR_off[l];
        A.bias[l] = tail + (int64_t)l * D;
    }
    A.w_out = tail + (int64_t)L.n_layers * D;
    A.tailR = tail + tail_r_off(L);
}

#ifndef GS_H2_DUAL
#define GS_H2_DUAL 0    // 1: two tiles per 1024-thread workgroup, deterministically anti-phased (measured: same time)
#endif

template <int MODE>
int launch_fwd(const H2Args& A, int64_t tiles, hipStream_t stream) {
    // per launch (cheap, and correct per device / per thread, unlike a process-wide "done" flag)
    if (GS_H2_DUAL) {
        GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_h2_fwd<MODE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * SMEM_BYTES)));
        hipLaunchKernelGGL((k_h2_fwd<MODE, true>), dim3((unsigned)gs::cdiv(tiles, 2)), dim3(2 * NT), 2 * SMEM_BYTES, stream, A);
    } else {
        GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_h2_fwd<MODE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_BYTES));
        hipLaunchKernelGGL((k_h2_fwd<MODE, false>), dim3((unsigned)tiles), dim3(NT), SMEM_BYTES, stream, A);
    }
    GS_LAUNCH_CHECK();
    return 0;
}


// Eikonal term on the <EIK> planes: virtual row 64 t + 16 c + j holds f (c = 0) and df/dx_{c-1} (c = 1..3) of sample 16 t + j.
// loss += sum_i (|J_i| - 1)^2;  g_unit = d loss / d (virtual row outputs) -- scaled by the upstream scalar in the backward pass.
__global__ void __launch_bounds__(256) k_eikonal_loss(const float* __restrict__ out, int64_t n, int64_t tiles, float* __restrict__ loss,
                                                      float* __restrict__ g_unit) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    if (s < tiles * 16) {
        const int64_t base = (s >> 4) * 64 + (s & 15);
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (s < n) {
            const float jx = out[base + 16], jy = out[base + 32], jz = out[base + 48];
            const float nrm = sqrtf(jx * jx + jy * jy + jz * jz);
            acc = (nrm - 1.0f) * (nrm - 1.0f);
            const float k = 2.0f * (nrm - 1.0f) / fmaxf(nrm, 1e-20f);
            gx = k * jx; gy = k * jy; gz = k * jz;
        }
        g_unit[base] = 0.f;
        g_unit[base + 16] = gx;
        g_unit[base + 32] = gy;
        g_unit[base + 48] = gz;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ float sw[4];
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (sw[0] + sw[1]) + (sw[2] + sw[3]);
        if (t != 0.f) atomicAdd(loss, t);
    }
}

// RR set-up: g_all [2 Rpad] = the per-row upstream gradients of the two launches that read one -- rows [Rpad, Rpad + n) = 1 (the reverse chain
// with g_out = 1 reads g_all + Rpad; the output layer's weight gradient sum_rows g_all[row] x[row] then picks the u rows), 0 elsewhere
__global__ void __launch_bounds__(256) k_rr_init(float* __restrict__ g_all, int64_t n, int64_t Rpad, float* __restrict__ loss) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < 2 * Rpad) g_all[i] = (i >= Rpad && i - Rpad < n) ? 1.0f : 0.0f;
    if (i == 0) loss[0] = 0.f;
}
// loss += sum_i (|g_i| - 1)^2,  gbar_i = d loss / d g_i  (the clamp of k_eikonal_loss: a zero gradient has a zero subgradient)
__global__ void __launch_bounds__(256) k_rr_loss(const float* __restrict__ gx, int64_t n, float* __restrict__ loss, float* __restrict__ gbar) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    if (i < n) {
        const float jx = gx[3 * i], jy = gx[3 * i + 1], jz = gx[3 * i + 2];
        const float nrm = sqrtf(jx * jx + jy * jy + jz * jz);
        acc = (nrm - 1.0f) * (nrm - 1.0f);
        const float k = 2.0f * (nrm - 1.0f) / fmaxf(nrm, 1e-20f);
        gbar[3 * i] = k * jx; gbar[3 * i + 1] = k * jy; gbar[3 * i + 2] = k * jz;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ float sw[4];
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (sw[0] + sw[1]) + (sw[2] + sw[3]);
        if (t != 0.f) atomicAdd(loss, t);
    }
}

}  // namespace

static int g_h1_impl = 0;      // see gs_sdf_mlp_h1_impl below

extern "C" int64_t gs_sdf_mlp_h2_packed_bytes(int n_freq, int n_hidden, int skip_layer) {
    PackLayout L = make_layout(n_freq, n_hidden, skip_layer);
    // + one staging block: k_h1r_fwd stages every weight chunk as R1_CHUNK fragments, the last chunk's unread excess runs over the tail
    return L.tail_off_bytes + (tail_r_off(L) + (int64_t)L.n_layers * D + D + 4) * 4 + (int64_t)R1_CHUNK * 16;
}

// weights / biases: HOST arrays of n_hidden + 2 DEVICE pointers (torch layout, Linear.weight [out, in] row-major; the
// output layer last).  packed: gs_sdf_mlp_h2_packed_bytes bytes, 16-byte aligned, WRITTEN.
extern "C" int gs_sdf_mlp_h2_pack(const float* const* weights, const float* const* biases, int n_freq, int n_hidden, int skip_layer, void* packed,
                                  uint32_t* status, gs_stream_t stream) {
    GS_REQUIRE(weights && biases && packed, "gs_sdf_mlp_h2_pack: null pointer");
    GS_REQUIRE(((uintptr_t)packed & 15) == 0, "gs_sdf_mlp_h2_pack: packed buffer must be 16-byte aligned");
    if (int rc = check_shape(n_freq, n_hidden, skip_layer)) return rc;
    PackArgs P{};
    P.L = make_layout(n_freq, n_hidden, skip_layer);
    for (int l = 0; l <= P.L.n_layers; ++l) {
        GS_REQUIRE(weights[l] && biases[l], "gs_sdf_mlp_h2_pack: null layer pointer");
        P.w[l] = weights[l];
        P.b[l] = biases[l];
    }
    P.frags = (h8*)packed;
    P.tail = (float*)((char*)packed + P.L.tail_off_bytes);
    P.status = status;
    P.with_r = g_h1_impl == 1;
    hipLaunchKernelGGL(k_h2_pack, dim3((unsigned)gs::cdiv(P.L.total_frags, 256)), dim3(256), 0, (hipStream_t)stream, P);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_sdf_mlp_fwd_h2(const float* x, int64_t N, const void* packed, int n_freq, int n_hidden, int skip_layer, float* out,
                                 uint64_t* occ_bits, uint32_t* status, gs_stream_t stream) {
    if (N == 0) return 0;
    GS_REQUIRE(x && packed && out, "gs_sdf_mlp_fwd_h2: null pointer");
    if (int rc = check_shape(n_freq, n_hidden, skip_layer)) return rc;
    H2Args A{};
    A.x = x; A.out = out; A.occ = occ_bits; A.status = status; A.N = N; A.n_freq = n_freq;
    fill_fwd_args(A, packed, make_layout(n_freq, n_hidden, skip_layer));
    return launch_fwd<MODE_GRID>(A, gs::cdiv(N, TM), (hipStream_t)stream);
}

// which kernel gs_sdf_mlp_fwd_h1 launches: 0 = k_h1_fwd (activations in LDS, weights through L1; default), 1 = k_h1r_fwd (activations in registers,
// weights through LDS: correct, but SLOWER on MI355X -- 5.6 ms against 2.35: a wave's VALU instructions are not hidden behind its OWN MFMAs, see
// tools/micro/mfma_fillers.hip and DESIGN.md 7.2; kept selectable as the measured record of that design).  Same arithmetic class, different K
// order: values agree to fp32 summation order.  impl < 0: query.  Returns the old value.  gs_sdf_mlp_h2_pack writes k_h1r_fwd's fragment section
// only while impl == 1 is selected.
extern "C" int gs_sdf_mlp_h1_impl(int impl) {
    const int old = g_h1_impl;
#if GS_ORACLE_KERNELS
    if (impl == 0 || impl == 1) g_h1_impl = impl;
#else
    if (impl == 1) {          // the shipped library holds ONE forward design
        gs::set_error("gs_sdf_mlp_h1_impl(1): k_h1r_fwd is an alternate-design kernel, not in the shipped library -- use gshell_amd/lib/variants/oracles.so");
        return -1;
    }
#endif
    return old;
}

// First pass of the two-pass forward: one fp16 product per algorithmic product.  Same packed image (its own fragment section for k_h1r_fwd).
extern "C" int gs_sdf_mlp_fwd_h1(const float* x, int64_t N, const void* packed, int n_freq, int n_hidden, int skip_layer, float* out,
                                 uint64_t* occ_bits, uint32_t* status, gs_stream_t stream) {
    if (N == 0) return 0;
    GS_REQUIRE(x && packed && out, "gs_sdf_mlp_fwd_h1: null pointer");
    if (int rc = check_shape(n_freq, n_hidden, skip_layer)) return rc;
    H2Args A{};
    A.x = x; A.out = out; A.occ = occ_bits; A.status = status; A.N = N; A.n_freq = n_freq;
    fill_fwd_args(A, packed, make_layout(n_freq, n_hidden, skip_layer));
#if GS_ORACLE_KERNELS
    if (g_h1_impl == 1) {          // register-resident activations (round 5)
        GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_h1r_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_H1R_BYTES));
        hipLaunchKernelGGL(k_h1r_fwd, dim3((unsigned)gs::cdiv(N, 32 * R1_NW)), dim3(R1_NT), SMEM_H1R_BYTES, (hipStream_t)stream, A);
        GS_LAUNCH_CHECK();
        return 0;
    }
#endif
    constexpr size_t smem = smem_h1_bytes<GS_H1_RM>();
    GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_h1_fwd<GS_H1_NW, GS_H1_RM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL((k_h1_fwd<GS_H1_NW, GS_H1_RM>), dim3((unsigned)gs::cdiv(N, 32 * GS_H1_RM)), dim3(64 * GS_H1_NW), smem, (hipStream_t)stream, A);
    GS_LAUNCH_CHECK();
    return 0;
}

// Second pass: rows[0 .. min(*count_dev, cap)) of x are recomputed with the three-product arithmetic of gs_sdf_mlp_fwd_h2 (bit
// identical values: a row's arithmetic does not depend on its tile), written over out[rows[r]]; the rows' sign bits in occ_bits
// are corrected; status[ST_MAXDEV] = bits of max |new - old| over the rows (atomicMax: zero it first); tau > 0: status has a third
// word, status[ST_MAXREL] = bits of max |new - old| / max(tau, |new|) -- the fraction of its sign margin a row's first-pass error used up.
extern "C" int gs_sdf_mlp_h2_refine_rows(const float* x, const int32_t* rows, int64_t cap, const int64_t* count_dev, const void* packed, int n_freq,
                                         int n_hidden, int skip_layer, float* out, uint64_t* occ_bits, uint32_t* status, float tau, gs_stream_t stream) {
    if (cap == 0) return 0;
    GS_REQUIRE(x && rows && packed && out, "gs_sdf_mlp_h2_refine_rows: null pointer");
    if (int rc = check_shape(n_freq, n_hidden, skip_layer)) return rc;
    H2Args A{};
    A.x = x; A.rows = rows; A.out = out; A.occ = occ_bits; A.status = status; A.N = cap; A.n_dev = count_dev; A.n_freq = n_freq; A.tau = tau;
    fill_fwd_args(A, packed, make_layout(n_freq, n_hidden, skip_layer));
    return launch_fwd<MODE_FIX>(A, gs::cdiv(cap, TM), (hipStream_t)stream);
}

extern "C" int64_t gs_sdf_mlp_h2_rows_padded(int mode, int64_t n) {      // virtual rows of the saved planes: whole PAIRS of 64-row tiles
    return mode == MODE_EIK ? gs::cdiv(n, 32) * 2 * TM : gs::cdiv(n, 2 * TM) * 2 * TM;
}

// mode 1 (ROWS): recompute rows `rows[0..n)` of x;  mode 2 (EIK): value + 3 tangent rows of the n sample points x [n,3].
// A [n_hidden+1][Rpad][256], EMB [Rpad][48] WRITTEN (Rpad = gs_sdf_mlp_h2_rows_padded);  out [Rpad] WRITTEN in mode 2
// (tile-major virtual rows: entry 64 t + 16 c + i belongs to sample 16 t + i; c = 0: f - b_out, c = 1..3: df/dx_{c-1}).
extern "C" int64_t gs_compact_rows_scratch_bytes(int64_t N) { return gs::cdiv(N, CZ_TILE) * (int64_t)sizeof(int32_t) + 16; }

// rows [cap] i32 = ascending indices with g[i] != 0, g_rows [>= cap] f32 = those values (the caller zero-fills the tail),
// count_dev [2] i64 = (min(count, cap), count > cap) -- all on the device, no synchronisation.
extern "C" int gs_compact_rows_strided(const float* g, int64_t stride, int64_t N, int64_t cap, void* scratch, int32_t* rows, float* g_rows,
                                       int64_t* count_dev, gs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GS_REQUIRE(g && scratch && rows && count_dev && cap >= 0 && stride >= 1, "gs_compact_rows: null pointer");      // g_rows may be NULL (indices only)
    const int64_t nb = gs::cdiv(N, CZ_TILE);
    if (nb == 0) { GS_HIP_CHECK(hipMemsetAsync(count_dev, 0, 16, stream)); return 0; }
    hipLaunchKernelGGL(k_cnz_count, dim3((unsigned)nb), dim3(256), 0, stream, g, stride, N, (int32_t*)scratch);
    hipLaunchKernelGGL(k_cnz_scan, dim3(1), dim3(256), 0, stream, (int32_t*)scratch, nb, cap, count_dev);
    hipLaunchKernelGGL(k_cnz_write, dim3((unsigned)nb), dim3(256), 0, stream, g, stride, N, (const int32_t*)scratch, cap, rows, g_rows);
    GS_LAUNCH_CHECK();
    return 0;
}
extern "C" int gs_compact_rows(const float* g, int64_t N, int64_t cap, void* scratch, int32_t* rows, float* g_rows, int64_t* count_dev,
                               gs_stream_t stream) {
    return gs_compact_rows_strided(g, 1, N, cap, scratch, rows, g_rows, count_dev, stream);
}

extern "C" int gs_sdf_mlp_h2_save_fwd(int mode, const float* x, const int32_t* rows, int64_t n, const int64_t* n_dev, const void* packed, int n_freq,
                                      int n_hidden, int skip_layer, float* A_save, float* EMB_save, float* out, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_REQUIRE(mode == MODE_ROWS || mode == MODE_EIK, "gs_sdf_mlp_h2_save_fwd: mode must be 1 (rows) or 2 (eikonal)");
    GS_REQUIRE(x && packed && A_save && EMB_save && (mode == MODE_EIK ? out != nullptr : rows != nullptr), "gs_sdf_mlp_h2_save_fwd: null pointer");
    if (int rc = check_shape(n_freq, n_hidden, skip_layer)) return rc;
    H2Args A{};
    A.x = x; A.rows = rows; A.out = out; A.N = n; A.n_dev = mode == MODE_ROWS ? n_dev : nullptr; A.n_freq = n_freq; A.A = A_save; A.EMB = EMB_save;
    A.Rpad = gs_sdf_mlp_h2_rows_padded(mode, n);
    fill_fwd_args(A, packed, make_layout(n_freq, n_hidden, skip_layer));
#if GS_ORACLE_KERNELS
    return mode == MODE_ROWS ? launch_fwd<MODE_ROWS>(A, A.Rpad / TM, (hipStream_t)stream) : launch_fwd<MODE_EIK>(A, A.Rpad / TM, (hipStream_t)stream);
#else
    if (mode != MODE_ROWS) GS_ORACLE_ONLY("gs_sdf_mlp_h2_save_fwd(mode 2: tangent-row eikonal planes)");      // the shipped eikonal term is gs_sdf_eikonal_rr_*
    return launch_fwd<MODE_ROWS>(A, A.Rpad / TM, (hipStream_t)stream);
#endif
}

// Backward chain over the saved planes: D [n_hidden+1][Rpad][256] WRITTEN (dL/d pre-activation of every layer, per virtual
// row); mode 1 with g_x != NULL: g_x[rows[r]] (of [N,3]) WRITTEN for r < n (dL/dx through the encoding).  g_out [Rpad]:
// upstream gradient per virtual row, 0 on padding rows.
extern "C" int gs_sdf_mlp_h2_bwd(int mode, const float* g_out, const int32_t* rows, int64_t n, const int64_t* n_dev, const void* packed, int n_freq,
                                 int n_hidden, int skip_layer, const float* A_save, const float* EMB_save, float* D_save, float* g_x,
                                 gs_stream_t stream) {
    if (n == 0) return 0;
    GS_REQUIRE(mode == MODE_ROWS || mode == MODE_EIK, "gs_sdf_mlp_h2_bwd: mode must be 1 (rows) or 2 (eikonal)");
    GS_REQUIRE(g_out && packed && A_save && EMB_save && D_save && (mode == MODE_EIK || rows != nullptr), "gs_sdf_mlp_h2_bwd: null pointer");
    if (int rc = check_shape(n_freq, n_hidden, skip_layer)) return rc;
    PackLayout L = make_layout(n_freq, n_hidden, skip_layer);
    BwdArgs B{};
    B.g_out = g_out; B.A = A_save; B.EMB = EMB_save; B.Dsave = D_save; B.rows = rows; B.g_x = mode == MODE_ROWS ? g_x : nullptr;
    B.R = n; B.n_dev = mode == MODE_ROWS ? n_dev : nullptr; B.Rpad = gs_sdf_mlp_h2_rows_padded(mode, n); B.E = L.E; B.n_layers = L.n_layers; B.skip_layer = L.skip_layer;
    for (int l = 0; l < L.n_layers; ++l) {
        B.wfragT[l] = (const h8*)packed + L.fragT_off[l];
        B.nblkT[l] = L.nblkT[l];
    }
    B.w_out = (const float*)((const char*)packed + L.tail_off_bytes) + (int64_t)L.n_layers * D;
    hipStream_t st = (hipStream_t)stream;
    if (mode == MODE_ROWS) {
        GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_h2_bwd<MODE_ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_BWD_BYTES));
        hipLaunchKernelGGL(k_h2_bwd<MODE_ROWS>, dim3((unsigned)(B.Rpad / TM)), dim3(NT), SMEM_BWD_BYTES, st, B);
    } else {
#if GS_ORACLE_KERNELS
        GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_h2_bwd<MODE_EIK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_BWD_BYTES));
        hipLaunchKernelGGL(k_h2_bwd<MODE_EIK>, dim3((unsigned)(B.Rpad / TM)), dim3(NT), SMEM_BWD_BYTES, st, B);
#else
        GS_ORACLE_ONLY("gs_sdf_mlp_h2_bwd(mode 2: tangent-row eikonal planes)");
#endif
    }
    GS_LAUNCH_CHECK();
    return 0;
}

// Weight / bias gradients from the saved planes, ACCUMULATED (float atomics) into torch-layout tensors:
// dW, db = HOST arrays of n_hidden + 2 DEVICE pointers (Linear.weight.grad [out,in], Linear.bias.grad), output layer last;
// db[n_hidden + 1] (the output bias, = sum of g_out over the value rows) is not touched.
static int wgrad_launch(int mode, const float* g_out, int64_t Rpad, int64_t n_rows, const int64_t* n_dev, int n_freq, int n_hidden, int skip_layer,
                        const float* A_save, const float* EMB_save, const float* D_save, float* const* dW, float* const* db, int exact_fp32,
                        hipStream_t stream) {
    WgradArgs W{};
    W.A = A_save; W.EMB = EMB_save; W.D = D_save; W.g_out = g_out; W.Rpad = Rpad;
    W.n_dev = n_dev;
    W.n = n_rows;
    W.E = 3 * (2 * n_freq + 1); W.n_layers = n_hidden + 1; W.skip_layer = skip_layer; W.mode = mode;
    for (int l = 0; l <= W.n_layers; ++l) {
        GS_REQUIRE(dW[l] && (l == W.n_layers || db[l]), "sdf_mlp_h2 weight gradients: null gradient pointer");
        W.dW[l] = dW[l];
        W.db[l] = db[l];
    }
    const int64_t nslabs = W.Rpad / WS;
#ifndef GS_WG_STRIPS
#define GS_WG_STRIPS 80      // row strips per layer (x 7 layers = workgroups, one per CU at a time: 92 KB of LDS).  Measured per iteration (both passes):
                            // 36: 1.33, 40: 1.47, 60: 1.26, 80: 1.23, 120: 1.28 ms -- the strips' final atomics are free (no change when removed)
#endif
    int strips = (int)std::min<int64_t>(nslabs, GS_WG_STRIPS);
    W.slabs_per_strip = (int)gs::cdiv(nslabs, strips);
    strips = (int)gs::cdiv(nslabs, W.slabs_per_strip);
    GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_h2_wgrad), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_WGRAD_BYTES));
    if (exact_fp32) {
        hipLaunchKernelGGL(k_h2_wgrad, dim3((unsigned)strips, (unsigned)(W.n_layers + 1)), dim3(NT), SMEM_WGRAD_BYTES, stream, W);
    } else {
        // ONE round of workgroups (the 92 KB image holds a CU), dealt to the layers in proportion to their work: every workgroup ends with 256 x K float atomics on its layer's gradient, and with 80
        // strips for every layer those 45 M atomics were HALF of the kernel's time (timing-only variants, round 4: 0.28 of 0.44 ms were left
        // with the loads, the staging and the MFMAs all removed)
        int n_wg = 256;
        {
            int dev = 0, cus = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) n_wg = cus;
        }
#ifdef GS_WG_ROUNDS
        n_wg *= GS_WG_ROUNDS;
#endif
        n_wg = (int)std::max<int64_t>(W.n_layers, std::min<int64_t>(n_wg, nslabs * W.n_layers));
#ifndef GS_WG_W0
#define GS_WG_W0 0.6
#endif
        // a layer's cost per slab follows the BYTES it stages (D + X: 38 / 64 / 70 KB for the first / a hidden / the skip layer), not its MFMAs
        auto weight = [&](int l) { return l == 0 ? (double)GS_WG_W0 : (l == W.skip_layer ? 1.1 : 1.0); };
        double wsum = 0.0;
        for (int l = 0; l < W.n_layers; ++l) wsum += weight(l);
        W.wg_first[0] = 0;
        double run = 0.0;
        for (int l = 0; l < W.n_layers; ++l) {
            run += weight(l);
            const int end = l + 1 == W.n_layers ? n_wg : (int)std::lround(run / wsum * n_wg);
            W.wg_first[l + 1] = std::max(end, W.wg_first[l] + 1);            // at least one workgroup per layer
        }
        GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_h2_wgrad16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_WGRAD16_BYTES));
        hipLaunchKernelGGL(k_h2_wgrad16, dim3((unsigned)W.wg_first[W.n_layers]), dim3(NT), SMEM_WGRAD16_BYTES, stream, W);
        W.only_output = 1;
        hipLaunchKernelGGL(k_h2_wgrad, dim3((unsigned)strips, 1), dim3(NT), SMEM_WGRAD_BYTES, stream, W);
    }
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_sdf_mlp_h2_wgrad(int mode, const float* g_out, int64_t n, const int64_t* n_dev, int n_freq, int n_hidden, int skip_layer,
                                   const float* A_save, const float* EMB_save, const float* D_save, float* const* dW, float* const* db,
                                   int exact_fp32, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_REQUIRE(mode == MODE_ROWS || mode == MODE_EIK, "gs_sdf_mlp_h2_wgrad: mode must be 1 (rows) or 2 (eikonal)");
    GS_REQUIRE(g_out && A_save && EMB_save && D_save && dW && db, "gs_sdf_mlp_h2_wgrad: null pointer");
    if (int rc = check_shape(n_freq, n_hidden, skip_layer)) return rc;
    const int64_t Rpad = gs_sdf_mlp_h2_rows_padded(mode, n);
    return wgrad_launch(mode, g_out, Rpad, mode == MODE_ROWS ? n : Rpad, mode == MODE_ROWS ? n_dev : nullptr, n_freq, n_hidden, skip_layer, A_save, EMB_save, D_save,
                        dW, db, exact_fp32, (hipStream_t)stream);
}

extern "C" int gs_sdf_eikonal_loss(const float* out_rows, int64_t n, int64_t rows_padded, float* loss, float* g_unit, gs_stream_t stream) {
    GS_REQUIRE(loss && (rows_padded == 0 || (out_rows && g_unit)), "gs_sdf_eikonal_loss: null pointer");
    GS_REQUIRE(rows_padded % 64 == 0 && 4 * n <= rows_padded, "gs_sdf_eikonal_loss: rows_padded must be gs_sdf_mlp_h2_rows_padded(2, n)");
    GS_HIP_CHECK(hipMemsetAsync(loss, 0, sizeof(float), (hipStream_t)stream));
    const int64_t tiles = rows_padded / 64;
    if (tiles == 0) return 0;
    hipLaunchKernelGGL(k_eikonal_loss, dim3((unsigned)gs::cdiv(tiles * 16, 256)), dim3(256), 0, (hipStream_t)stream, out_rows, n, tiles, loss, g_unit);
    GS_LAUNCH_CHECK();
    return 0;
}

// ---- eikonal term, reverse over reverse (MODE_RR above) ----------------------------------------------------------------------------
// Buffers (caller-owned, alive from _fwd to _bwd): with Rpad = gs_sdf_eikonal_rr_rows_padded(n) and L = n_hidden + 1 layers
//   A_all [L][2 Rpad][256], D_all [L][2 Rpad][256] (plane layout), EMB_all [2 Rpad][48], g_all [2 Rpad], gbar [n][3], grad_f [n][3].
extern "C" int64_t gs_sdf_eikonal_rr_rows_padded(int64_t n) { return gs::cdiv(n, TM) * TM; }      // every 64-row tile holds a sample: no tile is skipped

// loss[0] = sum_i (|grad_x f(x_i)| - 1)^2 (WRITTEN), grad_f [n][3] = grad_x f (WRITTEN), gbar = d loss / d grad_f (WRITTEN); first halves of A_all / EMB_all
// (a_l, e) and second half of D_all (delta_l) WRITTEN.
extern "C" int gs_sdf_eikonal_rr_fwd(const float* x, int64_t n, const void* packed, int n_freq, int n_hidden, int skip_layer, float* A_all, float* EMB_all,
                                     float* D_all, float* g_all, float* grad_f, float* gbar, float* loss, gs_stream_t stream) {
    GS_REQUIRE(loss, "gs_sdf_eikonal_rr_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { GS_HIP_CHECK(hipMemsetAsync(loss, 0, sizeof(float), st)); return 0; }
    GS_REQUIRE(x && packed && A_all && EMB_all && D_all && g_all && grad_f && gbar, "gs_sdf_eikonal_rr_fwd: null pointer");
    if (int rc = check_shape(n_freq, n_hidden, skip_layer)) return rc;
    const PackLayout L = make_layout(n_freq, n_hidden, skip_layer);
    const int64_t Rpad = gs_sdf_eikonal_rr_rows_padded(n);
    hipLaunchKernelGGL(k_rr_init, dim3((unsigned)gs::cdiv(2 * Rpad, 256)), dim3(256), 0, st, g_all, n, Rpad, loss);
    H2Args A{};
    A.x = x; A.N = n; A.n_freq = n_freq; A.A = A_all; A.EMB = EMB_all; A.Rpad = Rpad; A.Pstride = 2 * Rpad;
    fill_fwd_args(A, packed, L);
    if (int rc = launch_fwd<MODE_ROWS>(A, Rpad / TM, st)) return rc;
    BwdArgs B{};
    B.g_out = g_all + Rpad; B.A = A_all; B.EMB = EMB_all; B.Dsave = D_all + Rpad * D; B.g_x = grad_f;
    B.R = n; B.Rpad = Rpad; B.Pstride = 2 * Rpad; B.E = L.E; B.n_layers = L.n_layers; B.skip_layer = L.skip_layer;
    for (int l = 0; l < L.n_layers; ++l) {
        B.wfragT[l] = (const h8*)packed + L.fragT_off[l];
        B.nblkT[l] = L.nblkT[l];
    }
    B.w_out = (const float*)((const char*)packed + L.tail_off_bytes) + (int64_t)L.n_layers * D;
    GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_h2_bwd<MODE_ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_BWD_BYTES));
    hipLaunchKernelGGL(k_h2_bwd<MODE_ROWS>, dim3((unsigned)(Rpad / TM)), dim3(NT), SMEM_BWD_BYTES, st, B);
    hipLaunchKernelGGL(k_rr_loss, dim3((unsigned)gs::cdiv(n, 256)), dim3(256), 0, st, (const float*)grad_f, n, loss, gbar);
    GS_LAUNCH_CHECK();
    return 0;
}

// Parameter gradients of g_up * loss, ACCUMULATED into dW / db (as gs_sdf_mlp_h2_wgrad; the output bias gets none): g_up = device scalar.
// Second halves of A_all / EMB_all (u_l, J gbar) and the first half of D_all (S_l, then zbar_l) WRITTEN.
extern "C" int gs_sdf_eikonal_rr_bwd(int64_t n, const void* packed, int n_freq, int n_hidden, int skip_layer, float* A_all, float* EMB_all, float* D_all,
                                     const float* g_all, const float* gbar, const float* g_up, float* const* dW, float* const* db, int exact_fp32,
                                     gs_stream_t stream) {
    if (n == 0) return 0;
    GS_REQUIRE(packed && A_all && EMB_all && D_all && g_all && gbar && g_up && dW && db, "gs_sdf_eikonal_rr_bwd: null pointer");
    if (int rc = check_shape(n_freq, n_hidden, skip_layer)) return rc;
    const PackLayout L = make_layout(n_freq, n_hidden, skip_layer);
    const int64_t Rpad = gs_sdf_eikonal_rr_rows_padded(n);
    hipStream_t st = (hipStream_t)stream;
    H2Args A{};
    A.N = n; A.n_freq = n_freq; A.Rpad = Rpad; A.Pstride = 2 * Rpad;
    A.A = A_all + Rpad * D; A.EMB = EMB_all + Rpad * EK; A.A_in = A_all; A.EMB_in = EMB_all; A.D_in = D_all + Rpad * D; A.S_out = D_all;
    A.gbar = gbar; A.gmul = g_up;
    fill_fwd_args(A, packed, L);
    if (int rc = launch_fwd<MODE_RR>(A, Rpad / TM, st)) return rc;
    BwdArgs B{};
    B.A = A_all; B.EMB = EMB_all; B.Dsave = D_all; B.R = n; B.Rpad = Rpad; B.Pstride = 2 * Rpad; B.E = L.E; B.n_layers = L.n_layers; B.skip_layer = L.skip_layer;
    B.gbar = gbar; B.gmul = g_up;
    for (int l = 0; l < L.n_layers; ++l) {
        B.wfragT[l] = (const h8*)packed + L.fragT_off[l];
        B.nblkT[l] = L.nblkT[l];
    }
    B.w_out = (const float*)((const char*)packed + L.tail_off_bytes) + (int64_t)L.n_layers * D;
    GS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_h2_bwd<MODE_RR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_BWD_BYTES));
    hipLaunchKernelGGL(k_h2_bwd<MODE_RR>, dim3((unsigned)(Rpad / TM)), dim3(NT), SMEM_BWD_BYTES, st, B);
    GS_LAUNCH_CHECK();
    return wgrad_launch(MODE_RR, g_all, 2 * Rpad, 2 * Rpad, nullptr, n_freq, n_hidden, skip_layer, A_all, EMB_all, D_all, dW, db, exact_fp32, st);
}

// ---- compile-time variants of this file (common.hpp): non-default values announce themselves through gs_build_flags(); switches that give
// wrong results (timing-only ablations) compile only under -DGS_EXPERIMENT
GS_TUNABLE(GS_H2_FLUSH, 0)
GS_TUNABLE(GS_H2_BPF, 0)
GS_TUNABLE(GS_H2_ORDER, 0)
GS_TUNABLE(GS_H2_PRIO, 0)
GS_TUNABLE(GS_H2_PD, 1)
GS_TUNABLE(GS_H1_PD, 2)
GS_TUNABLE(GS_H1_ASM, 0)
GS_TUNABLE(GS_H1_PRE, 0)
GS_TUNABLE(GS_H1_POLY, 0)
GS_TUNABLE(GS_H1_ASMMAX, 1)
GS_TUNABLE(GS_H1_WAVES, 6)
GS_TUNABLE(GS_H1_NW, 8)
GS_TUNABLE(GS_H1_RM, 2)
GS_TUNABLE(GS_H1_WAVES4, (GS_H1_RM == 4 ? 2 : 3))
GS_TUNABLE(GS_H1R_ABL, 0)
GS_TUNABLE(GS_H1R_PD, 1)
GS_TUNABLE(GS_H1R_FENCE, 1)
GS_TUNABLE(GS_H2_ABL, 0)
GS_TUNABLE(GS_WG_PIPE, 0)
GS_TUNABLE(GS_WG_CONTIG, 1)
GS_TUNABLE(GS_WG_ABL, 0)
GS_TUNABLE(GS_WG_DEPTH, 1)
GS_TUNABLE(GS_H2_DUAL, 0)
GS_TUNABLE(GS_WG_STRIPS, 80)
GS_TUNABLE_F(GS_WG_W0, 0.6)
#if GS_H1R_ABL != 0
GS_EXPERIMENT_ONLY(GS_H1R_ABL)
#endif
#if GS_H2_ABL != 0
GS_EXPERIMENT_ONLY(GS_H2_ABL)
#endif
#ifdef GS_H2_EMU1
GS_EXPERIMENT_ONLY(GS_H2_EMU1)
#endif
#ifdef GS_H2_NOBAR
GS_EXPERIMENT_ONLY(GS_H2_NOBAR)
#endif
#ifdef GS_H2_NOEPI
GS_EXPERIMENT_ONLY(GS_H2_NOEPI)
#endif
#ifdef GS_H2_NOLDS
GS_EXPERIMENT_ONLY(GS_H2_NOLDS)
#endif
#ifdef GS_H2_NOW
GS_EXPERIMENT_ONLY(GS_H2_NOW)
#endif
#if GS_WG_ABL != 0
GS_EXPERIMENT_ONLY(GS_WG_ABL)
#endif

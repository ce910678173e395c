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
#include "atomics.hpp"
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


// ---------------------------------------------------------------------------------------------------------------------
// The texture-field path of MLPTexture3D.sample (world position -> AABB-normalised, clamped coordinate -> encoding), with
// the feature tensor LEVEL MAJOR ([L][N][2]) between the encoding and the texture MLP (texmlp.hip reads it that way):
//   * the [N, 32] layout makes a per-level workgroup store 8 bytes at a 128-byte stride: 9x the algorithmic traffic
//     measured (profiles/r01_pmc_traffic.json).  Level major, every store and every load of the pair is a full line;
//   * rows with mask <= 0 are never written (the texture MLP never reads them);
//   * the normalisation (x - lo) / (hi - lo), the clamp to [0,1], its gradient mask and the reference's 1/128 gradient
//     scaling hook (mlptexture.py:74) are folded in (they were ~12 + 35 ATen launches around the two kernels).
// Backward: float atomics on gfx950 are fabric writes at a flat 21 G atomics/s whatever the address pattern
// (tools/micro/atomic_scope.hip), so the kernel's time IS its atomic count.  The two features of an entry go in ONE
// 64-bit atomic (atomics.hpp).  A workgroup owns a 16x16-pixel tile of an
// image (neighbouring pixels = neighbouring surface points), walks the levels itself and combines the tile's
// contributions per table entry in an LDS hash table (ds_cmpst + ds_add_f32) before ONE global atomic per entry,
// feature and tile; once a level shows no sharing (>= 3/4 of the inserts claim a fresh slot) the finer levels go to
// global atomics directly.  d loss / d position is accumulated over the levels in registers and written once.
#ifndef GS_HG_LOG_SLOTS
#define GS_HG_LOG_SLOTS 10
#endif
constexpr int CMB_LOG_SLOTS = GS_HG_LOG_SLOTS;
constexpr int CMB_SLOTS = 1 << CMB_LOG_SLOTS;
constexpr uint32_t CMB_EMPTY = 0xffffffffu;

struct EncArgs {
    const float* pos; const float* aabb; const float* mask; int64_t N;
    const float* params; float* feat;
    const float* g_feat; float* g_params; float* g_pos; float grad_scale, table_scale;
    int img_w, img_h;
    const int32_t* rows; const int64_t* count_dev;     // forward: optional compact list of the points to encode (count on the device)
};

__device__ __forceinline__ bool enc_coord(const EncArgs& A, int64_t i, float (&t)[3], bool (&inside)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = A.pos[3 * i + k];
        if (A.aabb) v = (v - A.aabb[k]) / (A.aabb[3 + k] - A.aabb[k]);
        inside[k] = v >= 0.0f && v <= 1.0f;              // torch.clamp passes the gradient on the closed interval
        t[k] = A.aabb ? fminf(fmaxf(v, 0.0f), 1.0f) : v;
        if (v != v) t[k] = v;                            // clamp propagates NaN
    }
    return true;
}

__global__ void __launch_bounds__(256) k_encode_fwd(GridMeta M, EncArgs A) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int l = blockIdx.y;
    if (A.rows) {
        if (i >= A.count_dev[0]) return;
        i = A.rows[i];
    } else if (i >= A.N || (A.mask && !(A.mask[i] > 0.0f))) {
        return;
    }
    float t[3];
    bool inside[3];
    enc_coord(A, i, t, inside);
    const float scale = M.scale[l];
    const uint32_t res = M.res[l], size = M.offset[l + 1] - M.offset[l];
    float w[3];
    uint32_t g0[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float p = t[k] * scale + 0.5f;
        const float fl = floorf(p);
        w[k] = p - fl;
        g0[k] = (uint32_t)(int)fl;
    }
    const float2* tab = reinterpret_cast<const float2*>(A.params) + M.offset[l];
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float wx = (c & 1) ? w[0] : 1.0f - w[0], wy = (c & 2) ? w[1] : 1.0f - w[1], wz = (c & 4) ? w[2] : 1.0f - w[2];
        const float wgt = wx * wy * wz;
        const float2 e = tab[grid_index(g0[0] + (c & 1), g0[1] + ((c >> 1) & 1), g0[2] + ((c >> 2) & 1), res, size)];
        a0 += wgt * e.x;
        a1 += wgt * e.y;
    }
    reinterpret_cast<float2*>(A.feat)[(int64_t)l * A.N + i] = make_float2(a0, a1);
}

#ifndef GS_HG_WAVES
#define GS_HG_WAVES 4
#endif
#ifndef GS_HG_RUNS
#define GS_HG_RUNS 8            // a wave combines in LDS at a level when < GS_HG_RUNS / 8 of its (lane, corner) updates head a run
                                // (measured, ms per backward of both lookups: 6 -> 2.16, 7 -> 1.84, 8 = any run at all -> 1.60, always -> 1.87,
                                //  never = compare-and-swap on contended coarse entries -> 4.9; float atomics instead of the pair swap: 2.2)
#endif
#ifndef GS_HG_ALTERNATE
#define GS_HG_ALTERNATE 1       // odd workgroups walk the levels fine -> coarse
#endif
__global__ void __launch_bounds__(256, GS_HG_WAVES) k_encode_bwd(GridMeta M, EncArgs A, int tiled) {
#ifndef GS_HG_WAVETAB
#define GS_HG_WAVETAB 0      // 1: every wave (4 rows x 16 pixels) combines in its OWN quarter of the table: no workgroup barriers
#endif
    __shared__ uint32_t s_key_all[CMB_SLOTS];
    __shared__ float2 s_val_all[CMB_SLOTS];
    __shared__ uint16_t s_list_all[CMB_SLOTS];   // the slots claimed at the current level (flush + clear walk this list, not the table)
    __shared__ int s_claimed_all[4][2];          // bank = level step & 1
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int TSLOTS = GS_HG_WAVETAB ? CMB_SLOTS / 4 : CMB_SLOTS, TLOG = GS_HG_WAVETAB ? CMB_LOG_SLOTS - 2 : CMB_LOG_SLOTS;
    const int tw = GS_HG_WAVETAB ? (tid >> 6) : 0, tn = GS_HG_WAVETAB ? 64 : 256, tl = GS_HG_WAVETAB ? lane : tid;   // table owner, its threads
    uint32_t* const s_key = s_key_all + tw * TSLOTS;
    float2* const s_val = s_val_all + tw * TSLOTS;
    uint16_t* const s_list = s_list_all + tw * TSLOTS;
    int* const s_claimed = s_claimed_all[tw];
    auto table_sync = [&]() {
        if (GS_HG_WAVETAB) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            __syncthreads();
        }
    };
    int64_t i;
    int wg_linear;
    if (tiled) {                         // 16 x 16 pixel tile of image blockIdx.z
        const int tx = blockIdx.x, ty = blockIdx.y;
        i = ((int64_t)blockIdx.z * A.img_h + ty * 16 + (tid >> 4)) * A.img_w + tx * 16 + (tid & 15);
        wg_linear = tx + ty;
    } else {
        i = (int64_t)blockIdx.x * 256 + tid;
        wg_linear = blockIdx.x;
    }
    const bool in_range = i < A.N;
    const bool active = in_range && !(A.mask && !(A.mask[i] > 0.0f));
    if (!__syncthreads_or(active)) {
        if (in_range && A.g_pos)
            for (int k = 0; k < 3; ++k) A.g_pos[3 * i + k] = 0.f;
        return;
    }
    float t[3] = {0.f, 0.f, 0.f};
    bool inside[3] = {false, false, false};
    if (active) enc_coord(A, i, t, inside);
    for (int s = tl; s < TSLOTS; s += tn) {
        s_key[s] = CMB_EMPTY;
        s_val[s] = make_float2(0.f, 0.f);
    }
    if (tl < 2) s_claimed[tl] = 0;
    table_sync();
    float gx[3] = {0.f, 0.f, 0.f};
    const float2* gf = reinterpret_cast<const float2*>(A.g_feat);
    const int n_active8 = 8 * __popcll(__ballot(active));
    // The coarse levels are LDS / barrier latency, the fine levels are global-atomic throughput: neighbouring workgroups walk the
    // levels in opposite directions so that the chip always has both kinds of work in flight.
    const bool descending = GS_HG_ALTERNATE && (wg_linear & 1);
    for (int step = 0; step < M.n_levels; ++step) {
        const int l = descending ? M.n_levels - 1 - step : step;
        const float scale = M.scale[l];
        const uint32_t res = M.res[l], size = M.offset[l + 1] - M.offset[l];
        float w[3];
        uint32_t g0[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float p = t[k] * scale + 0.5f;
            const float fl = floorf(p);
            w[k] = p - fl;
            g0[k] = (uint32_t)(int)fl;
        }
        const float2* tab = reinterpret_cast<const float2*>(A.params) + M.offset[l];
        float* gtab = A.g_params ? A.g_params + (int64_t)M.offset[l] * 2 : nullptr;
        const float2 go = active ? gf[(int64_t)l * A.N + i] : make_float2(0.f, 0.f);
        float lx = 0.f, ly = 0.f, lz = 0.f;
        int* const claimed = &s_claimed[step & 1];
        // phase A: the eight corner entries (independent gathers, all in flight together) and d loss / d coordinate
        uint32_t idx[8];
        float wgt[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float wx = (c & 1) ? w[0] : 1.0f - w[0], wy = (c & 2) ? w[1] : 1.0f - w[1], wz = (c & 4) ? w[2] : 1.0f - w[2];
            wgt[c] = wx * wy * wz;
            idx[c] = grid_index(g0[0] + (c & 1), g0[1] + ((c >> 1) & 1), g0[2] + ((c >> 2) & 1), res, size);
        }
        if (active && A.g_pos) {
            float2 e[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) e[c] = tab[idx[c]];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float wx = (c & 1) ? w[0] : 1.0f - w[0], wy = (c & 2) ? w[1] : 1.0f - w[1], wz = (c & 4) ? w[2] : 1.0f - w[2];
                const float dotp = go.x * e[c].x + go.y * e[c].y;
                lx += ((c & 1) ? 1.f : -1.f) * wy * wz * dotp;
                ly += ((c & 2) ? 1.f : -1.f) * wx * wz * dotp;
                lz += ((c & 4) ? 1.f : -1.f) * wx * wy * dotp;
            }
        }
        gx[0] += lx * scale;
        gx[1] += ly * scale;
        gx[2] += lz * scale;
        if (!gtab) continue;
        // phase B: runs of equal entries along the wave (16-pixel row segments): segmented suffix sums, the head lane owns the run
        float v0[8], v1[8];
        uint32_t todo = 0u;      // corners this lane has to add to the table
        int n_heads = 0;         // wave-uniform
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t key = active ? idx[c] : (0x80000000u | (uint32_t)lane);
            const uint32_t prev = __shfl_up(key, 1, 64);
            const bool head = lane == 0 || key != prev;
            const uint64_t heads = __ballot(head);
            n_heads += __popcll(__ballot(head && active));
            const uint64_t after = lane == 63 ? 0ull : (heads & ~((2ull << lane) - 1ull));
            const int end = after ? (__ffsll((long long)after) - 2) : 63;
            float a = active ? wgt[c] * (go.x * A.table_scale) : 0.0f, b = active ? wgt[c] * (go.y * A.table_scale) : 0.0f;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const float o0 = __shfl_down(a, d, 64), o1 = __shfl_down(b, d, 64);
                if (lane + d <= end) {
                    a += o0;
                    b += o1;
                }
            }
            v0[c] = a;
            v1[c] = b;
            if (head && active && (a != 0.f || b != 0.f)) todo |= 1u << c;
        }
        // phase C: where neighbouring pixels share entries along the rows they share them across the rows too: combine the tile's
        // updates per entry in LDS (stateless per level and wave, so the level order is free) ...
        const bool combine = 8 * n_heads < GS_HG_RUNS * n_active8;       // wave-uniform
        if (combine) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (!((todo >> c) & 1u)) continue;
                uint32_t slot = (idx[c] * 2654435761u) >> (32 - TLOG);
                bool done = false;
                for (int pr = 0; pr < 8 && !done; ++pr) {
                    const uint32_t old = atomicCAS(&s_key[slot], CMB_EMPTY, idx[c]);
                    if (old == CMB_EMPTY || old == idx[c]) {
                        if (old == CMB_EMPTY) s_list[atomicAdd(claimed, 1)] = (uint16_t)slot;
                        atomicAdd(&s_val[slot].x, v0[c]);
                        atomicAdd(&s_val[slot].y, v1[c]);
                        done = true;
                    } else {
                        slot = (slot + 1) & (TSLOTS - 1);
                    }
                }
                if (done) todo &= ~(1u << c);
            }
        }
        // ... and whatever is left straight to the table: both features in ONE 64-bit compare-and-swap (atomics.hpp), the eight
        // loads and then the eight swaps issued back to back so that their latencies overlap
#ifndef GS_HG_DIRECT_PAIR
#define GS_HG_DIRECT_PAIR 1
#endif
        if (todo && !GS_HG_DIRECT_PAIR) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if ((todo >> c) & 1u) {
                    if (v0[c] != 0.f) atomicAdd(&gtab[(int64_t)idx[c] * 2], v0[c]);
                    if (v1[c] != 0.f) atomicAdd(&gtab[(int64_t)idx[c] * 2 + 1], v1[c]);
                }
        } else if (todo) {
            union PairBits {
                unsigned long long u;
                float2 f;
            };
            PairBits cur[8], seen[8];
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if ((todo >> c) & 1u)
                    cur[c].u = __hip_atomic_load(reinterpret_cast<unsigned long long*>(gtab) + idx[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if ((todo >> c) & 1u) {
                    PairBits nxt;
                    nxt.f = make_float2(cur[c].f.x + v0[c], cur[c].f.y + v1[c]);
                    seen[c].u = atomicCAS(reinterpret_cast<unsigned long long*>(gtab) + idx[c], cur[c].u, nxt.u);
                }
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (((todo >> c) & 1u) && seen[c].u != cur[c].u) {      // lost a race: retry from the value the swap returned
                    PairBits c2 = seen[c];
                    for (;;) {
                        PairBits nxt;
                        nxt.f = make_float2(c2.f.x + v0[c], c2.f.y + v1[c]);
                        const unsigned long long sn = atomicCAS(reinterpret_cast<unsigned long long*>(gtab) + idx[c], c2.u, nxt.u);
                        if (sn == c2.u) break;
                        c2.u = sn;
                    }
                }
        }
        if (GS_HG_WAVETAB ? (table_sync(), combine) : (bool)__syncthreads_or(combine)) {   // uniform over the table's owner; orders inserts before the flush
            const int n = *claimed;
            for (int j = tl; j < n; j += tn) {
                const int s = s_list[j];
                const uint32_t k = s_key[s];
                const float2 v = s_val[s];
                // two float atomics: coarse entries are shared by many tiles, where compare-and-swap retries cost what they save
                if (v.x != 0.f) atomicAdd(&gtab[(int64_t)k * 2], v.x);
                if (v.y != 0.f) atomicAdd(&gtab[(int64_t)k * 2 + 1], v.y);
                s_key[s] = CMB_EMPTY;
                s_val[s] = make_float2(0.f, 0.f);
            }
            table_sync();
            if (tl == 0) *claimed = 0;      // this bank is next used two steps on, behind the next step's barrier
        }
    }
    if (active && A.g_pos) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float g = gx[k] * A.grad_scale;
            if (A.aabb) g = inside[k] ? g / (A.aabb[3 + k] - A.aabb[k]) : 0.0f;
            A.g_pos[3 * i + k] = g;
        }
    } else if (in_range && A.g_pos) {
        for (int k = 0; k < 3; ++k) A.g_pos[3 * i + k] = 0.f;
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

extern "C" int gs_hashgrid_encode_fwd(int n_levels, int F, int log2_T, int base_res, float per_level_scale, const float* pos, const float* aabb,
                                      const float* mask, int64_t N, const float* params, float* feat_level_major, gs_stream_t stream) {
    GridMeta M;
    int rc = make_meta(M, n_levels, F, log2_T, base_res, per_level_scale);
    if (rc) return rc;
    GS_REQUIRE(F == 2, "gs_hashgrid_encode_fwd: the level-major path is built for 2 features per level");
    if (N == 0) return 0;
    GS_REQUIRE(pos && params && feat_level_major, "gs_hashgrid_encode_fwd: null pointer");
    EncArgs A{};
    A.pos = pos; A.aabb = aabb; A.mask = mask; A.N = N; A.params = params; A.feat = feat_level_major;
    hipLaunchKernelGGL(k_encode_fwd, dim3((unsigned)gs::cdiv(N, 256), (unsigned)n_levels), dim3(256), 0, (hipStream_t)stream, M, A);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_hashgrid_encode_fwd_rows(int n_levels, int F, int log2_T, int base_res, float per_level_scale, const float* pos, const float* aabb,
                                           const int32_t* rows, const int64_t* count_dev, int64_t cap, int64_t N, const float* params,
                                           float* feat_level_major, gs_stream_t stream) {
    GridMeta M;
    int rc = make_meta(M, n_levels, F, log2_T, base_res, per_level_scale);
    if (rc) return rc;
    GS_REQUIRE(F == 2, "gs_hashgrid_encode_fwd_rows: the level-major path is built for 2 features per level");
    if (N == 0 || cap == 0) return 0;
    GS_REQUIRE(pos && params && feat_level_major && rows && count_dev && cap <= N, "gs_hashgrid_encode_fwd_rows: null pointer");
    EncArgs A{};
    A.pos = pos; A.aabb = aabb; A.N = N; A.params = params; A.feat = feat_level_major; A.rows = rows; A.count_dev = count_dev;
    hipLaunchKernelGGL(k_encode_fwd, dim3((unsigned)gs::cdiv(cap, 256), (unsigned)n_levels), dim3(256), 0, (hipStream_t)stream, M, A);
    GS_LAUNCH_CHECK();
    return 0;
}

extern "C" int gs_hashgrid_encode_bwd(int n_levels, int F, int log2_T, int base_res, float per_level_scale, const float* pos, const float* aabb,
                                      const float* mask, int64_t N, const float* params, const float* g_feat_level_major, float* g_params,
                                      float* g_pos, float grad_scale, float table_scale, int64_t img_w, int64_t img_h, gs_stream_t stream) {
    GridMeta M;
    int rc = make_meta(M, n_levels, F, log2_T, base_res, per_level_scale);
    if (rc) return rc;
    GS_REQUIRE(F == 2, "gs_hashgrid_encode_bwd: the level-major path is built for 2 features per level");
    if (N == 0) return 0;
    GS_REQUIRE(pos && params && g_feat_level_major, "gs_hashgrid_encode_bwd: null pointer");
    GS_REQUIRE(((uintptr_t)g_params & 7) == 0, "gs_hashgrid_encode_bwd: g_params must be 8-byte aligned (feature pairs are updated with one 64-bit atomic)");
    EncArgs A{};
    A.pos = pos; A.aabb = aabb; A.mask = mask; A.N = N; A.params = params; A.g_feat = g_feat_level_major; A.g_params = g_params; A.g_pos = g_pos;
    A.grad_scale = grad_scale; A.table_scale = table_scale;
    const bool tiled = img_w > 0 && img_h > 0 && img_w % 16 == 0 && img_h % 16 == 0 && N % (img_w * img_h) == 0 && N / (img_w * img_h) < 65536 &&
                       img_h / 16 < 65536;
    A.img_w = (int)img_w; A.img_h = (int)img_h;
    dim3 grid = tiled ? dim3((unsigned)(img_w / 16), (unsigned)(img_h / 16), (unsigned)(N / (img_w * img_h))) : dim3((unsigned)gs::cdiv(N, 256));
    hipLaunchKernelGGL(k_encode_bwd, grid, dim3(256), 0, (hipStream_t)stream, M, A, tiled ? 1 : 0);
    GS_LAUNCH_CHECK();
    return 0;
}

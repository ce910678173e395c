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
    // `idx % size` without the 40-instruction integer division (8 corners x 16 levels per point) wherever it is the identity: a hashed
    // level's table has 2^log2_T entries (make_meta); a dense level's size is >= res^3, so only coordinates outside [0,1] wrap.
    uint64_t dense = (uint64_t)res * res * res;
    if (dense <= (uint64_t)size) {
        const uint32_t idx = x + y * res + z * res * res;
        return idx >= size ? idx % size : idx;
    }
    return (x ^ (y * 2654435761u) ^ (z * 805459861u)) & (size - 1u);
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
// (tools/micro/atomic_scope.hip), so an atomic-only kernel's time IS its atomic count (1.55 ms for the 25 M of the bench frame).
// A workgroup owns a 16x16-pixel tile of an image (neighbouring pixels = neighbouring surface points) and walks the levels itself:
//   * runs of equal entries along a 16-pixel row are summed with DPP row shifts, the head lane owns the run;
//   * DENSE levels (res^3 fits the table: spatially coherent): the tile's updates are combined per entry in an LDS hash table
//     (ds_cmpst + ds_add_f32) and flushed with ONE global atomic per entry, feature and tile; what does not combine goes to the table as
//     one 64-bit compare-and-swap per feature pair (atomics.hpp);
//   * HASHED levels (3/4 of all updates, scattered pseudo-randomly, nothing to combine): no atomics on the table at all, see "Binned
//     table gradient" above -- records sorted by bin in LDS, written as contiguous runs, summed per bin by k_encode_bin_reduce.
// Measured (profiles/r03_hashgrid_bwd.txt): 1.55 ms -> 0.58 + 0.28 ms; LDS float adds cost 3 cycles per lane on gfx950, integer ones
// 0.34 (tools/micro/lds_atomic.hip), which is what the reducer's 0.28 ms is.  d loss / d position is accumulated over the levels in
// registers and written once.
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
    // backward, binned table gradient (below): per-bin fill counters, per-bin record arrays of `bin_cap` records, first bin of a level (-1: atomics)
    uint32_t* bin_count; uint32_t* spill; struct BinRec* bin_rec; uint32_t bin_cap; int bin_base[MAX_LEVELS];
};
// Binned table gradient.  The hashed levels (res^3 > table size) scatter a tile's updates pseudo-randomly over the level's table: no
// sharing to combine, one fabric atomic per (pixel, corner) -- 3/4 of the kernel's atomics.  Instead the table of such a level is cut
// into bins of BIN_ENTRIES consecutive entries; a workgroup ranks its updates per bin in LDS, reserves a run of records per bin with ONE
// returning atomic, and writes (entry, d feature pair) records there with plain 12-byte stores; k_encode_bin_reduce then sums a bin's
// records in 32 KB of LDS and adds them to the table gradient with plain loads and stores (it owns the bin: no atomics at all).
// A reservation past the bin's capacity falls back to the atomic path, so any capacity is correct; the reducer resets the counters.
struct BinRec { uint32_t e; float a, b; };
#ifndef GS_HG_BIN_LOG
#define GS_HG_BIN_LOG 12
#endif
constexpr int BIN_LOG = GS_HG_BIN_LOG, BIN_ENTRIES = 1 << BIN_LOG, BIN_MAX_PER_LEVEL = 256;

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
    // One LDS block, two uses: the combine table of the dense levels (keys, values, the list of slots claimed at the current level: flush
    // and clear walk that list, not the table) and the staging buffer of a binned level's records (sorted by bin before they leave).
    constexpr int STAGE = 256 * 8;                                                     // records of one level: 8 corners per pixel
    constexpr int TAB_WORDS = CMB_SLOTS + 2 * CMB_SLOTS + CMB_SLOTS / 2, RAW_WORDS = TAB_WORDS > 3 * STAGE ? TAB_WORDS : 3 * STAGE;
    __shared__ uint32_t s_raw[RAW_WORDS];
    uint32_t* const s_key_all = s_raw;
    float2* const s_val_all = reinterpret_cast<float2*>(s_raw + CMB_SLOTS);
    uint16_t* const s_list_all = reinterpret_cast<uint16_t*>(s_raw + 3 * CMB_SLOTS);
    uint32_t* const s_rec_e = s_raw;
    float* const s_rec_a = reinterpret_cast<float*>(s_raw + STAGE);
    float* const s_rec_b = reinterpret_cast<float*>(s_raw + 2 * STAGE);
    __shared__ uint32_t s_wave_total[4];
    static_assert(!GS_HG_WAVETAB, "the per-wave combine tables are not laid out for the staging buffer");
    __shared__ int s_claimed_all[4][2];          // bank = level step & 1
    __shared__ uint32_t s_bin_fill[BIN_MAX_PER_LEVEL], s_bin_at[BIN_MAX_PER_LEVEL], s_bin_start[BIN_MAX_PER_LEVEL];
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
    if (tid < BIN_MAX_PER_LEVEL) s_bin_fill[tid] = 0u;
    table_sync();
    float gx[3] = {0.f, 0.f, 0.f};
    const float2* gf = reinterpret_cast<const float2*>(A.g_feat);
    const int n_active8 = 8 * __popcll(__ballot(active));
    // The coarse levels are LDS / barrier latency, the fine levels are global-atomic throughput: neighbouring workgroups walk the
    // levels in opposite directions so that the chip always has both kinds of work in flight.
    const bool descending = GS_HG_ALTERNATE && (wg_linear & 1);
    // Updates straight to the table: both features in ONE 64-bit compare-and-swap (atomics.hpp), the eight loads and then the eight swaps
    // issued back to back so that their latencies overlap.
    auto direct_pairs = [&](float* gtab, const uint32_t (&idx)[8], const float (&v0)[8], const float (&v1)[8], uint32_t todo) {
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
    };
    bool table_dirty = false;            // the staging buffer overwrote the (empty) combine table
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
        // phase B: runs of equal entries along a 16-pixel row segment (= a DPP row of the wave): segmented suffix sums, the head lane owns
        // the run.  Row shifts are DPP operands of the adds (no LDS traffic); a run never leaves its row, which costs nothing: merging is an
        // optimisation, every head issues its own update.
        float v0[8], v1[8];
        uint32_t todo = 0u;      // corners this lane has to add to the table
        int n_heads = 0;         // wave-uniform
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t key = active ? idx[c] : (0x80000000u | (uint32_t)lane);
            const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x111, 0xf, 0xf, false);      // row_shr:1 = lane - 1's key
            const bool head = (lane & 15) == 0 || key != prev;
            const uint64_t heads = __ballot(head);
            n_heads += __popcll(__ballot(head && active));
            const uint64_t after = lane == 63 ? 0ull : (heads & ~((2ull << lane) - 1ull));
            const int end = after ? (__ffsll((long long)after) - 2) : 63;        // <= the row's last lane: the next row starts with a head
            float a = active ? wgt[c] * (go.x * A.table_scale) : 0.0f, b = active ? wgt[c] * (go.y * A.table_scale) : 0.0f;
#define GS_HG_ROW_STEP(D)                                                                                                  \
    if (__ballot(lane + D <= end)) {                                                                                       \
        const float o0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x100 + D, 0xf, 0xf, true));     \
        const float o1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), 0x100 + D, 0xf, 0xf, true));     \
        if (lane + D <= end) {                                                                                             \
            a += o0;                                                                                                       \
            b += o1;                                                                                                       \
        }                                                                                                                  \
    }
            if (__ballot(!head)) {           // (wave-uniform) some run is longer than one lane
                GS_HG_ROW_STEP(1) GS_HG_ROW_STEP(2) GS_HG_ROW_STEP(4) GS_HG_ROW_STEP(8)
            }
#undef GS_HG_ROW_STEP
            v0[c] = a;
            v1[c] = b;
            if (head && active && (a != 0.f || b != 0.f)) todo |= 1u << c;
        }
        // phase C: where neighbouring pixels share entries along the rows they share them across the rows too: combine the tile's
        // updates per entry in LDS (stateless per level and wave, so the level order is free) ...
        const int bbase = A.bin_rec ? A.bin_base[l] : -1;               // workgroup-uniform
        if (bbase >= 0) {
            // (1) rank of every update inside its bin, (2) per bin: reserve a run of records in the bin's array (ONE returning atomic) and
            // scan the counts, (3) stage the records in LDS sorted by bin, (4) copy them out: consecutive lanes write consecutive records of
            // a run (scattered 12-byte stores leave L2 as partial lines: 2.5x the bytes measured, profiles/r03_pmc_hashgrid_bwd.json)
            uint32_t rank[8];
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if ((todo >> c) & 1u) rank[c] = atomicAdd(&s_bin_fill[idx[c] >> BIN_LOG], 1u);
            __syncthreads();
            {
                const uint32_t cnt = tid < (int)((size + BIN_ENTRIES - 1) >> BIN_LOG) ? s_bin_fill[tid] : 0u;    // <= 256 bins: one per thread
                uint32_t incl = cnt;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = __shfl_up(incl, d, 64);
                    if (lane >= d) incl += o;
                }
                if (lane == 63) s_wave_total[tid >> 6] = incl;
                s_bin_start[tid] = incl - cnt;                               // exclusive inside the wave; the waves before are added by the readers
                if (cnt) {
                    s_bin_fill[tid] = 0u;
                    s_bin_at[tid] = atomicAdd(&A.bin_count[bbase + tid], cnt);
                }
            }
            __syncthreads();
            const uint32_t wt0 = s_wave_total[0], wt1 = wt0 + s_wave_total[1], wt2 = wt1 + s_wave_total[2], n_rec = wt2 + s_wave_total[3];
            auto bin_start = [&](uint32_t bin) { return s_bin_start[bin] + (bin < 64u ? 0u : bin < 128u ? wt0 : bin < 192u ? wt1 : wt2); };
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if ((todo >> c) & 1u) {
                    const uint32_t j = bin_start(idx[c] >> BIN_LOG) + rank[c];
                    s_rec_e[j] = idx[c];
                    s_rec_a[j] = v0[c];
                    s_rec_b[j] = v1[c];
                }
            todo = 0u;
            table_dirty = true;
            __syncthreads();
            for (uint32_t j = tid; j < n_rec; j += 256) {
                const uint32_t e = s_rec_e[j], bin = e >> BIN_LOG, at = s_bin_at[bin] + (j - bin_start(bin));
                const float a = s_rec_a[j], b = s_rec_b[j];
                if (at < A.bin_cap) {
                    A.bin_rec[(size_t)(bbase + bin) * A.bin_cap + at] = BinRec{e & (BIN_ENTRIES - 1), a, b};
                } else {                                                     // past the capacity: the atomic path
                    if (a != 0.f) atomicAdd(&gtab[(int64_t)e * 2], a);
                    if (b != 0.f) atomicAdd(&gtab[(int64_t)e * 2 + 1], b);
                }
            }
            continue;                     // the next level's first barrier orders these reads before the buffers' next writes
        }
        if (table_dirty) {                // (workgroup-uniform) first dense level after binned ones: the combine table starts empty again
            __syncthreads();
            for (int sl = tid; sl < CMB_SLOTS; sl += 256) {
                s_key[sl] = CMB_EMPTY;
                s_val[sl] = make_float2(0.f, 0.f);
            }
            table_dirty = false;
            __syncthreads();
        }
        const bool combine = bbase < 0 && 8 * n_heads < GS_HG_RUNS * n_active8;       // wave-uniform
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
        // ... and whatever is left straight to the table
        direct_pairs(gtab, idx, v0, v1, todo);
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

// One workgroup per bin: sum the bin's records per entry in LDS, add to the table gradient (plain read-modify-write: the bin is this
// workgroup's alone and every atomic of k_encode_bwd has landed), reset the bin's counter for the next call.
// The sums are taken in 64-bit FIXED POINT: ds_add_f32 costs 3 - 4 cycles per lane and CU on gfx950, ds_add_u64 0.36 - 0.65
// (tools/micro/lds_atomic.hip).  Pass 1 finds max |v| of the bin's records, pass 2 (the records come back from L2 / Infinity Cache) adds
// round(v * 2^e), e such that the bin's n records cannot overflow 2^61: exact to 2^-37 of the largest record, independent of the order.
// A bin holding a non-finite record is summed with float adds, which propagate it.
__global__ void __launch_bounds__(256) k_encode_bin_reduce(GridMeta M, EncArgs A) {
    __shared__ long long s_fix[BIN_ENTRIES * 2];
    __shared__ float s_max[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = min(A.bin_count[b], A.bin_cap);
    if (n == 0u) return;                                    // (a counter that overflowed is > 0, so it is reset below)
    // records that did not fit took the atomic path: counted in the word after the last bin's counter (never reset here), so that a caller
    // can see that its capacity is too small for its frames
    if (tid == 0 && A.spill && A.bin_count[b] > A.bin_cap) atomicAdd(A.spill, A.bin_count[b] - A.bin_cap);
    int l = 0;
    for (int k = 0; k < M.n_levels; ++k)
        if (A.bin_base[k] >= 0 && A.bin_base[k] <= b) l = k;
    const BinRec* rec = A.bin_rec + (size_t)b * A.bin_cap;
    float vmax = 0.f;
    bool finite = true;
    for (uint32_t j0 = tid; j0 < n; j0 += 256 * 8) {            // eight record loads in flight per lane (clamped, not predicated: a branch
        BinRec r[8];                                            //  around a load makes the compiler wait for it at the join)
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = rec[min(j0 + 256u * u, n - 1u)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            vmax = fmaxf(vmax, fmaxf(fabsf(r[u].a), fabsf(r[u].b)));
            finite = finite && (r[u].a - r[u].a == 0.f) && (r[u].b - r[u].b == 0.f);
        }
    }
    if (!finite) vmax = __builtin_inff();
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d, 64));
    if ((tid & 63) == 0) s_max[tid >> 6] = vmax;
    for (int e = tid; e < BIN_ENTRIES * 2; e += 256) s_fix[e] = 0ll;
    __syncthreads();
    vmax = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    const bool fixed = vmax - vmax == 0.f;                      // workgroup-uniform
    const int ev = (int)((__float_as_uint(vmax) >> 23) & 0xffu) - 126, en = 32 - __clz((int)(n - 1u));       // vmax < 2^ev, n <= 2^en
    const double scale = fixed ? ldexp(1.0, 61 - ev - en) : 1.0;
    float* const s_acc = reinterpret_cast<float*>(s_fix);
    for (uint32_t j0 = tid; j0 < n; j0 += 256 * 8) {
        BinRec r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = rec[min(j0 + 256u * u, n - 1u)];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (j0 + 256u * u < n) {
                if (fixed) {
                    unsigned long long* a = reinterpret_cast<unsigned long long*>(s_fix) + 2 * r[u].e;
                    atomicAdd(a, (unsigned long long)__double2ll_rn((double)r[u].a * scale));
                    atomicAdd(a + 1, (unsigned long long)__double2ll_rn((double)r[u].b * scale));
                } else {
                    atomicAdd(&s_acc[2 * r[u].e], r[u].a);
                    atomicAdd(&s_acc[2 * r[u].e + 1], r[u].b);
                }
            }
    }
    __syncthreads();
    if (tid == 0) A.bin_count[b] = 0u;
    const uint32_t first = (uint32_t)(b - A.bin_base[l]) << BIN_LOG, size = M.offset[l + 1] - M.offset[l];
    float2* g = reinterpret_cast<float2*>(A.g_params) + M.offset[l] + first;
    const int ne = (int)min((uint32_t)BIN_ENTRIES, size - first);
    const double inv = 1.0 / scale;
    for (int e = tid; e < ne; e += 256) {
        const float2 v = fixed ? make_float2((float)((double)s_fix[2 * e] * inv), (float)((double)s_fix[2 * e + 1] * inv)) : make_float2(s_acc[2 * e], s_acc[2 * e + 1]);
        if (v.x != 0.f || v.y != 0.f) {
            float2 t = g[e];
            t.x += v.x;
            t.y += v.y;
            g[e] = t;
        }
    }
}

// levels whose table gradient is binned: hashed (no spatial coherence to combine) and at most BIN_MAX_PER_LEVEL bins; -> number of bins
static int bin_layout(const GridMeta& M, int (&bin_base)[MAX_LEVELS]) {
    int n = 0;
    for (int l = 0; l < M.n_levels; ++l) {
        const uint64_t size = M.offset[l + 1] - M.offset[l], dense = (uint64_t)M.res[l] * M.res[l] * M.res[l];
        const uint64_t nb = (size + BIN_ENTRIES - 1) >> BIN_LOG;
        if (dense > size && nb <= (uint64_t)BIN_MAX_PER_LEVEL) {
            bin_base[l] = n;
            n += (int)nb;
        } else {
            bin_base[l] = -1;
        }
    }
    return n;
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
        n = std::min<uint64_t>(n, 1ull << log2_T);        // (a hashed level, res^3 > n, therefore has exactly 2^log2_T entries: grid_index masks)
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

static int encode_bwd_impl(int n_levels, int F, int log2_T, int base_res, float per_level_scale, const float* pos, const float* aabb, const float* mask,
                           int64_t N, const float* params, const float* g_feat_level_major, float* g_params, float* g_pos, float grad_scale,
                           float table_scale, int64_t img_w, int64_t img_h, uint32_t* bin_count, uint32_t* spill_count, void* bin_records,
                           int64_t bin_capacity, gs_stream_t stream, const char* who) {
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
    int n_bins = 0;
    if (bin_records && g_params) {
        GS_REQUIRE(bin_count && bin_capacity > 0 && bin_capacity < (1ll << 31), "gs_hashgrid_encode_bwd_binned: bin counters / capacity missing");
        n_bins = bin_layout(M, A.bin_base);
        if (n_bins > 0) {
            A.bin_count = bin_count; A.spill = spill_count; A.bin_rec = static_cast<BinRec*>(bin_records); A.bin_cap = (uint32_t)bin_capacity;
        }
    }
    const bool tiled = img_w > 0 && img_h > 0 && img_w % 16 == 0 && img_h % 16 == 0 && N % (img_w * img_h) == 0 && N / (img_w * img_h) < 65536 &&
                       img_h / 16 < 65536;
    A.img_w = (int)img_w; A.img_h = (int)img_h;
    dim3 grid = tiled ? dim3((unsigned)(img_w / 16), (unsigned)(img_h / 16), (unsigned)(N / (img_w * img_h))) : dim3((unsigned)gs::cdiv(N, 256));
    hipLaunchKernelGGL(k_encode_bwd, grid, dim3(256), 0, (hipStream_t)stream, M, A, tiled ? 1 : 0);
    if (n_bins > 0) hipLaunchKernelGGL(k_encode_bin_reduce, dim3((unsigned)n_bins), dim3(256), 0, (hipStream_t)stream, M, A);
    GS_LAUNCH_CHECK();
    (void)who;
    return 0;
}

extern "C" int gs_hashgrid_encode_bwd(int n_levels, int F, int log2_T, int base_res, float per_level_scale, const float* pos, const float* aabb,
                                      const float* mask, int64_t N, const float* params, const float* g_feat_level_major, float* g_params,
                                      float* g_pos, float grad_scale, float table_scale, int64_t img_w, int64_t img_h, gs_stream_t stream) {
    return encode_bwd_impl(n_levels, F, log2_T, base_res, per_level_scale, pos, aabb, mask, N, params, g_feat_level_major, g_params, g_pos, grad_scale,
                           table_scale, img_w, img_h, nullptr, nullptr, nullptr, 0, stream, "gs_hashgrid_encode_bwd");
}

extern "C" int64_t gs_hashgrid_bin_count(int n_levels, int F, int log2_T, int base_res, float per_level_scale) {
    GridMeta M;
    if (make_meta(M, n_levels, F, log2_T, base_res, per_level_scale)) return -1;
    int bin_base[MAX_LEVELS];
    return bin_layout(M, bin_base);
}

extern "C" int64_t gs_hashgrid_bin_entries(void) { return BIN_ENTRIES; }

extern "C" int gs_hashgrid_encode_bwd_binned(int n_levels, int F, int log2_T, int base_res, float per_level_scale, const float* pos, const float* aabb,
                                             const float* mask, int64_t N, const float* params, const float* g_feat_level_major, float* g_params,
                                             float* g_pos, float grad_scale, float table_scale, int64_t img_w, int64_t img_h, uint32_t* bin_count,
                                             uint32_t* spill_count, void* bin_records, int64_t bin_capacity, gs_stream_t stream) {
    return encode_bwd_impl(n_levels, F, log2_T, base_res, per_level_scale, pos, aabb, mask, N, params, g_feat_level_major, g_params, g_pos, grad_scale,
                           table_scale, img_w, img_h, bin_count, spill_count, bin_records, bin_capacity, stream, "gs_hashgrid_encode_bwd_binned");
}

// ---- compile-time variants of this file (common.hpp): non-default values announce themselves through gs_build_flags(); switches that give
// wrong results (timing-only ablations) compile only under -DGS_EXPERIMENT
GS_TUNABLE(GS_HG_LOG_SLOTS, 10)
GS_TUNABLE(GS_HG_BIN_LOG, 12)
GS_TUNABLE(GS_HG_WAVES, 4)
GS_TUNABLE(GS_HG_RUNS, 8)
GS_TUNABLE(GS_HG_ALTERNATE, 1)
GS_TUNABLE(GS_HG_WAVETAB, 0)
GS_TUNABLE(GS_HG_DIRECT_PAIR, 1)

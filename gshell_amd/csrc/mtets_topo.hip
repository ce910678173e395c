// Static tet-grid topology for G-MarchingTets, built once per grid on the device.
//
// Replaces the reference's per-iteration `torch.unique(all_edges, dim=0, return_inverse=True)`
// (geometry/gshell_tets.py:266-268) and `generate_edges` (geometry/gshell_tets_geometry.py:149-155):
// the mesh vertex id of a crossing edge is its rank among crossing edges in lexicographic
// (min,max) order, which only needs this sorted list + a per-call prefix count.
#include <algorithm>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../include/gshell_hip.h"
#include "mtets_internal.hpp"

namespace {

__global__ void k_tets_to_i32_keys(const int64_t* __restrict__ tet64, int64_t F, int32_t* __restrict__ tet32,
                                   uint64_t* __restrict__ keys, uint32_t* __restrict__ slots) {
    int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    int32_t v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = (int32_t)tet64[f * 4 + i];
        tet32[f * 4 + i] = v[i];
    }
    const int ca[6] = {0, 0, 0, 1, 1, 2}, cb[6] = {1, 2, 3, 2, 3, 3};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        uint32_t a = (uint32_t)min(v[ca[k]], v[cb[k]]), b = (uint32_t)max(v[ca[k]], v[cb[k]]);
        keys[f * 6 + k] = ((uint64_t)a << 32) | b;
        slots[f * 6 + k] = (uint32_t)(f * 6 + k);
    }
}

__global__ void k_head_flags(const uint64_t* __restrict__ keys, int64_t n, uint32_t* __restrict__ head) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ void k_scatter_unique(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ slots,
                                 const uint32_t* __restrict__ rank_incl, int64_t n, int32_t* __restrict__ edges,
                                 int32_t* __restrict__ tet_edge) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r = rank_incl[i] - 1u;
    tet_edge[slots[i]] = (int32_t)r;
    if (i == 0 || keys[i] != keys[i - 1]) {
        edges[2 * (int64_t)r] = (int32_t)(keys[i] >> 32);
        edges[2 * (int64_t)r + 1] = (int32_t)(keys[i] & 0xffffffffu);
    }
}

}  // namespace

extern "C" int gs_mtets_topo_create(const int64_t* tet_fx4, int64_t F, int64_t N, gs_stream_t stream_,
                                    gs_mtets_topo** out) {
    GS_REQUIRE(out != nullptr, "gs_mtets_topo_create: out is null");
    GS_REQUIRE(F >= 0 && N >= 0 && N < (1ll << 31) && F * 6 < (1ll << 32), "grid too large for int32 indices");
    hipStream_t stream = (hipStream_t)stream_;
    gs_mtets_topo* t = new gs_mtets_topo();
    t->N = N;
    t->F = F;
    const int64_t n = F * 6;
    uint64_t *keys = nullptr, *keys2 = nullptr;
    uint32_t *slots = nullptr, *slots2 = nullptr, *head = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0, tmp2 = 0;
    GS_HIP_CHECK(hipMalloc(&t->tet, sizeof(int32_t) * 4 * (size_t)std::max<int64_t>(F, 1)));
    GS_HIP_CHECK(hipMalloc(&t->tet_edge, sizeof(int32_t) * (size_t)std::max<int64_t>(n, 1)));
    if (F > 0) {
        GS_HIP_CHECK(hipMalloc(&keys, sizeof(uint64_t) * n));
        GS_HIP_CHECK(hipMalloc(&keys2, sizeof(uint64_t) * n));
        GS_HIP_CHECK(hipMalloc(&slots, sizeof(uint32_t) * n));
        GS_HIP_CHECK(hipMalloc(&slots2, sizeof(uint32_t) * n));
        GS_HIP_CHECK(hipMalloc(&head, sizeof(uint32_t) * n));
        k_tets_to_i32_keys<<<gs::cdiv(F, 256), 256, 0, stream>>>(tet_fx4, F, t->tet, keys, slots);
        GS_LAUNCH_CHECK();
        // vertex ids < 2^31: sort only the bits that can be set
        int bits_lo = 1;
        while ((1ll << bits_lo) < std::max<int64_t>(N, 2)) ++bits_lo;
        GS_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys2, slots, slots2, (size_t)n, 0, 32 + bits_lo,
                                               stream));
        GS_HIP_CHECK(rocprim::inclusive_scan(nullptr, tmp2, head, head, (size_t)n, rocprim::plus<uint32_t>(), stream));
        tmp_bytes = std::max(tmp_bytes, tmp2);
        GS_HIP_CHECK(hipMalloc(&tmp, tmp_bytes));
        GS_HIP_CHECK(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, slots, slots2, (size_t)n, 0, 32 + bits_lo,
                                               stream));
        k_head_flags<<<gs::cdiv(n, 256), 256, 0, stream>>>(keys2, n, head);
        GS_LAUNCH_CHECK();
        GS_HIP_CHECK(rocprim::inclusive_scan(tmp, tmp_bytes, head, head, (size_t)n, rocprim::plus<uint32_t>(), stream));
        uint32_t E32 = 0;
        GS_HIP_CHECK(hipMemcpyAsync(&E32, head + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        GS_HIP_CHECK(hipStreamSynchronize(stream));
        t->E = E32;
        GS_HIP_CHECK(hipMalloc(&t->edges, sizeof(int32_t) * 2 * (size_t)t->E));
        k_scatter_unique<<<gs::cdiv(n, 256), 256, 0, stream>>>(keys2, slots2, head, n, t->edges, t->tet_edge);
        GS_LAUNCH_CHECK();
        GS_HIP_CHECK(hipStreamSynchronize(stream));
        hipFree(keys); hipFree(keys2); hipFree(slots); hipFree(slots2); hipFree(head); hipFree(tmp);
    } else {
        GS_HIP_CHECK(hipMalloc(&t->edges, 8));
    }
    // per-call scratch
    t->nchunks = gs::cdiv(std::max<int64_t>(t->E, 1), 64);
    t->nb_e = gs::cdiv(t->nchunks, MT_CHUNKS_PER_BLOCK);
    t->nb_t = gs::cdiv(std::max<int64_t>(F, 1), MT_TETS_PER_BLOCK);
    GS_HIP_CHECK(hipMalloc(&t->occ_bits, sizeof(uint64_t) * (size_t)gs::cdiv(std::max<int64_t>(N, 1), 64)));
    GS_HIP_CHECK(hipMalloc(&t->tet_code, (size_t)std::max<int64_t>(F, 1)));
    GS_HIP_CHECK(hipMalloc(&t->edge_mask, sizeof(uint64_t) * (size_t)t->nchunks));
    GS_HIP_CHECK(hipMalloc(&t->chunk_base, sizeof(int32_t) * (size_t)t->nchunks));
    GS_HIP_CHECK(hipMalloc(&t->tet_blk, sizeof(int32_t) * MT_NCAT * (size_t)t->nb_t));
    GS_HIP_CHECK(hipMalloc(&t->edge_blk, sizeof(int32_t) * (size_t)t->nb_e));
    GS_HIP_CHECK(hipMalloc(&t->counts_dev, sizeof(int64_t) * GS_MTETS_NCOUNTS));
    GS_HIP_CHECK(hipHostMalloc(&t->counts_host, sizeof(int64_t) * GS_MTETS_NCOUNTS));
    *out = t;
    return 0;
}

extern "C" int gs_mtets_topo_destroy(gs_mtets_topo* t) {
    if (!t) return 0;
    hipFree(t->tet); hipFree(t->edges); hipFree(t->tet_edge); hipFree(t->occ_bits); hipFree(t->tet_code);
    hipFree(t->edge_mask); hipFree(t->chunk_base); hipFree(t->tet_blk); hipFree(t->edge_blk); hipFree(t->counts_dev);
    if (t->counts_host) hipHostFree(t->counts_host);
    delete t;
    return 0;
}

extern "C" int gs_mtets_topo_info(const gs_mtets_topo* t, int64_t* N, int64_t* F, int64_t* E,
                                  const int32_t** edges_dev, const int32_t** tet_i32_dev) {
    GS_REQUIRE(t != nullptr, "gs_mtets_topo_info: topo is null");
    if (N) *N = t->N;
    if (F) *F = t->F;
    if (E) *E = t->E;
    if (edges_dev) *edges_dev = t->edges;
    if (tet_i32_dev) *tet_i32_dev = t->tet;
    return 0;
}

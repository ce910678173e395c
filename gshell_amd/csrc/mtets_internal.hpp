// Internal layout of gs_mtets_topo (static grid topology + per-call scratch).
#pragma once
#include "common.hpp"

// Launch geometry shared by the count and fill passes (they must agree: the fill pass
// re-derives per-block offsets that the count pass accumulated).
constexpr int MT_BLOCK = 256;            // threads per block = 4 waves
constexpr int MT_TILES = 16;             // 256-tet tiles per k_classify block (counts granularity)
constexpr int MT_TETS_PER_BLOCK = MT_BLOCK * MT_TILES;
constexpr int MT_COMPACT_SPAN = 4;       // k_compact block = 4 consecutive k_classify blocks
constexpr int MT_CHUNKS_PER_BLOCK = 256; // 64-edge chunks per block (one per thread in scan)
constexpr int MT_NCAT = 8;               // n1, n2, tri->1, tri->2, quad->1..4

struct gs_mtets_topo {
    int64_t N = 0, F = 0, E = 0;
    // static
    int32_t* tet = nullptr;        // [F,4]
    int32_t* edges = nullptr;      // [E,2] sorted (min,max) lexicographic
    int32_t* tet_edge = nullptr;   // [F,6] index into edges, base order 01 02 03 12 13 23
    // per-call scratch
    uint64_t* occ_bits = nullptr;  // [ceil(N/64)]  bit i = sdf[i] > 0
    uint8_t* tet_code = nullptr;   // [F] sign pattern | (mSDF cut index << 4)
    uint64_t* edge_mask = nullptr; // [nchunks] crossing bit per edge
    int32_t* chunk_base = nullptr; // [nchunks] vertex id of the chunk's first crossing edge
    int32_t* tet_blk = nullptr;    // [nb_t, 8] per-block category counts
    int32_t* edge_blk = nullptr;   // [nb_e]    per-block crossing counts
    int64_t* counts_dev = nullptr; // [GS_MTETS_NCOUNTS]
    int64_t* counts_host = nullptr;// pinned
    int64_t nchunks = 0, nb_t = 0, nb_e = 0;
    int64_t last_counts[16] = {0};
};

// Scattered float accumulation on gfx950.
//
// A float atomic is a read-modify-write at the memory side of the fabric: 21-24 G atomics/s for the whole chip whatever the
// type (f32, f64, u64), scope or footprint (tools/micro/atomic_scope.hip, atomic_pair.hip).  What can be chosen is how many
// VALUES an atomic carries.  There is no packed-f32 atomic add, but a 64-bit compare-and-swap on a float PAIR -- agent-scope
// load of the pair, add both halves, global_atomic_cmpswap_x2, retry with the returned value on a race -- is ONE atomic for two
// floats and exact fp32: 23.6 G pairs/s against 10.4 G pairs/s for two global_atomic_add_f32 (8 M-entry table; 17.7 G pairs/s
// with 64 colliding updates per entry).  It pays where updates of one address are RARE (fine hash-grid levels); on heavily
// shared addresses (the light probe's bright texels) the retries make it slower than plain float atomics.
#pragma once
#include <hip/hip_runtime.h>

namespace gs {

// p must be 8-byte aligned; *p += a, *(p+1) += b as one atomic
__device__ __forceinline__ void atomic_add_pair(float* p, float a, float b) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
    union {
        unsigned long long u;
        float2 f;
    } cur, nxt;
    cur.u = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        nxt.f = make_float2(cur.f.x + a, cur.f.y + b);
        const unsigned long long seen = atomicCAS(q, cur.u, nxt.u);
        if (seen == cur.u) break;
        cur.u = seen;
    }
}

}  // namespace gs

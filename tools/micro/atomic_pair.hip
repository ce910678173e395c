// How cheaply can gfx950 add a float PAIR (the two features of a hash-grid entry) into a table at a random address?
//   0: two global_atomic_add_f32                                 (what hashgrid.hip does)
//   1: 64-bit load (agent-scope relaxed) + global_atomic_cmpswap_x2 loop on the pair
//   2: cmpswap_x2 against an expected (0,0) first, loop on failure
//   3: one 64-bit integer atomic add  (rate reference: a fixed-point pair would need only this)
//   4: one f64 atomic add             (rate reference)
// hipcc --offload-arch=gfx950 -O2 -munsafe-fp-atomics -o /tmp/ap tools/micro/atomic_pair.hip && /tmp/ap
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ uint32_t pcg(uint32_t v) { uint32_t s = v * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u; return (w >> 22u) ^ w; }
union Pair { unsigned long long u; float2 f; };
template <int MODE>
__global__ void k(float2* tab, uint32_t n, int per_thread) {
    uint32_t r = pcg(blockIdx.x * blockDim.x + threadIdx.x);
    for (int i = 0; i < per_thread; ++i) {
        r = pcg(r);
        float2* p = tab + (r % n);
        if (MODE == 0) { atomicAdd(&p->x, 1.0f); atomicAdd(&p->y, 2.0f); }
        if (MODE == 1 || MODE == 2) {
            unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
            Pair cur, nxt;
            cur.u = MODE == 1 ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            for (;;) {
                nxt.f = make_float2(cur.f.x + 1.0f, cur.f.y + 2.0f);
                const unsigned long long seen = atomicCAS(q, cur.u, nxt.u);
                if (seen == cur.u) break;
                cur.u = seen;
            }
        }
        if (MODE == 3) atomicAdd(reinterpret_cast<unsigned long long*>(p), (2ull << 32) | 1ull);
        if (MODE == 4) atomicAdd(reinterpret_cast<double*>(p), 1.0);
    }
}
__global__ void check(const float2* tab, uint32_t n, int mode, double* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a, b;
    if (mode == 3) { Pair v; v.f = tab[i]; a = (double)(uint32_t)v.u; b = (double)(uint32_t)(v.u >> 32); }
    else if (mode == 4) { a = reinterpret_cast<const double*>(tab)[i]; b = 2 * a; }
    else { a = tab[i].x; b = tab[i].y; }
    atomicAdd(out, a); atomicAdd(out + 1, b);
}
int main() {
    const int per_thread = 32, blocks = 4096, threads = 256;
    const double total = (double)per_thread * blocks * threads;
    const char* names[5] = {"2 x atomic_add_f32", "load + cmpswap_x2", "cmpswap_x2 vs (0,0) first", "1 x atomic_add_u64", "1 x atomic_add_f64"};
    for (uint32_t n : {1u << 19, 1u << 23}) {
        float2* tab; double* out; hipMalloc(&tab, (size_t)n * 8); hipMalloc(&out, 16);
        for (int mode = 0; mode < 5; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(tab, 0, (size_t)n * 8); hipMemset(out, 0, 16);
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, tab, n, per_thread);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, tab, n, per_thread);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, tab, n, per_thread);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, 0, tab, n, per_thread);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(threads), 0, 0, tab, n, per_thread);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            hipLaunchKernelGGL(check, dim3((n + 255) / 256), dim3(256), 0, 0, tab, n, mode, out);
            double h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            printf("entries %8u  %-28s %7.3f ms  %6.1f G pairs/s  sums %.0f %.0f (want %.0f %.0f) %s\n", n, names[mode], ms, total / ms / 1e6, h[0], h[1], total,
                   2 * total, (h[0] == total && h[1] == 2 * total) ? "OK" : "MISMATCH");
        }
        hipFree(tab); hipFree(out);
    }
    return 0;
}

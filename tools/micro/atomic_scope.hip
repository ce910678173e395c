// Float-atomic throughput on gfx950 by memory scope and footprint: does a workgroup-scope atomic stay in the XCD's L2 (no
// fabric write per atomic)?  If so, per-XCD replicas (selected by HW_REG_XCC_ID) + one final reduction make scattered gradient
// accumulation (light probe, hash-grid tables) several times cheaper.   hipcc --offload-arch=gfx950 -O2 -o /tmp/as tools/micro/atomic_scope.hip && /tmp/as
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ uint32_t pcg(uint32_t v) { uint32_t s = v * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u; return (w >> 22u) ^ w; }
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7; }   // HW_REG_XCC_ID = 20, bits [3:0]
template <int MODE>
__global__ void k(float* buf, uint32_t n_entries, int per_thread, int replicas) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    float* base = buf;
    if (MODE == 2) base = buf + (size_t)(xcc_id() % replicas) * n_entries;
    uint32_t r = pcg(tid);
    for (int i = 0; i < per_thread; ++i) {
        r = pcg(r);
        float* p = base + (r % n_entries);
        if (MODE == 0) atomicAdd(p, 1.0f);                                                              // agent scope (default)
        else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);           // workgroup scope
    }
}
__global__ void reduce(const float* buf, uint32_t n, int replicas, double* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0; for (int r = 0; r < replicas; ++r) s += buf[(size_t)r * n + i];
    atomicAdd(out, (double)s);
}
int main() {
    const int per_thread = 64, blocks = 4096, threads = 256;
    const double total = (double)per_thread * blocks * threads;
    for (uint32_t n : {196608u /* 256x256x3 */, 1u << 20, 1u << 23}) {
        float* buf; double* out; hipMalloc(&buf, (size_t)n * 8 * 4); hipMalloc(&out, 8);
        for (int mode = 0; mode < 3; ++mode) {
            hipMemset(buf, 0, (size_t)n * 8 * 4); hipMemset(out, 0, 8);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                if (rep == 1) { hipMemset(buf, 0, (size_t)n * 8 * 4); hipEventRecord(e0); }
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, buf, n, per_thread, 1);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, buf, n, per_thread, 1);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, buf, n, per_thread, 8);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipLaunchKernelGGL(reduce, dim3((n + 255) / 256), dim3(256), 0, 0, buf, n, mode == 2 ? 8 : 1, out);
            double h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
            const char* names[3] = {"agent-scope atomicAdd, 1 buffer", "workgroup-scope, 1 buffer (NOT coherent across XCDs)", "workgroup-scope, 8 per-XCD replicas + reduce"};
            printf("entries %8u  %-55s %7.3f ms  %6.1f G atomics/s  sum %.0f / %.0f %s\n", n, names[mode], ms, total / ms / 1e6, h, total, h == total ? "OK" : "LOST UPDATES");
        }
        hipFree(buf); hipFree(out);
    }
    return 0;
}

// LDS atomic throughput on gfx950 (per CU): ds_add_f32 / ds_add_rtn_u32 / ds_add_u32 over random addresses, by footprint and by how
// many lanes of a wave share an address.  Decides how the hash-grid table gradient bins and reduces its records (hashgrid.hip).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o /tmp/la tools/micro/lds_atomic.hip && /tmp/la
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ uint32_t pcg(uint32_t v) { uint32_t s = v * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u; return (w >> 22u) ^ w; }
// MODE 0: float add no return, 1: u32 add returning, 2: u32 add no return, 3: plain ds_write (no atomic) as the LDS-rate reference,
//      4: float add, interleaved float2 layout (x at even dwords only), 5: u64 add no return (fixed-point accumulation), 6: the same + the
//      float -> fixed-point conversion through double
template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t n_entries, int per_thread, int share, float* out) {
    __shared__ uint32_t s[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) s[i] = 0;
    __syncthreads();
    uint32_t r = pcg(blockIdx.x * 256 + (threadIdx.x / share) * share);     // `share` consecutive lanes draw the same addresses
    uint32_t acc = 0;
    for (int i = 0; i < per_thread; ++i) {
        r = pcg(r);
        uint32_t e = r % n_entries;
        if (MODE == 0) atomicAdd(reinterpret_cast<float*>(s) + e, 1.0f);
        if (MODE == 1) acc += atomicAdd(s + e, 1u);
        if (MODE == 2) atomicAdd(s + e, 1u);
        if (MODE == 3) s[e] = r;
        if (MODE == 4) atomicAdd(reinterpret_cast<float*>(s) + 2 * (e % 4096), 1.0f);
        if (MODE == 5) atomicAdd(reinterpret_cast<unsigned long long*>(s) + (e % 4096), (unsigned long long)r);
        if (MODE == 6) atomicAdd(reinterpret_cast<unsigned long long*>(s) + (e % 4096), (unsigned long long)(long long)((double)__uint_as_float((r & 0x007fffffu) | 0x3f000000u) * 1.0e12));
    }
    __syncthreads();
    if (acc == 0xdeadbeef || s[threadIdx.x] == 0xdeadbeef) out[0] = 1.f;
}
int main() {
    const int per_thread = 512, blocks = 256 * 8, threads = 256;
    float* out; hipMalloc(&out, 4);
    const char* names[7] = {"ds_add_f32", "ds_add_rtn_u32", "ds_add_u32", "ds_write_b32", "ds_add_f32 (even dwords)", "ds_add_u64", "ds_add_u64 + f32->fixed"};
    for (uint32_t n : {4096u})
        for (int share : {1, 4, 16})
            for (int mode = 0; mode < 7; ++mode) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                for (int rep = 0; rep < 2; ++rep) {
                    if (rep == 1) hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, n, per_thread, share, out);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, n, per_thread, share, out);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, n, per_thread, share, out);
                    if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, 0, n, per_thread, share, out);
                    if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(threads), 0, 0, n, per_thread, share, out);
                    if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(threads), 0, 0, n, per_thread, share, out);
                    if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(threads), 0, 0, n, per_thread, share, out);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double ops = (double)per_thread * blocks * threads;
                printf("entries %5u share %2d  %-26s %7.3f ms  %7.1f G lane-ops/s  = %5.2f cycles per lane-op per CU (256 CUs, 2.4 GHz)\n", n, share, names[mode], ms,
                       ops / ms / 1e6, ms * 1e-3 * 2.4e9 * 256 / ops);
            }
    return 0;
}

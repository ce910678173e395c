// A side-queue load whose only special property is that it uses SCRATCH (private memory): a per-lane array indexed dynamically (tools/raster_race_probe5.py).
// `rounds` controls the duration; with use_scratch = 0 the same arithmetic runs out of registers (control).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/scratch_load.hip -o gshell_amd/lib/variants/scratch_load.so
#include <hip/hip_runtime.h>
#include <cstdint>

template <bool SCRATCH>
__global__ void __launch_bounds__(256) k_load(uint32_t* __restrict__ out, int rounds, uint32_t seed) {
    uint32_t x = seed ^ (blockIdx.x * 256u + threadIdx.x) * 0x9e3779b9u;
    if (SCRATCH) {
        uint32_t a[96];
#pragma unroll 1
        for (int i = 0; i < 96; ++i) a[i] = x + i;
#pragma unroll 1
        for (int r = 0; r < rounds; ++r) {
            x = x * 1664525u + 1013904223u;
            uint32_t j = (x >> 8) % 96u;
            a[j] += x;                              // dynamic index: the array lives in private memory
            x ^= a[(j * 7u + 3u) % 96u];
        }
    } else {
#pragma unroll 1
        for (int r = 0; r < rounds; ++r) {
            x = x * 1664525u + 1013904223u;
            x ^= (x >> 7) * 2654435761u;
        }
    }
    if (x == 0x1234567u) out[0] = x;
}

extern "C" int load_launch(int use_scratch, int blocks, int rounds, void* out, void* stream) {
    if (use_scratch) hipLaunchKernelGGL(k_load<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (uint32_t*)out, rounds, 12345u);
    else hipLaunchKernelGGL(k_load<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (uint32_t*)out, rounds, 12345u);
    return (int)hipGetLastError();
}

// How many independent accumulators keep ONE wave per SIMD at the v_mfma_f32_32x32x16_f16 issue rate?  (round 5, k_h1r_fwd design question)
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_chain.hip -o tools/micro/bin/mfma_chain && tools/micro/bin/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int NACC, int FILL>
__global__ void __launch_bounds__(256, 1) k(const h8* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc, int iters) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x;
    h8 a = in[tid], b = in[256 + tid];
    v16f acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = (float)(i + j);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)a[j];
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < FILL; ++j) f[j % 8] = __builtin_fmaf(f[j % 8], 1.0001f, 0.5f);      // independent single-issue fillers
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += f[j];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    (void)smem;
}

template <int NACC, int FILL>
void run(const h8* in, float* out, long long* cyc) {
    const int iters = 200, grid = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<NACC, FILL>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL((k<NACC, FILL>), dim3(grid), dim3(256), 100 * 1024, 0, in, out, cyc, iters);
    hipLaunchKernelGGL((k<NACC, FILL>), dim3(grid), dim3(256), 100 * 1024, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < grid; ++i) m += (double)h[i];
    m /= grid;
    printf("accumulators %d, fillers per MFMA %d: %.1f clock64 ticks per MFMA\n", NACC, FILL, m / (iters * 8.0 * NACC));
}

int main() {
    h8* in; float* out; long long* cyc;
    hipMalloc(&in, 512 * sizeof(h8)); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    hipMemset(in, 0, 512 * sizeof(h8));
    run<1, 0>(in, out, cyc); run<2, 0>(in, out, cyc); run<4, 0>(in, out, cyc); run<8, 0>(in, out, cyc);
    run<2, 4>(in, out, cyc); run<2, 8>(in, out, cyc); run<4, 4>(in, out, cyc); run<4, 8>(in, out, cyc); run<4, 12>(in, out, cyc);
    run<1, 4>(in, out, cyc); run<1, 8>(in, out, cyc);
    return 0;
}

// Where does a one-wave-per-SIMD MFMA stream lose its issue rate?  B operands from a 32-fragment register array (AGPRs), A operands from LDS with
// a prefetch ring, two accumulators -- the operand pattern of k_h1r_fwd, piece by piece.  (round 5)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <bool B_ARRAY, bool A_LDS, int PD, bool BARRIER>
__global__ void __launch_bounds__(256, 1) k(const h8* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) h8 smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 1280 * 3; i += 256) smem[i] = in[i & 511];
    __syncthreads();
    h8 Bin[16][2];
#pragma unroll
    for (int s = 0; s < 16; ++s) { Bin[s][0] = in[(tid + s) & 511]; Bin[s][1] = in[(tid + 2 * s + 1) & 511]; }
    v16f acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = (float)(i + j);
    const h8 a_reg = in[tid];
    const long long t0 = clock64();
    int buf = 0;
    for (int it = 0; it < iters; ++it) {
        const h8* wl = smem + buf * 1280 + lane;
        h8 fr[PD + 1];
        if (A_LDS) {
#pragma unroll
            for (int i = 0; i < PD; ++i) fr[i] = wl[i * 64];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (A_LDS && s + PD < 16) fr[(s + PD) % (PD + 1)] = wl[(s + PD) * 64];
            const h8 a = A_LDS ? fr[s % (PD + 1)] : a_reg;
            const h8 b0 = B_ARRAY ? Bin[s][0] : Bin[0][0], b1 = B_ARRAY ? Bin[s][1] : Bin[0][1];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BARRIER) __syncthreads();
        buf = buf == 2 ? 0 : buf + 1;
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
#pragma unroll
    for (int q = 0; q < 16; ++q) s += (float)Bin[q][0][0] + (float)Bin[q][1][1];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <bool B_ARRAY, bool A_LDS, int PD, bool BARRIER>
void run(const h8* in, float* out, long long* cyc, const char* what) {
    const int iters = 400, grid = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<B_ARRAY, A_LDS, PD, BARRIER>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<B_ARRAY, A_LDS, PD, BARRIER>), dim3(grid), dim3(256), 100 * 1024, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < grid; ++i) m += (double)h[i];
    printf("%-70s %.1f ticks per MFMA\n", what, m / grid / (iters * 32.0));
}

int main() {
    h8* in; float* out; long long* cyc;
    hipMalloc(&in, 512 * sizeof(h8)); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    hipMemset(in, 0, 512 * sizeof(h8));
    run<false, false, 3, false>(in, out, cyc, "A register, B one fragment");
    run<true, false, 3, false>(in, out, cyc, "A register, B from a 32-fragment register array");
    run<false, true, 3, false>(in, out, cyc, "A from LDS (ring of 3), B one fragment");
    run<true, true, 3, false>(in, out, cyc, "A from LDS (ring of 3), B array");
    run<true, true, 3, true>(in, out, cyc, "A from LDS (ring of 3), B array, barrier per 32 MFMAs");
    run<true, true, 1, true>(in, out, cyc, "A from LDS (ring of 1), B array, barrier per 32 MFMAs");
    run<true, true, 6, true>(in, out, cyc, "A from LDS (ring of 6), B array, barrier per 32 MFMAs");
    return 0;
}

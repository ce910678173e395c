// What does ONE wave per SIMD pay for an instruction issued between two v_mfma_f32_32x32x16_f16?  Per filler kind, 6 fillers per MFMA.  (round 5)
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_fillers.hip -o tools/micro/bin/mfma_fillers
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

// KIND: 0 none, 1 v_fma_f32, 2 v_exp_f32, 3 v_log_f32, 4 v_accvgpr_read of the OTHER accumulator (+ 1 fma to consume), 5 v_cvt_pk_f16_f32,
//       6 ds_read_b128 (1 per MFMA), 7 v_pk_fma_f32
template <int KIND, int NTHR>
__global__ void __launch_bounds__(NTHR, NTHR / 256) k(const h8* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) h8 smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += NTHR) smem[i] = in[i & 511];
    __syncthreads();
    const h8 a = in[tid & 255], b = in[256 + (tid & 255)];
    v16f acc[2], other[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) { acc[i][j] = (float)(i + j); other[i][j] = (float)(3 * i + j); }
    // `other` goes through one MFMA so that it lives where accumulators live
    other[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, other[0], 0, 0, 0);
    other[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, other[1], 0, 0, 0);
    float f[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) f[j] = (float)a[j] + 1.5f;
    h8 ld = a;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16; ++rep) {
            acc[rep & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(KIND == 6 ? ld : a, b, acc[rep & 1], 0, 0, 0);
            if (KIND == 6) ld = smem[((rep + it) & 31) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if (KIND == 1) f[j] = __builtin_fmaf(f[j], 1.0001f, 0.5f);
                if (KIND == 2) f[j] = __builtin_amdgcn_exp2f(f[j]);
                if (KIND == 3) f[j] = __builtin_amdgcn_logf(f[j]);
                if (KIND == 4) f[j] = __builtin_fmaf(other[(rep >> 1) & 1][(2 * j + rep) & 15], 0.5f, f[j]);
                if (KIND == 5) { const h2 q = __builtin_convertvector(f2{f[j], f[(j + 1) % 6]}, h2); f[j] = (float)q.x + (float)q.y; }
                if (KIND == 7 && j < 3) { const f2 r = __builtin_elementwise_fma(f2{f[2 * j], f[2 * j + 1]}, f2{1.0001f, 1.0001f}, f2{0.5f, 0.5f}); f[2 * j] = r.x; f[2 * j + 1] = r.y; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j] + other[i][j];
#pragma unroll
    for (int j = 0; j < 6; ++j) s += f[j];
    s += (float)ld[0];
    if (tid < 256) out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int NTHR>
void run(const h8* in, float* out, long long* cyc, const char* what) {
    const int iters = 400, grid = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<KIND, NTHR>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<KIND, NTHR>), dim3(grid), dim3(NTHR), 100 * 1024, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND, NTHR>), dim3(grid), dim3(NTHR), 100 * 1024, 0, in, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < grid; ++i) m += (double)h[i];
    printf("%d waves per SIMD, %-60s %.1f ticks per MFMA of one wave = %.1f per MFMA of the SIMD; kernel %.1f us = %.2f ns per MFMA of the SIMD (peak rate: 13.4 ns)\n", NTHR / 256, what, m / grid / (iters * 16.0), m / grid / (iters * 16.0) / (NTHR / 256), ms * 1e3, ms * 1e6 / (iters * 16.0 * (NTHR / 256)));
}

int main() {
    h8* in; float* out; long long* cyc;
    hipMalloc(&in, 512 * sizeof(h8)); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    hipMemset(in, 0, 512 * sizeof(h8));
    run<0, 256>(in, out, cyc, "no filler");
    run<1, 256>(in, out, cyc, "6 x v_fma_f32");
    run<2, 256>(in, out, cyc, "6 x v_exp_f32");
    run<6, 256>(in, out, cyc, "1 x ds_read_b128 one step ahead");
    run<0, 512>(in, out, cyc, "no filler");
    run<1, 512>(in, out, cyc, "6 x v_fma_f32");
    run<2, 512>(in, out, cyc, "6 x v_exp_f32");
    run<6, 512>(in, out, cyc, "1 x ds_read_b128 one step ahead");
    run<0, 1024>(in, out, cyc, "no filler");
    run<1, 1024>(in, out, cyc, "6 x v_fma_f32");
    run<2, 1024>(in, out, cyc, "6 x v_exp_f32");
    run<6, 1024>(in, out, cyc, "1 x ds_read_b128 one step ahead");
    return 0;
}

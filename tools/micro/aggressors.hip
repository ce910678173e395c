// Synthetic side-queue loads, one instruction class each (tools/raster_race_probe8.py): which property of k_h2_fwd / k_h2_bwd makes a co-resident kernel's
// packed-fp32 arithmetic go wrong?  Every kernel: 256 lanes per block, `rounds` iterations of an 8-instruction body, results folded into a sink nobody reads.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/aggressors.hip -o gshell_amd/lib/variants/aggressors.so
#include <hip/hip_runtime.h>
#include <cstdint>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

enum { A_PK = 0, A_SDWA = 1, A_MFMA = 2, A_MFMA_PK = 3, A_TRANS = 4, A_LDS = 5, A_CVT_PK = 6, A_MIX = 7, A_MFMA_SDWA = 8, A_N = 9 };

template <int MODE>
__global__ void __launch_bounds__(256) k_aggr(float* __restrict__ sink, int rounds, float seed) {
    __shared__ float lds[10240];      // 40 KB: four blocks = 16 waves per CU, half of the wave slots stay free for the other queue
    const int tid = threadIdx.x;
    float x0 = seed + tid * 1e-3f, x1 = x0 * 1.5f, x2 = x0 - 0.25f, x3 = x1 + 0.125f;
    v2f p = {x0, x1}, q = {x2, x3};
    v16f acc = {};
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(x0 + i); hb[i] = (_Float16)(x1 - i); }
    uint32_t u = __float_as_uint(x0), w = __float_as_uint(x1);
    lds[tid] = x0; lds[tid + 256] = x1;
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        if (MODE == A_PK || MODE == A_MFMA_PK) {
            asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]\n\tv_pk_add_f32 %1, %1, %0 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                         "v_pk_fma_f32 %1, %0, %1, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_mul_f32 %1, %1, %0 op_sel_hi:[0,1]\n\t" : "+v"(p), "+v"(q));
        }
        if (MODE == A_SDWA || MODE == A_MFMA_SDWA) {
            asm volatile("v_cvt_f32_f16_sdwa %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\tv_cvt_f32_f16_sdwa %1, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
                         "v_add_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\tv_cvt_f16_f32_sdwa %2, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
                         "v_cvt_f32_f16_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\t" : "+v"(x0), "+v"(x1), "+v"(u), "+v"(w));
        }
        if (MODE == A_MFMA || MODE == A_MFMA_PK || MODE == A_MFMA_SDWA) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc, 0, 0, 0);
        }
        if (MODE == A_TRANS) {
            asm volatile("v_exp_f32 %0, %0\n\tv_log_f32 %1, %1\n\tv_rcp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_sqrt_f32 %0, %0\n\tv_log_f32 %1, %1\n\t" : "+v"(x0), "+v"(x1));
        }
        if (MODE == A_LDS) {
            float a = lds[(tid * 5 + r) & 4095], b = lds[(tid * 9 + r * 3) & 4095];
            lds[(tid + r * 17) & 4095] = a + b;
            x0 += a - b;
        }
        if (MODE == A_CVT_PK) {
            asm volatile("v_cvt_pk_f16_f32 %2, %0, %1\n\tv_cvt_f32_f16 %0, %2\n\tv_cvt_pk_bf16_f32 %3, %0, %1\n\tv_fma_mixlo_f16 %2, %0, %1, %0\n\tv_cvt_f32_f16 %1, %2\n\t" : "+v"(x0), "+v"(x1), "+v"(u), "+v"(w));
        }
        if (MODE == A_MIX) {      // what the chain kernels' epilogues look like: trans + sdwa converts + packed + LDS writes of halves
            asm volatile("v_exp_f32 %0, %0\n\tv_cvt_pk_f16_f32 %2, %0, %1\n\tv_cvt_f32_f16_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t" : "+v"(x0), "+v"(x1), "+v"(u));
            asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]\n\tv_pk_add_f32 %1, %1, %0\n\t" : "+v"(p), "+v"(q));
            ((_Float16*)lds)[(tid * 3 + r) & 8191] = (_Float16)x1;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
        }
    }
    float s = x0 + x1 + p[0] + p[1] + q[0] + q[1] + __uint_as_float(u) + __uint_as_float(w) + lds[(tid + 1) & 4095];
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 1.2345e-30f) sink[0] = s;
}

extern "C" int aggr_launch(int mode, int blocks, int rounds, void* sink, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (mode) {
#define CASE(M) case M: hipLaunchKernelGGL(k_aggr<M>, dim3(blocks), dim3(256), 0, st, (float*)sink, rounds, 0.37f); break;
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
        default: return -1;
    }
    return (int)hipGetLastError();
}

// ---- the "epilogue mix" taken apart: bit 0 v_exp_f32, 1 v_cvt_pk_f16_f32, 2 v_cvt_f32_f16_sdwa, 3 packed fp32, 4 16-bit LDS write, 5 MFMA
template <int MASK>
__global__ void __launch_bounds__(256) k_mix(float* __restrict__ sink, int rounds, float seed) {
    __shared__ float lds[10240];
    const int tid = threadIdx.x;
    float x0 = seed + tid * 1e-3f, x1 = x0 * 1.5f;
    v2f p = {x0, x1}, q = {x0 - 0.25f, x1 + 0.125f};
    v16f acc = {};
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(x0 + i); hb[i] = (_Float16)(x1 - i); }
    uint32_t u = __float_as_uint(x0);
    lds[tid] = x0;
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        if (MASK & 1) asm volatile("v_exp_f32 %0, %0\n\t" : "+v"(x0));
        if (MASK & 2) asm volatile("v_cvt_pk_f16_f32 %2, %0, %1\n\t" : "+v"(x0), "+v"(x1), "+v"(u));
        if (MASK & 4) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t" : "+v"(x1), "+v"(u));
        if (MASK & 8) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]\n\tv_pk_add_f32 %1, %1, %0\n\t" : "+v"(p), "+v"(q));
        if (MASK & 16) ((_Float16*)lds)[(tid * 3 + r) & 8191] = (_Float16)x1;
        if (MASK & 32) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
    }
    float s = x0 + x1 + p[0] + p[1] + q[0] + q[1] + __uint_as_float(u) + lds[(tid + 1) & 4095];
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 1.2345e-30f) sink[0] = s;
}

template <int M>
static void mix_dispatch(int mask, int blocks, int rounds, float* sink, hipStream_t st) {
    if (mask == M) hipLaunchKernelGGL(k_mix<M>, dim3(blocks), dim3(256), 0, st, sink, rounds, 0.37f);
    else if constexpr (M > 0) mix_dispatch<M - 1>(mask, blocks, rounds, sink, st);
}

extern "C" int mix_launch(int mask, int blocks, int rounds, void* sink, void* stream) {
    if (mask < 0 || mask > 63) return -1;
    mix_dispatch<63>(mask, blocks, rounds, (float*)sink, (hipStream_t)stream);
    return (int)hipGetLastError();
}

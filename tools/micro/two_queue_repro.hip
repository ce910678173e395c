// Stand-alone reproducer of round 6's two-queue finding, independent of the library: a VICTIM kernel whose lanes evaluate
//     d = x * y - z * w
// twice -- once with packed-fp32 instructions (v_pk_mul_f32 + v_pk_add_f32, as hipcc's SLP vectoriser emits them) and once with scalar-per-lane
// v_mul_f32 / v_sub_f32 -- and count bitwise disagreements, while an AGGRESSOR kernel (a loop of two packed-fp32 instructions and one dependent
// v_mfma_f32_32x32x16_f16) runs on a second stream.  Both halves are IEEE fp32 with the same roundings: any disagreement is a wrong result.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/two_queue_repro.hip -o gpurun_out/two_queue_repro && gpurun_out/two_queue_repro
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

// -DSAME_VGPRS: both kernels of the two-stream configurations touch v255, i.e. get the same (maximal) architectural register allocation -- is it the MIX of
// allocation sizes on a SIMD that matters?
#define REGS
#ifdef SAME_VGPRS
#define PAD_REGS() asm volatile("v_mov_b32 v255, 0" ::: "v255")
#else
#define PAD_REGS()
#endif

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ float unit(uint32_t u) { return __uint_as_float(0x3f000000u | (u >> 9)) ; }      // [0.5, 1)

// VICTIM.  counts[0] = disagreements of the packed evaluation with the per-lane one, counts[1] = of a SECOND per-lane evaluation with the first (control)
template <bool PACKED, bool DIVERGENT = false>
__global__ void __launch_bounds__(256) k_victim(int rounds, uint32_t seed, unsigned long long* __restrict__ counts) {
    uint32_t s = seed ^ ((blockIdx.x * 256u + threadIdx.x) * 0x9e3779b9u);
    unsigned bad = 0, bad_ctl = 0;
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        float x = unit(lcg(s)), y = unit(lcg(s)) - 0.75f, z = unit(lcg(s)), w = unit(lcg(s)) - 0.75f;
        float m0, m1, d_ref, d_ctl;
        asm volatile("v_mul_f32 %0, %2, %3\n\tv_mul_f32 %1, %4, %5" : "=&v"(m0), "=&v"(m1) : "v"(x), "v"(y), "v"(z), "v"(w));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d_ref) : "v"(m0), "v"(m1));
        asm volatile("v_mul_f32 %0, %2, %3\n\tv_mul_f32 %1, %4, %5" : "=&v"(m0), "=&v"(m1) : "v"(x), "v"(y), "v"(z), "v"(w));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d_ctl) : "v"(m0), "v"(m1));
        bad_ctl += __float_as_uint(d_ctl) != __float_as_uint(d_ref);
        if (PACKED && (!DIVERGENT || (lcg(s) & 0x30000u) == 0)) {          // DIVERGENT: a quarter of the lanes, chosen per lane and round (partial EXEC, like a rasteriser's coverage test)
            v2f a = {x, z}, b = {y, w}, p, q;
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(a), "v"(b));                                              // (x y, z w)
            asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(q) : "v"(p));   // lo = p.lo - p.hi
            bad += __float_as_uint(q[0]) != __float_as_uint(d_ref);
        }
    }
    if (bad) atomicAdd(&counts[0], (unsigned long long)bad);
    if (bad_ctl) atomicAdd(&counts[1], (unsigned long long)bad_ctl);
}

// VICTIM 2: the rasteriser's per-sample arithmetic as hipcc compiles it (csrc/raster.hip bary_eval: the SLP vectoriser packs it into v_pk_mul_f32 / v_pk_add_f32
// with op_sel / neg modifiers), evaluated twice on the same operands behind asm barriers; counts[0] = samples whose two evaluations differ in any bit.
struct Bary { float a0, a1, a2, s, zw; };
__device__ __forceinline__ Bary bary_eval(const float4 p0, const float4 p1, const float4 p2, float fx, float fy) {
    float p0x = p0.x - fx * p0.w, p0y = p0.y - fy * p0.w;
    float p1x = p1.x - fx * p1.w, p1y = p1.y - fy * p1.w;
    float p2x = p2.x - fx * p2.w, p2y = p2.y - fy * p2.w;
    Bary r;
    r.a0 = p1x * p2y - p1y * p2x;
    r.a1 = p2x * p0y - p2y * p0x;
    r.a2 = p0x * p1y - p0y * p1x;
    r.s = r.a0 + r.a1 + r.a2;
    float z = p0.z * r.a0 + p1.z * r.a1 + p2.z * r.a2;
    float w = p0.w * r.a0 + p1.w * r.a1 + p2.w * r.a2;
    r.zw = z / w;
    return r;
}
__global__ void __launch_bounds__(256) REGS k_victim_bary(int rounds, uint32_t seed, unsigned long long* __restrict__ counts) {
    uint32_t s = seed ^ ((blockIdx.x * 256u + threadIdx.x) * 0x9e3779b9u);
    float4 p0 = make_float4(unit(lcg(s)) - 0.75f, unit(lcg(s)) - 0.75f, unit(lcg(s)) + 1.0f, unit(lcg(s)) + 1.5f);
    float4 p1 = make_float4(p0.x + 0.01f * unit(lcg(s)), p0.y - 0.004f * unit(lcg(s)), p0.z + 0.01f, p0.w + 0.012f);
    float4 p2 = make_float4(p0.x - 0.003f * unit(lcg(s)), p0.y + 0.009f * unit(lcg(s)), p0.z - 0.008f, p0.w - 0.01f);
    unsigned bad = 0;
    PAD_REGS();
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        if ((lcg(s) & 0x30000u) != 0) continue;                      // a quarter of the lanes per round (coverage-test divergence)
        const float fx = (p0.x / p0.w) + 0.002f * (unit(lcg(s)) - 0.75f), fy = (p0.y / p0.w) + 0.002f * (unit(lcg(s)) - 0.75f);
        Bary r1 = bary_eval(p0, p1, p2, fx, fy);
        float4 q0 = p0, q1 = p1, q2 = p2;
        float gx = fx, gy = fy;
        asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w));
        asm volatile("" : "+v"(q2.x), "+v"(q2.y), "+v"(q2.z), "+v"(q2.w), "+v"(gx), "+v"(gy));
        Bary r2 = bary_eval(q0, q1, q2, gx, gy);
        bad += (__float_as_uint(r1.a0) != __float_as_uint(r2.a0)) | (__float_as_uint(r1.a1) != __float_as_uint(r2.a1)) | (__float_as_uint(r1.a2) != __float_as_uint(r2.a2)) |
               (__float_as_uint(r1.zw) != __float_as_uint(r2.zw));
    }
    if (bad) atomicAdd(&counts[0], (unsigned long long)bad);
}

// AGGRESSOR: mode bit 0 = two packed-fp32 instructions per round, bit 1 = one dependent MFMA per round
template <int MODE>
__global__ void __launch_bounds__(256) REGS k_aggressor(float* __restrict__ sink, int rounds) {
    __shared__ float lds[10240];          // 40 KB: four blocks per CU, half of the wave slots stay free for the other queue
    const int tid = threadIdx.x;
    float x0 = 0.37f + tid * 1e-3f, x1 = x0 * 1.5f;
    v2f p = {x0, x1}, q = {x0 - 0.25f, x1 + 0.125f};
    v16f acc = {};
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(x0 + i); hb[i] = (_Float16)(x1 - i); }
    lds[tid] = x0;
    PAD_REGS();
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        if (MODE & 1) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]\n\tv_pk_add_f32 %1, %1, %0\n\t" : "+v"(p), "+v"(q));
        if (MODE & 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
    }
    float s = p[0] + p[1] + q[0] + q[1] + lds[(tid + 1) & 4095];
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 1.2345e-30f) sink[0] = s;
}

// ONE kernel, ONE queue: even blocks play the aggressor (40 KB of LDS each, like k_aggressor), odd blocks the bary_eval victim -- is a second QUEUE needed, or only
// co-resident waves of the two kinds?
__global__ void __launch_bounds__(256) k_both(float* __restrict__ sink, int aggr_rounds, int rounds, uint32_t seed, unsigned long long* __restrict__ counts) {
    __shared__ float lds[10240];
    if ((blockIdx.x & 1) == 0) {
        const int tid = threadIdx.x;
        float x0 = 0.37f + tid * 1e-3f, x1 = x0 * 1.5f;
        v2f p = {x0, x1}, q = {x0 - 0.25f, x1 + 0.125f};
        v16f acc = {};
        h8 ha, hb;
        for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(x0 + i); hb[i] = (_Float16)(x1 - i); }
        lds[tid] = x0;
        __syncthreads();
#pragma unroll 1
        for (int r = 0; r < aggr_rounds; ++r) {
            asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]\n\tv_pk_add_f32 %1, %1, %0\n\t" : "+v"(p), "+v"(q));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
        }
        float s = p[0] + p[1] + q[0] + q[1] + lds[(tid + 1) & 4095];
        for (int i = 0; i < 16; ++i) s += acc[i];
        if (s == 1.2345e-30f) sink[0] = s;
        return;
    }
    uint32_t s = seed ^ ((blockIdx.x * 256u + threadIdx.x) * 0x9e3779b9u);
    float4 p0 = make_float4(unit(lcg(s)) - 0.75f, unit(lcg(s)) - 0.75f, unit(lcg(s)) + 1.0f, unit(lcg(s)) + 1.5f);
    float4 p1 = make_float4(p0.x + 0.01f * unit(lcg(s)), p0.y - 0.004f * unit(lcg(s)), p0.z + 0.01f, p0.w + 0.012f);
    float4 p2 = make_float4(p0.x - 0.003f * unit(lcg(s)), p0.y + 0.009f * unit(lcg(s)), p0.z - 0.008f, p0.w - 0.01f);
    unsigned bad = 0;
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        if ((lcg(s) & 0x30000u) != 0) continue;
        const float fx = (p0.x / p0.w) + 0.002f * (unit(lcg(s)) - 0.75f), fy = (p0.y / p0.w) + 0.002f * (unit(lcg(s)) - 0.75f);
        Bary r1 = bary_eval(p0, p1, p2, fx, fy);
        float4 q0 = p0, q1 = p1, q2 = p2;
        float gx = fx, gy = fy;
        asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w));
        asm volatile("" : "+v"(q2.x), "+v"(q2.y), "+v"(q2.z), "+v"(q2.w), "+v"(gx), "+v"(gy));
        Bary r2 = bary_eval(q0, q1, q2, gx, gy);
        bad += (__float_as_uint(r1.a0) != __float_as_uint(r2.a0)) | (__float_as_uint(r1.a1) != __float_as_uint(r2.a1)) | (__float_as_uint(r1.a2) != __float_as_uint(r2.a2)) |
               (__float_as_uint(r1.zw) != __float_as_uint(r2.zw));
    }
    if (bad) atomicAdd(&counts[0], (unsigned long long)bad);
}

static int run_both(const char* label, int iters, hipStream_t s1, unsigned long long* counts, float* sink) {
    CHECK(hipMemsetAsync(counts, 0, 16, s1));
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(k_both, dim3(4096), dim3(256), 0, s1, sink, 1500, 450, 777u + it, counts);
        CHECK(hipStreamSynchronize(s1));
    }
    unsigned long long h[2];
    CHECK(hipMemcpy(h, counts, 16, hipMemcpyDeviceToHost));
    printf("  %-78s samples whose evaluations differ: %llu   (of %.3g samples)\n", label, h[0], (double)iters * 2048 * 256 * 450 * 0.25);
    return 0;
}

template <bool PACKED, bool DIVERGENT = false, bool BARY = false>
static int run(const char* label, int aggr_mode, int iters, hipStream_t s1, hipStream_t s2, unsigned long long* counts, float* sink) {
    CHECK(hipMemsetAsync(counts, 0, 16, s1));
    CHECK(hipStreamSynchronize(s1));
    for (int it = 0; it < iters; ++it) {
        if (aggr_mode == 1) hipLaunchKernelGGL(k_aggressor<1>, dim3(1024), dim3(256), 0, s2, sink, 6000);
        if (aggr_mode == 2) hipLaunchKernelGGL(k_aggressor<2>, dim3(1024), dim3(256), 0, s2, sink, 1500);
        if (aggr_mode == 3) hipLaunchKernelGGL(k_aggressor<3>, dim3(1024), dim3(256), 0, s2, sink, 1500);
        for (int k = 0; k < 4; ++k) {
            if (BARY) hipLaunchKernelGGL(k_victim_bary, dim3(3530), dim3(256), 0, s1, 64, 1234u + it * 4 + k, counts);
            else hipLaunchKernelGGL((k_victim<PACKED, DIVERGENT>), dim3(3530), dim3(256), 0, s1, 64, 1234u + it * 4 + k, counts);
        }
        CHECK(hipDeviceSynchronize());
    }
    unsigned long long h[2];
    CHECK(hipMemcpy(h, counts, 16, hipMemcpyDeviceToHost));
    const double evals = (double)iters * 4 * 3530 * 256 * 64;
    printf("  %-78s packed != per-lane: %llu, per-lane repeat != per-lane: %llu   (of %.3g evaluations)\n", label, h[0], h[1], evals);
    return 0;
}

int main(int argc, char** argv) {
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    unsigned long long* counts;
    float* sink;
    CHECK(hipMalloc(&counts, 16));
    CHECK(hipMalloc(&sink, 64));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("%s (%s), %d CUs\n", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    if (run<true>("victim with packed fp32, second queue idle", 0, iters, s1, s2, counts, sink)) return 1;
    if (run<true>("victim with packed fp32, second queue: packed fp32 only", 1, iters, s1, s2, counts, sink)) return 1;
    if (run<true>("victim with packed fp32, second queue: MFMA only", 2, iters, s1, s2, counts, sink)) return 1;
    if (run<true>("victim with packed fp32, second queue: packed fp32 + MFMA", 3, iters, s1, s2, counts, sink)) return 1;
    if (run<false>("victim WITHOUT packed fp32, second queue: packed fp32 + MFMA", 3, iters, s1, s2, counts, sink)) return 1;
    if (run<true, true>("victim with packed fp32 under a divergent branch, second queue idle", 0, iters, s1, s2, counts, sink)) return 1;
    if (run<true, true>("victim with packed fp32 under a divergent branch, second queue: MFMA only", 2, iters, s1, s2, counts, sink)) return 1;
    if (run<true, true>("victim with packed fp32 under a divergent branch, second queue: packed fp32 + MFMA", 3, iters, s1, s2, counts, sink)) return 1;
    if (run<true, true, true>("victim = bary_eval twice (first count = samples whose evaluations differ), second queue idle", 0, iters, s1, s2, counts, sink)) return 1;
    if (run<true, true, true>("victim = bary_eval twice, second queue: MFMA only", 2, iters, s1, s2, counts, sink)) return 1;
    if (run<true, true, true>("victim = bary_eval twice, second queue: packed fp32 + MFMA", 3, iters, s1, s2, counts, sink)) return 1;
    if (run<true>("victim with packed fp32, second queue idle (again)", 0, iters, s1, s2, counts, sink)) return 1;
    if (run_both("ONE kernel on ONE queue: even blocks aggressor, odd blocks bary_eval victim", iters, s1, counts, sink)) return 1;
    return 0;
}

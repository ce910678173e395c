// Host execution shim for the reference's CUDA sources (TEST INFRASTRUCTURE -- checker only, never linked into the product).
//
// oracle/Makefile compiles the reference's own kernels
//     render/optixutils/c_src/envsampling/kernel.cu, c_src/denoising.cu,
//     render/renderutils/c_src/loss.cu, normal.cu, mesh.cu
// FROM /root/reference (nothing is copied into this repository) with g++ for the host cores.  This header supplies what
// nvcc / NVRTC would: the qualifier keywords, the built-in vector types, min / max overloads, the launch-geometry
// variables, __syncthreads / __shfl_xor_sync / atomicAdd, and a launcher that runs a <<<grid, block>>> launch block by
// block with one ucontext fibre per CUDA thread (barriers and warp shuffles need every thread of a block alive at once).
//
// What the shim cannot reproduce: `-use_fast_math` (NVRTC option of the OptiX program, optix_wrapper.cpp:35) -- libm's
// correctly-rounded-ish float functions and IEEE division stand in for the approximate device intrinsics -- and the
// hardware ray / triangle predicate of `optixTrace` (see optix.h in this directory).
#pragma once
#ifndef GS_CUDA_HOST_SHIM_H
#define GS_CUDA_HOST_SHIM_H

#ifndef __CUDACC__
#define __CUDACC__ 1          // the reference guards its device helpers with this
#endif
#define __device__
#define __host__
#define __global__
#define __constant__
#define __shared__ static     // blocks run one after the other: one static copy is one block's shared memory
#define __forceinline__ inline
#define __restrict__

#include <math.h>             // libstdc++'s wrapper: float overloads of sqrt / cos / pow / ... in the global namespace, as in CUDA
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <algorithm>
#include <functional>
#include <vector>

// NVRTC has no <math.h> macro M_PI: the OptiX-side bsdf.h then defines it as a FLOAT (c_src/bsdf.h:14-16).  The nvcc-built
// renderutils sources see glibc's double M_PI but never use it.  GS_SHIM_KEEP_M_PI keeps glibc's.
#ifndef GS_SHIM_KEEP_M_PI
#undef M_PI
#endif

using std::abs;

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct uint3 { unsigned int x, y, z; };
struct dim3 {
    unsigned int x, y, z;
    dim3(unsigned int x_ = 1, unsigned int y_ = 1, unsigned int z_ = 1) : x(x_), y(y_), z(z_) {}
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint3 make_uint3(unsigned int x, unsigned int y, unsigned int z) { return uint3{x, y, z}; }

// CUDA's global min / max overload set (math_functions.hpp): same-type and the mixed forms the sources use.
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }
static inline double min(float a, double b) { return fmin((double)a, b); }
static inline double max(float a, double b) { return fmax((double)a, b); }
static inline double min(double a, float b) { return fmin(a, (double)b); }
static inline double max(double a, float b) { return fmax(a, (double)b); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }
static inline unsigned int max(unsigned int a, unsigned int b) { return a > b ? a : b; }
static inline unsigned int min(int a, unsigned int b) { return min((unsigned int)a, b); }
static inline unsigned int max(int a, unsigned int b) { return max((unsigned int)a, b); }
static inline unsigned int min(unsigned int a, int b) { return min(a, (unsigned int)b); }
static inline unsigned int max(unsigned int a, int b) { return max(a, (unsigned int)b); }

// C++ overload CUDA offers next to sincos(double, double*, double*); kernel.cu:128-131 calls it with a double argument
// and float pointers, which selects this one (argument rounded to float first).
static inline void sincos(float a, float* s, float* c) { *s = sinf(a); *c = cosf(a); }

static inline float atomicAdd(float* addr, float v) {       // relaxed CAS loop: callers may run pixels on several host threads
    uint32_t* p = reinterpret_cast<uint32_t*>(addr);
    uint32_t old = __atomic_load_n(p, __ATOMIC_RELAXED);
    for (;;) {
        float f;
        memcpy(&f, &old, 4);
        float nf = f + v;
        uint32_t nu;
        memcpy(&nu, &nf, 4);
        if (__atomic_compare_exchange_n(p, &old, nu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return f;
    }
}

// ---- launch geometry + fibres ------------------------------------------------------------------------------------------
inline thread_local uint3 threadIdx = {0, 0, 0}, blockIdx = {0, 0, 0};
inline thread_local dim3 blockDim, gridDim;

namespace cuhost {
enum { READY = 0, WAITING = 1, DONE = 2 };
constexpr size_t STACK_BYTES = 256 * 1024;
struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    int state = DONE;
    uint3 tid;
};
struct Block {
    std::vector<Fiber> f;
    ucontext_t sched;
    int cur = 0;
    std::function<void()> body;
    std::vector<float> xchg;
};
inline thread_local Block* g_blk = nullptr;

inline void trampoline() {
    Block* b = g_blk;
    b->body();
    Fiber& me = b->f[b->cur];
    me.state = DONE;
    swapcontext(&me.ctx, &b->sched);
}
inline void barrier() {                                       // block-wide: every live thread of the block must arrive
    Block* b = g_blk;
    Fiber& me = b->f[b->cur];
    me.state = WAITING;
    swapcontext(&me.ctx, &b->sched);
}

// Runs kernel(p) for every thread of every block of a <<<grid, block>>> launch; blocks in x-fastest order, threads of a
// block as fibres in linear-thread-id order (x + y*Dx + z*Dx*Dy: CUDA's warp formation order).
template <class P>
void launch(void (*kernel)(P), dim3 grid, dim3 block, P p) {
    Block b;
    const int n = (int)(block.x * block.y * block.z);
    b.f.resize(n);
    b.xchg.resize(n);
    for (auto& f : b.f) f.stack = (char*)malloc(STACK_BYTES);
    b.body = [&]() { kernel(p); };
    g_blk = &b;
    blockDim = block;
    gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx = uint3{bx, by, bz};
                int i = 0;
                for (unsigned tz = 0; tz < block.z; ++tz)
                    for (unsigned ty = 0; ty < block.y; ++ty)
                        for (unsigned tx = 0; tx < block.x; ++tx, ++i) {
                            Fiber& f = b.f[i];
                            f.tid = uint3{tx, ty, tz};
                            f.state = READY;
                            getcontext(&f.ctx);
                            f.ctx.uc_stack.ss_sp = f.stack;
                            f.ctx.uc_stack.ss_size = STACK_BYTES;
                            f.ctx.uc_link = nullptr;
                            makecontext(&f.ctx, (void (*)())trampoline, 0);
                        }
                for (;;) {
                    bool any = false;
                    for (i = 0; i < n; ++i)
                        if (b.f[i].state == READY) {       // runs until its next barrier or its end
                            any = true;
                            b.cur = i;
                            threadIdx = b.f[i].tid;
                            swapcontext(&b.sched, &b.f[i].ctx);
                        }
                    if (!any) break;
                    for (auto& f : b.f)
                        if (f.state == WAITING) f.state = READY;   // everyone alive has arrived: release
                }
            }
    for (auto& f : b.f) free(f.stack);
    g_blk = nullptr;
}
}  // namespace cuhost

static inline void __syncthreads() { cuhost::barrier(); }

// Warp = 32 consecutive linear thread ids.  The sources shuffle with every thread of the block alive (loss.cu:22-26), so
// the exchange is done under two block-wide barriers; a source lane that has exited returns the caller's own value.
static inline float __shfl_xor_sync(unsigned int, float v, int lane_mask) {
    cuhost::Block* b = cuhost::g_blk;
    const int me = b->cur, n = (int)b->f.size();
    b->xchg[me] = v;
    cuhost::barrier();
    const int src = (me & ~31) | ((me & 31) ^ lane_mask);
    const float r = (src < n && b->f[src].state != cuhost::DONE) ? b->xchg[src] : v;
    cuhost::barrier();
    return r;
}

#endif  // GS_CUDA_HOST_SHIM_H

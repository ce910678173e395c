// Does v_mfma_f32_32x32x16_f16 honour fp16 DENORMAL inputs (or flush them to zero)?   hipcc --offload-arch=gfx950 -O2 -o /tmp/dn tools/micro/mfma_f16_denorm.hip && /tmp/dn
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out, float a_val, float b_val) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    v16f c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float tests[][2] = {{1.0f, 1.0f}, {3.0e-5f, 1024.0f}, {1.0e-6f, 1024.0f}, {6.0e-8f, 4096.0f}, {3.0e-5f, 3.0e-5f}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t[0], t[1]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        float ah = (float)(_Float16)t[0], bh = (float)(_Float16)t[1];
        printf("a=%g (fp16 %g) b=%g: mfma sum over K=16 -> %g   expected %g\n", t[0], ah, t[1], h, 16.0f * ah * bh);
    }
    return 0;
}

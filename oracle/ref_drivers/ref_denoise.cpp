// Host driver for the reference's bilateral denoiser kernels (TEST INFRASTRUCTURE -- checker only).
// Compiles /root/reference/render/optixutils/c_src/denoising.cu unchanged through oracle/ref_stub; launch geometry and
// parameter set-up mirror bilateral_denoiser_fwd / _bwd (c_src/torch_bindings.cpp:268-314): 8x8x1 blocks, zero-filled outputs.
#include <cuda.h>
#include <denoising.cu>

#include <initializer_list>

namespace {
template <class T, size_t N>
struct AccMaker : PackedTensorAccessor32<T, N> {
    AccMaker(T* d, std::initializer_list<int> sizes) {
        this->data_ = d;
        int i = 0;
        for (int s : sizes) this->sizes_[i++] = s;
        int st = 1;
        for (int k = (int)N - 1; k >= 0; --k) { this->strides_[k] = st; st *= this->sizes_[k]; }
    }
};
PackedTensorAccessor32<float, 4> acc4(const float* d, int a, int b, int c, int e) {
    AccMaker<float, 4> m(const_cast<float*>(d), {a, b, c, e});
    return m;
}
}  // namespace

extern "C" {

int ref_bilateral_fwd(const float* col, const float* nrm, const float* zdz, int B, int H, int W, float sigma, float* out) {
    memset(out, 0, sizeof(float) * (size_t)B * H * W * 4);
    dim3 blockSize(8, 8, 1);
    dim3 gridSize((W - 1) / blockSize.x + 1, (H - 1) / blockSize.y + 1, (B - 1) / blockSize.z + 1);
    BilateralDenoiserParams p;
    p.col = acc4(col, B, H, W, 3);
    p.nrm = acc4(nrm, B, H, W, 3);
    p.zdz = acc4(zdz, B, H, W, 2);
    p.out = acc4(out, B, H, W, 4);
    p.sigma = sigma;
    cuhost::launch(bilateral_denoiser_fwd_kernel, gridSize, blockSize, p);
    return 0;
}

// out_grad has C channels (the python side hands the gradient of the [B,H,W,4] output; the kernel reads the first three)
int ref_bilateral_bwd(const float* col, const float* nrm, const float* zdz, const float* out_grad, int B, int H, int W, int C, float sigma,
                      float* col_grad) {
    memset(col_grad, 0, sizeof(float) * (size_t)B * H * W * 3);
    dim3 blockSize(8, 8, 1);
    dim3 gridSize((W - 1) / blockSize.x + 1, (H - 1) / blockSize.y + 1, (B - 1) / blockSize.z + 1);
    BilateralDenoiserParams p;
    p.col = acc4(col, B, H, W, 3);
    p.nrm = acc4(nrm, B, H, W, 3);
    p.zdz = acc4(zdz, B, H, W, 2);
    p.out_grad = acc4(out_grad, B, H, W, C);
    p.col_grad = acc4(col_grad, B, H, W, 3);
    p.sigma = sigma;
    cuhost::launch(bilateral_denoiser_bwd_kernel, gridSize, blockSize, p);
    return 0;
}

}  // extern "C"

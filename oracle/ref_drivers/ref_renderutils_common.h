// Shared by the renderutils drivers (TEST INFRASTRUCTURE): the `Tensor` set-up of make_cuda_tensor
// (render/renderutils/c_src/torch_bindings.cpp:116-155) for contiguous fp32 arrays, and the launch-size helpers compiled
// from the reference's own common.cpp (included by ONE driver; declared in c_src/common.h).
#pragma once
#include <cuda.h>
#include <common.h>

#define BLOCK_X 8   // torch_bindings.cpp:40-41
#define BLOCK_Y 8

// dims [n] (n = 3 or 4), contiguous strides; outDims = launch grid (x = W, y = H, z = B)
static inline Tensor make_tensor(const float* val, const int* dims, int n, dim3 outDims, float* grad = nullptr) {
    Tensor res;
    memset(&res, 0, sizeof(res));
    int st = 1;
    for (int i = n - 1; i >= 0; --i) { res.dims[i] = dims[i]; res.strides[i] = st; st *= dims[i]; }
    if (n == 4)
        res._dims[0] = outDims.z, res._dims[1] = outDims.y, res._dims[2] = outDims.x, res._dims[3] = dims[3];
    else
        res._dims[0] = outDims.z, res._dims[1] = outDims.x, res._dims[2] = dims[2], res._dims[3] = 1;
    res.fp16 = false;
    res.val = (void*)val;
    res.d_val = (void*)grad;
    return res;
}

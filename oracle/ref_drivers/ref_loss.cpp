// Host driver for the reference's image-loss kernels (TEST INFRASTRUCTURE -- checker only).
// Compiles /root/reference/render/renderutils/c_src/loss.cu unchanged through oracle/ref_stub; set-up mirrors
// image_loss_fwd / image_loss_bwd (c_src/torch_bindings.cpp:907-966).
#define GS_SHIM_KEEP_M_PI
#include "ref_renderutils_common.h"
#include <loss.cu>

dim3 getLaunchBlockSize(int maxWidth, int maxHeight, dim3 dims);   // common.cpp (ref_common.cpp)
dim3 getLaunchGridSize(dim3 blockSize, dim3 dims);

extern "C" {

// shape of the per-warp partial sums (torch_bindings.cpp:922-928)
void ref_image_loss_out_dims(int B, int H, int W, int* dims) {
    dim3 grid(W, H, B);
    dim3 blockSize = getLaunchBlockSize(BLOCK_X, BLOCK_Y, grid);
    dim3 warpSize = getWarpSize(blockSize);
    dims[0] = (grid.z - 1) / warpSize.z + 1;
    dims[1] = (grid.y - 1) / warpSize.y + 1;
    dims[2] = (grid.x - 1) / warpSize.x + 1;
}

int ref_image_loss_fwd(const float* img, const float* target, int B, int H, int W, int loss, int tonemapper, float* out) {
    LossKernelParams p;
    memset(&p, 0, sizeof(p));
    p.loss = (LossType)loss;
    p.tonemapper = (TonemapperType)tonemapper;
    p.gridSize = dim3(W, H, B);
    dim3 blockSize = getLaunchBlockSize(BLOCK_X, BLOCK_Y, p.gridSize);
    dim3 warpSize = getWarpSize(blockSize);
    dim3 gridSize = getLaunchGridSize(blockSize, p.gridSize);
    int d[4] = {B, H, W, 3};
    int od[4] = {(int)((p.gridSize.z - 1) / warpSize.z + 1), (int)((p.gridSize.y - 1) / warpSize.y + 1), (int)((p.gridSize.x - 1) / warpSize.x + 1), 1};
    p.img = make_tensor(img, d, 4, p.gridSize);
    p.target = make_tensor(target, d, 4, p.gridSize);
    p.out = make_tensor(out, od, 4, p.gridSize);
    cuhost::launch(imgLossFwdKernel, gridSize, blockSize, p);
    return 0;
}

int ref_image_loss_bwd(const float* img, const float* target, const float* grad, int B, int H, int W, int loss, int tonemapper,
                       float* img_grad, float* target_grad) {
    LossKernelParams p;
    memset(&p, 0, sizeof(p));
    p.loss = (LossType)loss;
    p.tonemapper = (TonemapperType)tonemapper;
    p.gridSize = dim3(W, H, B);
    dim3 blockSize = getLaunchBlockSize(BLOCK_X, BLOCK_Y, p.gridSize);
    dim3 warpSize = getWarpSize(blockSize);
    dim3 gridSize = getLaunchGridSize(blockSize, p.gridSize);
    int d[4] = {B, H, W, 3};
    int od[4] = {(int)((p.gridSize.z - 1) / warpSize.z + 1), (int)((p.gridSize.y - 1) / warpSize.y + 1), (int)((p.gridSize.x - 1) / warpSize.x + 1), 1};
    p.img = make_tensor(img, d, 4, p.gridSize, img_grad);
    p.target = make_tensor(target, d, 4, p.gridSize, target_grad);
    p.out = make_tensor(grad, od, 4, p.gridSize);
    cuhost::launch(imgLossBwdKernel, gridSize, blockSize, p);
    return 0;
}

}  // extern "C"

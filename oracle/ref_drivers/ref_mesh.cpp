// Host driver for the reference's xfm_points kernels (TEST INFRASTRUCTURE -- checker only).
// Compiles /root/reference/render/renderutils/c_src/mesh.cu unchanged (its __shared__ matrix + __syncthreads run on the
// fibre launcher); set-up mirrors xfm_fwd / xfm_bwd (c_src/torch_bindings.cpp:970-1032) with isPoints = true.
#define GS_SHIM_KEEP_M_PI
#include "ref_renderutils_common.h"
#include <mesh.cu>

dim3 getLaunchGridSize(dim3 blockSize, dim3 dims);   // common.cpp, ref_common.cpp

extern "C" {

int ref_xfm_points_fwd(const float* points, int pB, int V, const float* matrix, int B, float* out) {
    XfmKernelParams p;
    memset(&p, 0, sizeof(p));
    p.isPoints = true;
    p.gridSize.x = V;
    p.gridSize.y = 1;
    p.gridSize.z = max(B, pB);
    dim3 blockSize(BLOCK_X * BLOCK_Y, 1, 1);
    dim3 gridSize = getLaunchGridSize(blockSize, p.gridSize);
    int dp[3] = {pB, V, 3}, dm[3] = {B, 4, 4}, dout[3] = {B, V, 4};
    p.points = make_tensor(points, dp, 3, p.gridSize);
    p.matrix = make_tensor(matrix, dm, 3, p.gridSize);
    p.out = make_tensor(out, dout, 3, p.gridSize);
    cuhost::launch(xfmPointsFwdKernel, gridSize, blockSize, p);
    return 0;
}

int ref_xfm_points_bwd(const float* points, int pB, int V, const float* matrix, int B, const float* grad, float* points_grad) {
    XfmKernelParams p;
    memset(&p, 0, sizeof(p));
    p.isPoints = true;
    p.gridSize.x = V;
    p.gridSize.y = 1;
    p.gridSize.z = max(B, pB);
    dim3 blockSize(BLOCK_X * BLOCK_Y, 1, 1);
    dim3 gridSize = getLaunchGridSize(blockSize, p.gridSize);
    int dp[3] = {pB, V, 3}, dm[3] = {B, 4, 4}, dout[3] = {B, V, 4};
    p.points = make_tensor(points, dp, 3, p.gridSize, points_grad);
    p.matrix = make_tensor(matrix, dm, 3, p.gridSize);
    p.out = make_tensor(grad, dout, 3, p.gridSize);
    cuhost::launch(xfmPointsBwdKernel, gridSize, blockSize, p);
    return 0;
}

}  // extern "C"

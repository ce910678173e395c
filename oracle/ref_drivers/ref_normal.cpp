// Host driver for the reference's shading-normal kernels (TEST INFRASTRUCTURE -- checker only).
// Compiles /root/reference/render/renderutils/c_src/normal.cu unchanged; set-up mirrors prepare_shading_normal_fwd / _bwd
// (c_src/torch_bindings.cpp:161-232).  Every input is [b,h,w,3] with b/h/w either 1 (broadcast) or the launch size.
#define GS_SHIM_KEEP_M_PI
#include "ref_renderutils_common.h"
#include <normal.cu>

dim3 getLaunchBlockSize(int maxWidth, int maxHeight, dim3 dims);   // common.cpp, ref_common.cpp
dim3 getLaunchGridSize(dim3 blockSize, dim3 dims);

namespace {
dim3 grid_of(const int* const dims[6]) {
    dim3 g(0, 0, 0);                                  // update_grid (torch_bindings.cpp:100-114); dim3 there starts at 1,1,1
    g = dim3(1, 1, 1);
    for (int i = 0; i < 6; ++i) {
        g.x = max(g.x, (unsigned)dims[i][2]);
        g.y = max(g.y, (unsigned)dims[i][1]);
        g.z = max(g.z, (unsigned)dims[i][0]);
    }
    return g;
}
}  // namespace

extern "C" {

int ref_shading_normal_fwd(const float* pos, const int* d0, const float* view_pos, const int* d1, const float* perturbed_nrm, const int* d2,
                           const float* smooth_nrm, const int* d3, const float* smooth_tng, const int* d4, const float* geom_nrm, const int* d5,
                           int two_sided, int opengl, float* out) {
    const int* const dims[6] = {d0, d1, d2, d3, d4, d5};
    PrepareShadingNormalKernelParams p;
    memset(&p, 0, sizeof(p));
    p.two_sided_shading = two_sided != 0;
    p.opengl = opengl != 0;
    p.gridSize = grid_of(dims);
    dim3 blockSize = getLaunchBlockSize(BLOCK_X, BLOCK_Y, p.gridSize);
    dim3 gridSize = getLaunchGridSize(blockSize, p.gridSize);
    int od[4] = {(int)p.gridSize.z, (int)p.gridSize.y, (int)p.gridSize.x, 3};
    p.pos = make_tensor(pos, d0, 4, p.gridSize);
    p.view_pos = make_tensor(view_pos, d1, 4, p.gridSize);
    p.perturbed_nrm = make_tensor(perturbed_nrm, d2, 4, p.gridSize);
    p.smooth_nrm = make_tensor(smooth_nrm, d3, 4, p.gridSize);
    p.smooth_tng = make_tensor(smooth_tng, d4, 4, p.gridSize);
    p.geom_nrm = make_tensor(geom_nrm, d5, 4, p.gridSize);
    p.out = make_tensor(out, od, 4, p.gridSize);
    cuhost::launch(PrepareShadingNormalFwdKernel, gridSize, blockSize, p);
    return 0;
}

int ref_shading_normal_bwd(const float* pos, const int* d0, const float* view_pos, const int* d1, const float* perturbed_nrm, const int* d2,
                           const float* smooth_nrm, const int* d3, const float* smooth_tng, const int* d4, const float* geom_nrm, const int* d5,
                           const float* grad, int two_sided, int opengl, float* g0, float* g1, float* g2, float* g3, float* g4, float* g5) {
    const int* const dims[6] = {d0, d1, d2, d3, d4, d5};
    PrepareShadingNormalKernelParams p;
    memset(&p, 0, sizeof(p));
    p.two_sided_shading = two_sided != 0;
    p.opengl = opengl != 0;
    p.gridSize = grid_of(dims);
    dim3 blockSize = getLaunchBlockSize(BLOCK_X, BLOCK_Y, p.gridSize);
    dim3 gridSize = getLaunchGridSize(blockSize, p.gridSize);
    int od[4] = {(int)p.gridSize.z, (int)p.gridSize.y, (int)p.gridSize.x, 3};
    p.pos = make_tensor(pos, d0, 4, p.gridSize, g0);
    p.view_pos = make_tensor(view_pos, d1, 4, p.gridSize, g1);
    p.perturbed_nrm = make_tensor(perturbed_nrm, d2, 4, p.gridSize, g2);
    p.smooth_nrm = make_tensor(smooth_nrm, d3, 4, p.gridSize, g3);
    p.smooth_tng = make_tensor(smooth_tng, d4, 4, p.gridSize, g4);
    p.geom_nrm = make_tensor(geom_nrm, d5, 4, p.gridSize, g5);
    p.out = make_tensor(grad, od, 4, p.gridSize);
    cuhost::launch(PrepareShadingNormalBwdKernel, gridSize, blockSize, p);
    return 0;
}

}  // extern "C"

/*
 * gshell_hip.h -- C ABI of libgshell_hip.so, the MI355X (gfx950) hot path of G-Shell.
 *
 * Every entry point takes raw DEVICE pointers (HBM), plain sizes and the HIP stream to
 * launch on (a hipStream_t passed as void*).  No torch types cross this boundary: the
 * reference-side binding (pybind/ctypes stub) only needs
 * tensor.data_ptr() and torch.cuda.current_stream().cuda_stream.
 *
 * Conventions
 *   - return value 0 = ok, non-zero = error; gs_last_error() gives the message
 *     (thread-local, valid until the next failing call on that thread).
 *   - all memory is owned by the caller (torch's caching allocator); the library owns
 *     only the persistent per-grid scratch inside gs_mtets_topo / gs_bvh objects.
 *   - two-phase protocol for data-dependent sizes: *_count() -> caller allocates ->
 *     *_fill().  *_count() synchronises `stream` once (never the device).
 *   - empty meshes (zero crossing tets, zero faces) are legal everywhere
 *     (reference: render/render.py:361-365, render/optixutils/ops.py:134-139).
 *
 * Each function cites the reference interface it replaces (paths relative to
 * lzzcd001/GShell).
 */
#ifndef GSHELL_HIP_H
#define GSHELL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gs_stream_t; /* hipStream_t */

const char* gs_last_error(void);
int gs_version(void);
/* async device-to-device copy on `stream` (used to export library-owned tables) */
int gs_memcpy_d2d(void* dst, const void* src, int64_t bytes, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * G-MarchingTets   (replaces geometry/gshell_tets.py:245-443  GShell_Tets.__call__)
 * ---------------------------------------------------------------------------------- */

/* Static per-grid topology: int32 copy of the tets, lexicographically sorted unique
 * edge list [E,2] and tet->edge table [F,6] (replaces the per-call torch.unique at
 * gshell_tets.py:266-268 and generate_edges at gshell_tets_geometry.py:149-155),
 * plus the per-call scratch the kernels need. */
typedef struct gs_mtets_topo gs_mtets_topo;

int gs_mtets_topo_create(const int64_t* tet_fx4, int64_t F, int64_t N, gs_stream_t stream,
                         gs_mtets_topo** out);
int gs_mtets_topo_destroy(gs_mtets_topo* topo);
/* E = number of unique edges; edges_dev = device pointer to int32 [E,2] (sorted). */
int gs_mtets_topo_info(const gs_mtets_topo* topo, int64_t* N, int64_t* F, int64_t* E,
                       const int32_t** edges_dev, const int32_t** tet_i32_dev);

/* counts[0]=V   watertight vertices (= sign-crossing edges)
 * counts[1]=M1  tets emitting 1 triangle      counts[2]=M2  tets emitting 2
 * counts[3..8]  tets per mSDF-cut group (tri->1, tri->2, quad->1, quad->2, quad->3, quad->4)
 * counts[9]=T   faces after the mSDF cut      counts[10]=V_aug = V + 3 M1 + 4 M2      */
#define GS_MTETS_NCOUNTS 16
int gs_mtets_count(gs_mtets_topo* topo, const float* pos_nx3, const float* sdf_n,
                   const float* msdf_n, gs_stream_t stream, int64_t* counts_host);

/* Fill phase; must follow gs_mtets_count on the same topo/stream with the same inputs.
 *   verts_aug [V_aug,3] f32, msdf_aug [V_aug] f32 (stop-gradient mSDF, ref :386-390),
 *   verts_wt [V,3] f32 (ref 'vertices_watertight'), faces_wt [M1+2 M2,3] i64,
 *   faces_aug [T,3] i64, faces_aug_i32 [T,3] i32 (same data, for the rasteriser; may be NULL)
 * saved for backward (caller-owned):
 *   vert_ab [V,2] i32 grid endpoints of each watertight vertex, used_wt [V] u8,
 *   poly [3 M1 + 4 M2] i32 polygon corner vertex ids, cut_code [M1+M2] u8,
 *   tet_id [M1+M2] i32 source tet of each polygon,
 *   sign_code [M1+M2] u8 SDF sign pattern, grp_rank [M1+M2] i32 rank inside the cut group
 *   (the last two are scratch the caller may drop after the call).                    */
int gs_mtets_fill(gs_mtets_topo* topo, const float* pos_nx3, const float* sdf_n,
                  const float* msdf_n, float* verts_aug, float* msdf_aug, float* verts_wt,
                  int64_t* faces_wt, int64_t* faces_aug, int32_t* faces_aug_i32,
                  int32_t* vert_ab, uint8_t* used_wt, int32_t* poly, uint8_t* cut_code,
                  int32_t* tet_id, uint8_t* sign_code, int32_t* grp_rank, gs_stream_t stream);

/* Backward of the extraction (autograd of gshell_tets.py:277-392).
 *   g_verts_aug [V_aug,3], g_msdf_aug [V_aug], g_verts_wt [V,3] (any may be NULL = 0)
 *   scratch [V,5] f32 (caller-allocated, need not be zeroed)
 *   outputs g_pos [N,3], g_sdf [N], g_msdf [N] are ACCUMULATED into (caller zero-fills). */
int gs_mtets_bwd(int64_t N, int64_t V, int64_t M1, int64_t M2, const float* pos_nx3,
                 const float* sdf_n, const float* msdf_n, const float* verts_wt,
                 const float* msdf_aug, const int32_t* vert_ab, const uint8_t* used_wt,
                 const int32_t* poly, const uint8_t* cut_code, const float* g_verts_aug,
                 const float* g_msdf_aug, const float* g_verts_wt, float* scratch,
                 float* g_pos, float* g_sdf, float* g_msdf, gs_stream_t stream);

/* Tangent frame of the watertight mesh interpolated to the boundary vertices
 * (gshell_tets.py:9-78, :318-319, :375-380; forward only -- dead on the training path).
 *   scratch [V,7] f32; lin [Nuv] f32 = torch.linspace(0, 1-1/Nuv, Nuv), Nuv=ceil(sqrt(F)) */
int gs_mtets_tangents(int64_t V, int64_t M1, int64_t M2, int64_t F, const float* verts_wt,
                      const int64_t* faces_wt, const float* msdf_aug, const int32_t* poly,
                      const float* lin, int64_t Nuv, float* scratch, float* v_tng_aug,
                      gs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GSHELL_HIP_H */

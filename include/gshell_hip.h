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
/* "" for the shipped library.  A library built with non-default compile-time variants (tools/build_variant.sh) lists them, e.g.
 * "GS_WG_STRIPS=64 GS_EXPERIMENT(mlp_h2.hip)=1" -- several switches are timing-only ablations that give wrong results; bench.py refuses to
 * report a number for a library whose string is not empty. */
const char* gs_build_flags(void);
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
/* Fused geometry front end: the topology's own sign-bit array (1 bit per grid vertex, bit i of word i/64
 * = sdf[i] > 0) can be WRITTEN by the SDF-network kernel's epilogue (gs_sdf_mlp_fwd_h2, occ_bits argument), in which case
 * gs_mtets_count_presigned skips the sign pass over sdf (sdf itself is still read by the classification for the
 * interpolation weights).  The caller guarantees that the bits belong to this sdf. */
int gs_mtets_occ_bits(gs_mtets_topo* topo, uint64_t** bits_dev, int64_t* n_words);
int gs_mtets_count_presigned(gs_mtets_topo* topo, const float* sdf_n, const float* msdf_n,
                             gs_stream_t stream, int64_t* counts_host);
/* Row selection of the two-pass SDF evaluation (gs_sdf_mlp_fwd_h1): flags [N] (float, zeroed by the caller) receives 1 at both
 * end points of every edge of the static edge list whose end points differ in sign (sdf > 0, gshell_tets.py:250) or have
 * |sdf| < tau at either end. */
int gs_mtets_flag_refine_rows(const gs_mtets_topo* topo, const float* sdf_n, float tau, float* flags,
                              gs_stream_t stream);
/* The same rule over an explicit edge list [E,2] i32 (the unique cube edges of a G-FlexiCubes grid, gshell_flexicubes_geometry.py:124-127:
 * its extraction consumes signs everywhere and values only at the end points of sign-changing cube edges, gshell_flexicubes.py:387-485). */
int gs_flag_refine_rows_edges(const int32_t* edges, int64_t E, const float* sdf_n, float tau, float* flags,
                              gs_stream_t stream);

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
                 const float* g_msdf_aug, const float* g_verts_wt,
                 const float* g_mv_full /* [V] or NULL: d loss / d watertight mSDF values WITH the weights' gradient (gs_mtets_tangents_bwd) */,
                 float* scratch, float* g_pos, float* g_sdf, float* g_msdf, gs_stream_t stream);

/* Tangent frame of the watertight mesh interpolated to the boundary vertices
 * (gshell_tets.py:9-78, :318-319, :375-380; dead on the training path: gshell_tets_geometry.py:206-208, render.py:264-267).
 *   scratch [V,7] f32; lin [Nuv] f32 = torch.linspace(0, 1-1/Nuv, Nuv), Nuv=ceil(sqrt(F)) */
int gs_mtets_tangents(int64_t V, int64_t M1, int64_t M2, int64_t F, const float* verts_wt,
                      const int64_t* faces_wt, const float* msdf_aug, const int32_t* poly,
                      const float* lin, int64_t Nuv, float* scratch, float* v_tng_aug,
                      gs_stream_t stream);
/* Adjoint of gs_mtets_tangents = autograd through compute_tangents + auto_normals + the boundary interpolation
 * (gshell_tets.py:9-78, :318-319, :375-380).  acc = the forward call's `scratch` (kept by the caller), v_tng_aug its output,
 * g_tng_aug [V + 3 M1 + 4 M2, 3] the upstream gradient.  WRITTEN: g_verts_wt [V,3] (add it to gs_mtets_bwd's g_verts_wt) and
 * g_mv [V] (gs_mtets_bwd's g_mv_full).  work [V,9] f32. */
int gs_mtets_tangents_bwd(int64_t V, int64_t M1, int64_t M2, const float* verts_wt, const int64_t* faces_wt,
                          const float* msdf_aug, const int32_t* poly, const float* lin, int64_t Nuv,
                          const float* acc, const float* v_tng_aug, const float* g_tng_aug, float* work,
                          float* g_verts_wt, float* g_mv, gs_stream_t stream);

/* Generative-decode variant of the extraction (replaces GShell_Tets.marching_from_auggrid,
 * geometry/gshell_tets.py:446-629; caller getMesh_from_augmented_grid_withocc,
 * geometry/gshell_tets_geometry.py:167-189; no autograd in the reference either).
 * The mesh topology, vertex numbering and face order are those of gs_mtets_count/fill;
 * per-edge quantities come from cubic grids indexed at canonical edge midpoints:
 *   sdf_n [N] f32 (only its sign is used), vdisc [N,3] i32 = 'verts_discretized',
 *   coeff_grid [G,G,G] f32 = 'coeff_sdf_interp' (clamped to [0,1], ref :483),
 *   msdf_sign_grid [G,G,G] f32 = 'midpoint_msdf_sign_n', occgrid [G2,G2,G2] f32.
 * Same count -> allocate -> fill protocol and output layout as gs_mtets_fill, except:
 * unreferenced vertices are NOT zeroed, msdf_aug is 0 on boundary vertices (ref :597-600),
 * and bnd_w [3 M1 + 4 M2, 2] f32 receives the boundary weights (for gs_mtets_aug_tangents).
 * tet_id [M1+M2] i32 doubles as the reference's 'valid_tet_gidx' (ref :507).             */
int gs_mtets_aug_count(gs_mtets_topo* topo, const float* sdf_n, const int32_t* vdisc_nx3,
                       const float* msdf_sign_grid, int64_t G, gs_stream_t stream,
                       int64_t* counts_host);
int gs_mtets_aug_fill(gs_mtets_topo* topo, const float* pos_nx3, const float* sdf_n,
                      const int32_t* vdisc_nx3, const float* coeff_grid,
                      const float* msdf_sign_grid, int64_t G, const float* occgrid, int64_t G2,
                      float* verts_aug, float* msdf_aug, float* verts_wt, int64_t* faces_wt,
                      int64_t* faces_aug, int32_t* faces_aug_i32, int32_t* vert_ab, int32_t* poly,
                      uint8_t* cut_code, int32_t* tet_id, uint8_t* sign_code, int32_t* grp_rank,
                      float* bnd_w, gs_stream_t stream);
int gs_mtets_aug_tangents(int64_t V, int64_t M1, int64_t M2, const float* verts_wt,
                          const int64_t* faces_wt, const float* bnd_w, const int32_t* poly,
                          const float* lin, int64_t Nuv, float* scratch, float* v_tng_aug,
                          gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Vertex transform   (replaces ru.xfm_points, render/renderutils/ops.py:518-537;
 *                     CUDA kernels render/renderutils/c_src/mesh.cu:22-94)
 *   pts [Bp,V,3] (Bp = 1 or B), mtx [B,4,4] row-major -> out [B,V,4] = [p,1] . M^T
 *   bwd: g_pts is WRITTEN (summed over views when Bp == 1).
 * ---------------------------------------------------------------------------------- */
int gs_xfm_points_fwd(const float* pts, int64_t Bp, const float* mtx, int64_t B, int64_t V,
                      float* out, gs_stream_t stream);
int gs_xfm_points_bwd(const float* g_out, int64_t Bp, const float* mtx, int64_t B, int64_t V,
                      float* g_pts, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Rasterise   (replaces nvdiffrast dr.DepthPeeler(...).rasterize_next_layer() /
 *              dr.rasterize as called at render/render.py:377-379, :458)
 *   pos_clip [B,V,4] f32, tri [T,3] i32 -> rast [B,H,W,4] = (u, v, z/w, id+1),
 *   rast_db [B,H,W,4] = (du/dX, du/dY, dv/dX, dv/dY) (may be NULL),
 *   tri_visible [T] u8 (may be NULL; caller zero-fills; set to 1 for ids that own a pixel:
 *   replaces rast[...,-1].long().unique(), render/render.py:380-383).
 *   scratch: gs_rasterize_scratch_bytes(B,T,H,W) bytes, caller-owned, contents undefined.
 *   bwd: g_rast [B,H,W,4] (only u,v components used) -> g_pos [B,V,4] ACCUMULATED.
 * ---------------------------------------------------------------------------------- */
int64_t gs_rasterize_scratch_bytes(int64_t B, int64_t T, int64_t H, int64_t W);
int gs_rasterize_fwd(const float* pos_clip, int64_t B, int64_t V, const int32_t* tri, int64_t T,
                     int64_t H, int64_t W, void* scratch, float* rast, float* rast_db,
                     uint8_t* tri_visible, gs_stream_t stream);
int gs_rasterize_bwd(const float* pos_clip, int64_t B, int64_t V, const int32_t* tri, int64_t T,
                     int64_t H, int64_t W, const float* rast, const float* g_rast, float* g_pos,
                     gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Interpolate   (replaces dr.interpolate, render/render.py:25-26 and its call sites
 *                :240, :248, :263, :275, :306)
 *   attr [Ba,V,A] (Ba = 1 or B) -> out [B,H,W,A]; out_da [B,H,W,2A] = (d/dX, d/dY) per
 *   attribute if out_da && rast_db.  bwd: g_attr ACCUMULATED (may be NULL), g_rast WRITTEN
 *   as (d/du, d/dv, 0, 0) (may be NULL).  out_da carries no gradient (the reference only
 *   uses it under no_grad, render/render.py:272-279).
 * ---------------------------------------------------------------------------------- */
int gs_interpolate_fwd(const float* attr, int64_t Ba, int64_t V, int64_t A, const float* rast,
                       const float* rast_db, const int32_t* tri, int64_t T, int64_t B, int64_t H,
                       int64_t W, float* out, float* out_da, gs_stream_t stream);
int gs_interpolate_bwd(const float* attr, int64_t Ba, int64_t V, int64_t A, const float* rast,
                       const int32_t* tri, int64_t T, int64_t B, int64_t H, int64_t W,
                       const float* g_out, float* g_attr, float* g_rast, gs_stream_t stream);
/* Several per-vertex attribute tensors [V, channels[k]] (k < n_groups <= 4, batch 1) interpolated in ONE pass into one contiguous
 * output each (the g-buffer of a frame: position, smooth normal, mSDF; reference render/render.py:240, :263, :306).  `attrs`, `outs`,
 * `g_outs`, `g_attrs` are HOST arrays of device pointers.  Per channel the same arithmetic as gs_interpolate_*, g_rast accumulated
 * in the channel order of the concatenated tensor: bit-identical to interpolating torch.cat(attrs, -1).
 * bwd: g_outs[k] NULL = no gradient into that output; g_attrs[k] ACCUMULATED (zero it first), NULL = not needed; g_rast WRITTEN or NULL. */
int gs_interpolate_groups_fwd(int n_groups, const int32_t* channels, const float* const* attrs,
                              const float* rast, const int32_t* tri, int64_t T, int64_t B, int64_t H,
                              int64_t W, float* const* outs, gs_stream_t stream);
int gs_interpolate_groups_bwd(int n_groups, const int32_t* channels, const float* const* attrs,
                              const float* rast, const int32_t* tri, int64_t T, int64_t B, int64_t H,
                              int64_t W, const float* const* g_outs, float* const* g_attrs,
                              float* g_rast, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Antialias   (replaces dr.antialias, render/render.py:352-359 composite_buffer; the
 *              reference calls it once per buffer key, :417-433)
 *   gs_tri_adjacency : opp [T,3] i32 = vertex opposite edge e in the neighbouring triangle
 *                      (-1 unless exactly two triangles share the edge)
 *   gs_aa_analyze    : alpha [B,H,W,2] blend factor of the (right, down) pixel pair
 *   gs_aa_apply_fwd  : out [B,H,W,C] = color + blends (C = any number of stacked channels)
 *   gs_aa_apply_bwd  : g_color [B,H,W,C] WRITTEN, g_alpha [B,H,W,2] WRITTEN (either may be NULL)
 *   gs_aa_analyze_bwd: g_pos [B,V,4] ACCUMULATED
 * ---------------------------------------------------------------------------------- */
int64_t gs_tri_adjacency_scratch_bytes(int64_t T);
int gs_tri_adjacency(const int32_t* tri, int64_t T, int64_t V, void* scratch, int32_t* opp,
                     gs_stream_t stream);
int gs_aa_analyze(const float* pos_clip, int64_t B, int64_t V, const int32_t* tri, int64_t T,
                  const int32_t* opp, const float* rast, int64_t H, int64_t W, float* alpha,
                  gs_stream_t stream);
int gs_aa_apply_fwd(const float* color, const float* alpha, int64_t B, int64_t H, int64_t W,
                    int64_t C, float* out, gs_stream_t stream);
int gs_aa_apply_bwd(const float* color, const float* alpha, int64_t B, int64_t H, int64_t W,
                    int64_t C, const float* g_out, float* g_color, float* g_alpha,
                    gs_stream_t stream);
/* The same apply IN PLACE, for a frame the caller owns (only silhouette pixels are touched: ~1 % of the frame instead of two full passes each way):
 *   fwd: color [B,H,W,C] UPDATED; scratch = a tensor of the same shape (uninitialised) that afterwards holds the ORIGINAL colours of the
 *        modified pixels (keep it for the backward pass);
 *   bwd: out_color = the updated frame, saved = that scratch, g [B,H,W,C] = d loss / d out on entry and d loss / d color on exit,
 *        g_scratch = another tensor of that shape (uninitialised), g_alpha [B,H,W,2] WRITTEN or NULL.  Values and gradients are
 *        bit-identical to gs_aa_apply_fwd / bwd (g_alpha up to the order in which the streaming kernel adds its channels). */
int gs_aa_apply_fwd_inplace(float* color, const float* alpha, int64_t B, int64_t H, int64_t W,
                            int64_t C, float* scratch, gs_stream_t stream);
int gs_aa_apply_bwd_inplace(const float* out_color, const float* saved, const float* alpha,
                            int64_t B, int64_t H, int64_t W, int64_t C, float* g, float* g_scratch,
                            float* g_alpha, gs_stream_t stream);
int gs_aa_analyze_bwd(const float* pos_clip, int64_t B, int64_t V, const int32_t* tri, int64_t T,
                      const int32_t* opp, const float* rast, int64_t H, int64_t W,
                      const float* alpha, const float* g_alpha, float* g_pos, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * prepare_shading_normal   (replaces ru.prepare_shading_normal, render/renderutils/ops.py:197-229;
 *                           CUDA render/renderutils/c_src/normal.cu:18-181)
 *   all tensors [B*pix_per_view, 3]; view_pos is [B,3] (view_full = 0) or per pixel (view_full = 1);
 *   perturbed_nrm may be NULL (= (0,0,1), the G-Shell case: material['no_perturbed_nrm']).
 *   bwd: every g_* pointer may be NULL; all are WRITTEN per pixel (g_view_pos_full is per pixel
 *   even when view_full = 0: the caller reduces it).
 * ---------------------------------------------------------------------------------- */
int gs_shading_normal_fwd(const float* pos, const float* view_pos, int view_full,
                          const float* perturbed_nrm, const float* smooth_nrm,
                          const float* smooth_tng, const float* geom_nrm, int64_t B,
                          int64_t pix_per_view, int two_sided, int opengl, float* out,
                          gs_stream_t stream);
int gs_shading_normal_bwd(const float* pos, const float* view_pos, int view_full,
                          const float* perturbed_nrm, const float* smooth_nrm,
                          const float* smooth_tng, const float* geom_nrm, int64_t B,
                          int64_t pix_per_view, int two_sided, int opengl, const float* g_out,
                          float* g_pos, float* g_view_pos_full, float* g_perturbed_nrm,
                          float* g_smooth_nrm, float* g_smooth_tng, float* g_geom_nrm,
                          gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * image_loss   (replaces ru.image_loss, render/renderutils/ops.py:479-503; c_src/loss.cu:15-210)
 *   img, target: n floats (any shape, same layout).  loss: 0 l1, 1 mse, 2 relmse, 3 smape;
 *   tonemapper: 0 none, 1 log_srgb.  fwd writes gs_image_loss_partials(n) per-block sums
 *   (the caller sums them and divides by n, like the reference's partial-sum tensor + torch.sum).
 *   bwd: g = *g_scalar_dev * scale per element; g_img / g_target WRITTEN (either may be NULL).
 * ---------------------------------------------------------------------------------- */
int64_t gs_image_loss_partials(int64_t n);
int gs_image_loss_fwd(const float* img, const float* target, int64_t n, int loss, int tonemapper,
                      float* partials, gs_stream_t stream);
int gs_image_loss_bwd(const float* img, const float* target, int64_t n, int loss, int tonemapper,
                      const float* g_scalar_dev, float scale, float* g_img, float* g_target,
                      gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * auto_normals   (replaces mesh.auto_normals, render/mesh.py:212-237)
 *   acc [V,3] = unnormalised area-weighted normal sums (saved for bwd), v_nrm [V,3].
 *   bwd: g_acc [V,3] scratch WRITTEN, g_pos [V,3] ACCUMULATED.
 * ---------------------------------------------------------------------------------- */
int gs_auto_normals_fwd(const float* v_pos, int64_t V, const int32_t* tri, int64_t T, float* acc,
                        float* v_nrm, gs_stream_t stream);
int gs_auto_normals_bwd(const float* v_pos, int64_t V, const int32_t* tri, int64_t T,
                        const float* acc, const float* g_nrm, float* g_acc, float* g_pos,
                        gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Bilinear image tap with clamp addressing   (replaces dr.texture(tex, uv, filter_mode='linear',
 *   boundary_mode='clamp') as used at render/render.py:59, :110)
 *   tex [B,H,W,C], uv [B,n_per_view,2] in [0,1] (texel centres at (i+.5)/W) -> out [B,n_per_view,C]
 *   bwd: g_tex [B,H,W,C] ACCUMULATED; uv carries no gradient (it is noise in the reference).
 * ---------------------------------------------------------------------------------- */
int gs_texture_linear_fwd(const float* tex, int64_t B, int64_t H, int64_t W, int64_t C,
                          const float* uv, int64_t n_per_view, float* out, gs_stream_t stream);
int gs_texture_linear_bwd(int64_t B, int64_t H, int64_t W, int64_t C, const float* uv,
                          int64_t n_per_view, const float* g_out, float* g_tex, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Shadow-ray acceleration structure   (replaces ou.OptiXContext / ou.optix_build_bvh,
 *   render/optixutils/ops.py:128-139 -> c_src/torch_bindings.cpp:37-116; rebuilt every
 *   iteration by geometry/gshell_tets_geometry.py:211)
 *   The object owns its device buffers (grown on demand); verts [V,3] f32, tris [T,3] i32.
 *   T == 0 is legal (every ray is unoccluded).  gs_bvh_any_hit: hit[i] = 1 if ray
 *   (origins[i], dirs[i]) meets any triangle at t in (0, 1e16)  (kernel.cu:101-117).
 * ---------------------------------------------------------------------------------- */
typedef struct gs_bvh gs_bvh;
int gs_bvh_create(gs_bvh** out);
int gs_bvh_destroy(gs_bvh* bvh);
int gs_bvh_build(gs_bvh* bvh, const float* verts, int64_t V, const int32_t* tris, int64_t T,
                 gs_stream_t stream);
int gs_bvh_info(const gs_bvh* bvh, int64_t* T, int64_t* depth, int64_t* leaf_size, int64_t* bytes);
int gs_bvh_any_hit(const gs_bvh* bvh, const float* origins, const float* dirs, int64_t n,
                   uint8_t* hit, gs_stream_t stream);
/* diagnostic variant: stats [n,2] i32 = (internal nodes visited, triangles tested) per ray */
int gs_bvh_any_hit_stats(const gs_bvh* bvh, const float* origins, const float* dirs, int64_t n,
                         uint8_t* hit, int32_t* stats, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Monte-Carlo environment shading   (replaces ou.optix_env_shade, render/optixutils/ops.py:81-108,
 *   :141-143; raygen program c_src/envsampling/kernel.cu:463-541; BSDF c_src/bsdf.h)
 *   pix [n_cov] i32 = linear indices (b*H*W + y*W + x) of the pixels with mask > 0, ascending;
 *   ro, gb_pos, gb_normal, gb_kd, gb_ks [B,H,W,3]; view_pos [B,3];
 *   light [Hl,Wl,3], pdf [Hl,Wl], rows [Hl], cols [Hl,Wl]; perms [P, n^2] i32;
 *   bsdf 0 'pbr' / 1 'diffuse' / 2 'white'; rays per covered pixel = 2 n^2.
 *   fwd: scratch = gs_env_shade_scratch_bytes(n_cov, n) bytes (ray directions + unshadowed contributions);
 *        vis_bits [gs_env_shade_vis_words(n_cov, n)] u64 WRITTEN: 1 bit per ray, 1 = unoccluded (saved for bwd);
 *        diff, spec [B,H,W,3] WRITTEN (both NULL = trace only, used when the backward pass wants fresh samples).
 *   bwd: identical sampling with the cached visibility (no rays are traced):
 *        g_pos, g_normal, g_kd, g_ks [B,H,W,3] WRITTEN; g_light [Hl,Wl,3] ACCUMULATED (atomics).
 *   ro and view_pos receive no gradient, as in the reference (ops.py:108).  ro NULL: gb_pos + gb_normal * 0.001f, what the reference's caller passes.
 *   gb_kd / gb_ks (and likewise g_kd / g_ks) may be the two channel halves of ONE [B,H,W,6] tensor -- the layout MLPTexture3D
 *        produces -- instead of two [B,H,W,3] tensors: pass gb_ks = gb_kd + 3 (g_ks = g_kd + 3); the pixel stride is then 6 floats
 *        and the gradient tensor is written (and zeroed) as one.  Saves the channel-split copies and the slice backward.
 *   view_offset / view_stride: local view b is view  b*view_stride + view_offset  of the GLOBAL batch; the per-pixel
 *        RNG stream hashes that global pixel index (kernel.cu:504), so a view-sharded N-GPU step draws exactly the
 *        samples of the single-GPU step.  Single GPU: (0, 1).
 * ---------------------------------------------------------------------------------- */
int64_t gs_env_shade_scratch_bytes(int64_t n_cov, int n_samples_x);
int64_t gs_env_shade_vis_words(int64_t n_cov, int n_samples_x);
int gs_env_shade_fwd(const gs_bvh* bvh, const int32_t* pix, int64_t n_cov, const float* ro,
                     const float* gb_pos, const float* gb_normal, const float* view_pos,
                     const float* gb_kd, const float* gb_ks, const float* light, const float* pdf,
                     const float* rows, const float* cols, int64_t Hl, int64_t Wl,
                     const int32_t* perms, int64_t P, int64_t B, int64_t H, int64_t W,
                     int64_t view_offset, int64_t view_stride, int bsdf,
                     int n_samples_x, uint32_t rnd_seed, float shadow_scale, void* scratch,
                     uint64_t* vis_bits, float* diff, float* spec, gs_stream_t stream);
/* gs_env_shade_fwd with a BOUNDED scratch: the covered pixels are shaded in chunks (multiples of 64 pixels) whose per-sample records
 * (40 B per ray) fit `scratch_bytes`; outputs and vis_bits are bit-identical to gs_env_shade_fwd for every bound (each pixel's samples hash its
 * GLOBAL index, a ray's visibility does not depend on its batch).  For frames whose records would not fit a fixed budget -- the reference's
 * own default workload, configs/deepfashion_mc_256.json:7-8,17 (2 x 1024^2, n_samples 24: 1152 rays per covered pixel, kernel.cu:490-529) needs
 * ~14 GB of them.  After a call that took more than one chunk only the last chunk's records exist: back-propagate with gs_env_shade_bwd_bounded
 * (records regenerated chunk by chunk) or gs_env_shade_bwd (sampler replay), not gs_env_shade_bwd_saved. */
int gs_env_shade_fwd_bounded(const gs_bvh* bvh, const int32_t* pix, int64_t n_cov, const float* ro,
                             const float* gb_pos, const float* gb_normal, const float* view_pos,
                             const float* gb_kd, const float* gb_ks, const float* light, const float* pdf,
                             const float* rows, const float* cols, int64_t Hl, int64_t Wl,
                             const int32_t* perms, int64_t P, int64_t B, int64_t H, int64_t W,
                             int64_t view_offset, int64_t view_stride, int bsdf,
                             int n_samples_x, uint32_t rnd_seed, float shadow_scale, void* scratch,
                             int64_t scratch_bytes, uint64_t* vis_bits, float* diff, float* spec,
                             gs_stream_t stream);
/* (gs_env_shade_bwd = the sampler replayed in ONE kernel, round 1's backward: an ORACLE KERNEL since round 6 -- lib/variants/oracles.so only; the shipped
 *  backward passes are gs_env_shade_bwd_saved and gs_env_shade_bwd_bounded below.) */
int gs_env_shade_bwd(const gs_bvh* bvh, const int32_t* pix, int64_t n_cov, const float* gb_pos,
                     const float* gb_normal, const float* view_pos, const float* gb_kd,
                     const float* gb_ks, const float* light, const float* pdf, const float* rows,
                     const float* cols, int64_t Hl, int64_t Wl, const int32_t* perms, int64_t P,
                     int64_t B, int64_t H, int64_t W, int64_t view_offset, int64_t view_stride,
                     int bsdf, int n_samples_x, uint32_t rnd_seed,
                     float shadow_scale, const uint64_t* vis_bits, const float* g_diff,
                     const float* g_spec, float* g_pos, float* g_normal, float* g_kd, float* g_ks,
                     float* g_light, gs_stream_t stream);
/* gs_env_shade_bwd_saved for a frame shaded by gs_env_shade_fwd_bounded: per chunk of covered pixels the sampler regenerates the chunk's records into
 * `scratch` (same RNG streams; no rays -- visibility from vis_bits) and the saved-samples gradient kernels run on them.  Per-pixel gradients
 * bit-identical to gs_env_shade_bwd_saved, g_light ACCUMULATED (equal up to float-atomic order). */
int gs_env_shade_bwd_bounded(const gs_bvh* bvh, const int32_t* pix, int64_t n_cov, const float* gb_pos,
                             const float* gb_normal, const float* view_pos, const float* gb_kd,
                             const float* gb_ks, const float* light, const float* pdf, const float* rows,
                             const float* cols, int64_t Hl, int64_t Wl, const int32_t* perms, int64_t P,
                             int64_t B, int64_t H, int64_t W, int64_t view_offset, int64_t view_stride,
                             int bsdf, int n_samples_x, uint32_t rnd_seed, float shadow_scale,
                             const uint64_t* vis_bits, void* scratch, int64_t scratch_bytes,
                             const float* g_diff, const float* g_spec, float* g_pos, float* g_normal,
                             float* g_kd, float* g_ks, float* g_light, gs_stream_t stream);
/* The same gradients from the forward pass's SAVED samples: fwd_scratch = the scratch buffer of the gs_env_shade_fwd call
 * (kept alive by the caller; it holds every ray's direction and MIS weight, which the reference's backward treats as
 * constants too).  No RNG replay / CDF searches / BSDF sampling; bit-identical per-pixel gradients. */
int gs_env_shade_bwd_saved(const gs_bvh* bvh, const int32_t* pix, int64_t n_cov, const float* gb_pos,
                           const float* gb_normal, const float* view_pos, const float* gb_kd,
                           const float* gb_ks, const float* light, const float* pdf, const float* rows,
                           const float* cols, int64_t Hl, int64_t Wl, const int32_t* perms, int64_t P,
                           int64_t B, int64_t H, int64_t W, int64_t view_offset, int64_t view_stride,
                           int bsdf, int n_samples_x, uint32_t rnd_seed, float shadow_scale,
                           const uint64_t* vis_bits, const void* fwd_scratch, const float* g_diff,
                           const float* g_spec, float* g_pos, float* g_normal, float* g_kd, float* g_ks,
                           float* g_light, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Bilateral denoiser   (replaces ou.bilateral_denoiser, render/optixutils/ops.py:110-123, :145-147;
 *   c_src/denoising.cu:14-130).  col, nrm [B,H,W,3], zdz [B,H,W,2] -> out [B,H,W,4] = (sum w c, max(sum w, 1e-4));
 *   bwd: g_out [B,H,W,4] -> g_col [B,H,W,3] WRITTEN (the reference's non-adjoint depth weight is kept).
 * ---------------------------------------------------------------------------------- */
int gs_bilateral_fwd(const float* col, const float* nrm, const float* zdz, int64_t B, int64_t H,
                     int64_t W, float sigma, float* out, gs_stream_t stream);
int gs_bilateral_bwd(const float* nrm, const float* zdz, int64_t B, int64_t H, int64_t W, float sigma,
                     const float* g_out, float* g_col, gs_stream_t stream);
/* the same filter when only the pixels with mask [B,H,W] > 0 are consumed downstream (covered pixels: the composite
 * multiplies the others by alpha = 0): unwanted pixels still act as taps, their own outputs are written as
 * (0, 0, 0, 1e-4) / zero gradient without running their tap loops.  mask = NULL: identical to the functions above. */
int gs_bilateral_fwd_masked(const float* col, const float* nrm, const float* zdz, const float* mask,
                            int64_t B, int64_t H, int64_t W, float sigma, float* out,
                            gs_stream_t stream);
int gs_bilateral_bwd_masked(const float* nrm, const float* zdz, const float* mask, int64_t B, int64_t H,
                            int64_t W, float sigma, const float* g_out, float* g_col,
                            gs_stream_t stream);
/* two colour images (diffuse and specular radiance, render.py:231-232) with the same guides in one pass: the filter weights
 * depend on the guides only and are most of a tap's arithmetic.  Same values as two single calls. */
int gs_bilateral_fwd_masked2(const float* col_a, const float* col_b, const float* nrm, const float* zdz,
                             const float* mask, int64_t B, int64_t H, int64_t W, float sigma,
                             float* out_a, float* out_b, gs_stream_t stream);
int gs_bilateral_bwd_masked2(const float* nrm, const float* zdz, const float* mask, int64_t B, int64_t H,
                             int64_t W, float sigma, const float* g_out_a, const float* g_out_b,
                             float* g_col_a, float* g_col_b, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Multiresolution hash-grid encoding   (replaces tinycudann.Encoding(3, HashGrid cfg) as configured by
 *   render/mlptexture.py:57-73 and sampled at render/render.py:68,70)
 *   x [N,3] in [0,1], mask [N] (NULL = all rows; rows with mask <= 0 produce zeros),
 *   params [gs_hashgrid_num_params] f32, out [N, n_levels*F].
 *   bwd: g_params ACCUMULATED (atomics; may be NULL), g_x_levels [n_levels,N,3] WRITTEN per level
 *   (the caller sums over levels; may be NULL).
 * ---------------------------------------------------------------------------------- */
int64_t gs_hashgrid_num_params(int n_levels, int F, int log2_T, int base_res, float per_level_scale);
int gs_hashgrid_fwd(int n_levels, int F, int log2_T, int base_res, float per_level_scale,
                    const float* x, const float* mask, int64_t N, const float* params, float* out,
                    gs_stream_t stream);
int gs_hashgrid_bwd(int n_levels, int F, int log2_T, int base_res, float per_level_scale,
                    const float* x, const float* mask, int64_t N, const float* params,
                    const float* g_out, float* g_params, float* g_x_levels, gs_stream_t stream);

/* The texture-field path of MLPTexture3D.sample in one kernel each way (render/mlptexture.py:87-99 incl. the AABB
 *   normalisation :89-90 and the encoder's 1/128 gradient hook :74), F = 2 only:
 *   pos [N,3] world positions; aabb [2,3] device (lo, hi; NULL = pos is already the [0,1] coordinate);
 *   t = clamp((pos - lo) / (hi - lo), 0, 1);   feat_level_major [n_levels, N, 2]: rows with mask <= 0 are NOT written
 *   (gs_texmlp_*_level_major never read them).
 *   bwd: g_params ACCUMULATED with table_scale * d/d table (NULL = skip); g_pos [N,3] WRITTEN (NULL = skip) =
 *   grad_scale * d/dt, through the clamp (closed interval) and the division; rows with mask <= 0 get 0.  img_w, img_h: when the N rows are whole row-major
 *   images of that size (multiples of 16) a workgroup owns a 16x16 pixel tile and combines its table updates in LDS
 *   before the global atomics (0 = rows in arbitrary order: 256 consecutive rows per workgroup). */
int gs_hashgrid_encode_fwd(int n_levels, int F, int log2_T, int base_res, float per_level_scale,
                           const float* pos, const float* aabb, const float* mask, int64_t N,
                           const float* params, float* feat_level_major, gs_stream_t stream);
/* forward over a compact list of points (rows [cap] i32, count_dev on the device: gs_compact_rows of the mask) */
int gs_hashgrid_encode_fwd_rows(int n_levels, int F, int log2_T, int base_res, float per_level_scale,
                                const float* pos, const float* aabb, const int32_t* rows,
                                const int64_t* count_dev, int64_t cap, int64_t N, const float* params,
                                float* feat_level_major, gs_stream_t stream);
int gs_hashgrid_encode_bwd(int n_levels, int F, int log2_T, int base_res, float per_level_scale,
                           const float* pos, const float* aabb, const float* mask, int64_t N,
                           const float* params, const float* g_feat_level_major, float* g_params,
                           float* g_pos, float grad_scale, float table_scale, int64_t img_w,
                           int64_t img_h,
                           gs_stream_t stream);
/* The same backward with the table gradient of the HASHED levels (res^3 > table size) gathered without atomics: the level's
 * table is cut into bins of 4096 entries, k_encode_bwd writes (entry, d pair) records into per-bin arrays of `bin_capacity`
 * 12-byte records (one returning atomic per workgroup, level and bin reserves the run), a second launch sums every bin in LDS
 * and adds it to g_params.  A reservation past `bin_capacity` takes the atomic path, so the result is correct for ANY capacity;
 * uniform hashing puts ~ 8 * (rows with mask > 0) / 128 records in a bin.  `bin_count` [gs_hashgrid_bin_count()] uint32, zero
 * before the first call (the reducer leaves the words zero); `spill_count` (ONE uint32, may be NULL): += the number of records that
 * spilled to the atomic path, never reset by the library; `bin_records` [bins * bin_capacity * 12 bytes] scratch. */
int64_t gs_hashgrid_bin_count(int n_levels, int F, int log2_T, int base_res, float per_level_scale);
int64_t gs_hashgrid_bin_entries(void);     /* table entries per bin */
int gs_hashgrid_encode_bwd_binned(int n_levels, int F, int log2_T, int base_res, float per_level_scale,
                                  const float* pos, const float* aabb, const float* mask, int64_t N,
                                  const float* params, const float* g_feat_level_major,
                                  float* g_params, float* g_pos, float grad_scale, float table_scale,
                                  int64_t img_w, int64_t img_h, uint32_t* bin_count,
                                  uint32_t* spill_count /* may be NULL */, void* bin_records,
                                  int64_t bin_capacity, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Softplus with first and second derivative   (the activation of geometry/mlp.py:19-33, nn.Softplus(beta=100);
 *   replaces ATen's softplus / softplus_backward / the ~9-op expansion of softplus_double_backward on the torch
 *   formulation of the SDF network: row-sparse backward and the eikonal term's double backward)
 *   y = x if beta x > 20 else log1p(exp(beta x)) / beta;   g_x = g s,  s = sigmoid(beta x) (1 if beta x > 20);
 *   bwd_bwd: d_g = gg s,  d_x = gg g beta s (1 - s)  (0 if beta x > 20);  either output may be NULL.
 * ---------------------------------------------------------------------------------- */
int gs_softplus_fwd(const float* x, int64_t n, float beta, float* y, gs_stream_t stream);
int gs_softplus_bwd(const float* x, const float* g, int64_t n, float beta, float* g_x, gs_stream_t stream);
int gs_softplus_bwd_bwd(const float* x, const float* g, const float* gg, int64_t n, float beta,
                        float* d_g, float* d_x, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Whole-frame loss / regulariser sums   (replaces the per-term torch expressions of
 *   geometry/gshell_tets_geometry.py:275-285 (alpha MSE, mSDF image terms) and render/regularizer.py:21-41
 *   (material / normal smoothness, monochrome-lighting prior) -- ~230 elementwise launches per iteration)
 *   stacked [n_pixels, C] f32 = the antialiased frame buffers side by side, color_ref [n_pixels,4] f32,
 *   offs_host[7] (HOST memory) = channel offset of shaded(4), msdf_image(1), diffuse_light(4),
 *   specular_light(4), kd_grad(4), ks_grad(4), normal_grad(4) inside a pixel record, -1 = absent.
 *   fwd: partials [gs_frame_sums_partials(n_pixels), 9] f32; the caller sums over rows to get
 *        (sum (a-m)^2, sum |msdf+ [m=0]|, sum |msdf- [m=1] - 1|, sum |logsrgb((d+s) m) - logsrgb(max(ref) m)|,
 *         sum luma(spec), sum luma(diff), sum kd_grad, sum ks_grad, sum normal_grad).
 *   bwd: g_sums_dev [9] f32 (device) -> g_stacked [n_pixels, C] WRITTEN (zeros in untouched channels).
 * ---------------------------------------------------------------------------------- */
int64_t gs_frame_sums_partials(int64_t n_pixels);
int gs_frame_sums_fwd(const float* stacked, const float* color_ref, int64_t n_pixels, int64_t C,
                      const int32_t* offs_host, float* partials, gs_stream_t stream);
int gs_frame_sums_bwd(const float* stacked, const float* color_ref, int64_t n_pixels, int64_t C,
                      const int32_t* offs_host, const float* g_sums_dev, float* g_stacked,
                      gs_stream_t stream);
/* the same pass with a TENTH sum: the colour term of the image loss, sum over pixels and rgb of
 * loss(tonemap(shaded * m), tonemap(reference * m))  (gshell_tets_geometry.py:277 through renderutils' image_loss,
 * loss / tonemapper ids as gs_image_loss_fwd) -- the frame then has one consumer and one gradient tensor.
 * partials10 [gs_frame_sums_partials(n_pixels), 10], g_sums10_dev [10]. */
int gs_frame_sums_img_fwd(const float* stacked, const float* color_ref, int64_t n_pixels, int64_t C,
                          const int32_t* offs_host, int loss, int tonemapper, float* partials10,
                          gs_stream_t stream);
int gs_frame_sums_img_bwd(const float* stacked, const float* color_ref, int64_t n_pixels, int64_t C,
                          const int32_t* offs_host, int loss, int tonemapper,
                          const float* g_sums10_dev, float* g_stacked, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Texture-field MLP   (replaces the torch `_MLP` + sigmoid range mapping of MLPTexture3D.sample,
 *   render/mlptexture.py:18-44, :87-99): out = sigmoid(W3 relu(W2 relu(W1 x))) * (hi - lo) + lo
 *   x [N,32] f32 (hash-grid features), mask [N] f32 or NULL (rows with mask <= 0 get the value of an
 *   all-zero feature row, 0.5 (hi - lo) + lo, and no gradient), w1, w2 [32,32], w3 [C,32] row-major
 *   [out][in] (nn.Linear.weight), lo, hi [C], C <= 8, out [N,C].
 *   bwd: g_x [N,32] WRITTEN (may be NULL); g_w1, g_w2, g_w3 ACCUMULATED (caller zero-fills; may be NULL).
 * ---------------------------------------------------------------------------------- */
int gs_texmlp_fwd(const float* x, const float* mask, int64_t N, const float* w1, const float* w2,
                  const float* w3, int C, const float* lo, const float* hi, float* out,
                  gs_stream_t stream);
int gs_texmlp_bwd(const float* x, const float* mask, int64_t N, const float* w1, const float* w2,
                  const float* w3, int C, const float* lo, const float* hi, const float* g_out,
                  float* g_x, float* g_w1, float* g_w2, float* g_w3, gs_stream_t stream);
/* the same network on the level-major feature tensor of gs_hashgrid_encode_* (x, g_x: [16, N, 2]); g_x is written
 * for rows with mask > 0 only */
int gs_texmlp_fwd_level_major(const float* x, const float* mask, int64_t N, const float* w1,
                              const float* w2, const float* w3, int C, const float* lo,
                              const float* hi, float* out, gs_stream_t stream);
int gs_texmlp_bwd_level_major(const float* x, const float* mask, int64_t N, const float* w1,
                              const float* w2, const float* w3, int C, const float* lo,
                              const float* hi, const float* g_out, float* g_x, float* g_w1,
                              float* g_w2, float* g_w3, gs_stream_t stream);
/* ... over a COMPACT list of rows (rows [cap] i32 ascending, count_dev [2] i64 on the device, e.g. gs_compact_rows of the mask:
 * no host sync): only listed rows are read / written -- out rows outside the list keep what the caller put there (the value
 * of an all-zero feature row, 0.5 (hi - lo) + lo); image rows in scan order leave two thirds of the lanes of a masked pass idle. */
int gs_texmlp_fwd_rows(const float* x_level_major, const int32_t* rows, const int64_t* count_dev,
                       int64_t cap, int64_t N, const float* w1, const float* w2, const float* w3, int C,
                       const float* lo, const float* hi, float* out, gs_stream_t stream);
int gs_texmlp_bwd_rows(const float* x_level_major, const int32_t* rows, const int64_t* count_dev,
                       int64_t cap, int64_t N, const float* w1, const float* w2, const float* w3, int C,
                       const float* lo, const float* hi, const float* g_out, float* g_x_level_major,
                       float* g_w1, float* g_w2, float* g_w3, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Adam step over a list of fp32 parameter tensors in one launch   (the reference steps three torch.optim.Adam optimisers,
 *   train_gshelltet_deepfashion.py:372-383, :446-452; amsgrad off, weight_decay 0, maximize off)
 *   params / grads / exp_avg / exp_avg_sq / step_tensors: HOST arrays [n_tensors] of device pointers, numel / lr host arrays;
 *   params, exp_avg, exp_avg_sq UPDATED in place; step_tensors[k] (device float32, may be NULL) WRITTEN = step_value (the state
 *   layout of torch's fused optimiser, so state_dicts are interchangeable); step_value = this step's number, from 1.
 *   Arithmetic: at::native's fused Adam (double scalars, fp32 operands, ATen/native/cuda/fused_adam_utils.cuh:61-76); `contract`
 *   0 / 1 / 2 = how the two moment updates round (csrc/adam.hip): ATen is built with -ffp-contract=fast.
 * ---------------------------------------------------------------------------------- */
int gs_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                 float* const* exp_avg_sq, float* const* step_tensors, const int64_t* numel,
                 const double* lr, double beta1, double beta2, double eps, double step_value,
                 int contract, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Depth + depth-slope guide image of the denoiser   (render/render.py:273-279: clamp / add / abs / div on one-channel images)
 *   clip_pos [n,4] interpolated clip-space position, clip_pos_deriv [n,8] its screen-space derivatives (gs_interpolate_fwd's out_da);
 *   out [n,2] WRITTEN = (z0, |z1 - z0|), z0 = max(z, eps) / max(w, eps), z1 = max(z + |dd[2]|, eps) / max(w + |dd[3]|, eps).  No gradient
 *   (the reference computes it under no_grad).
 * ---------------------------------------------------------------------------------- */
int gs_depth_zgrad(const float* clip_pos, const float* clip_pos_deriv, int64_t n, float eps, float* out,
                   gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Surface samples for the eikonal term   (stands in for kaolin.ops.mesh.sample_points, third party, called at
 *   geometry/gshell_tets_geometry.py:236; the samples are detached at :303, so no gradient is needed)
 *   gs_tri_area: area [T] WRITTEN = |(v1 - v0) x (v2 - v0)| (non-finite -> 0) + 1e-20 = the weights of the face draw;
 *   gs_surface_points: out [n,3] = (1-u) v0 + u (1-v) v1 + u v v2 of face face_id[i] (int64, e.g. torch.multinomial),
 *     (u, v) = (sqrt(r01[i,0]), r01[i,1]).
 * ---------------------------------------------------------------------------------- */
int gs_tri_area(const float* v_pos, const int32_t* tri, int64_t T, float* area, gs_stream_t stream);
/* ... with the face draw folded in: area_cdf [T] = inclusive prefix sum of gs_tri_area's output, r01x3 [n,3] uniform numbers
 * (u, v, face); face = first t with cdf[t] > r * cdf[T-1]; face_id [n] int64 WRITTEN (may be NULL). */
int gs_surface_points_cdf(const float* v_pos, const int32_t* tri, const float* area_cdf, int64_t T,
                          const float* r01x3, int64_t n, float* out, int64_t* face_id, gs_stream_t stream);
/* EnvironmentLight.update_pdf (render/light.py:46-59) in two launches: base [H,W,3] -> pdf [H,W] = max_c(base) sin(theta) / sum,
 * cols [H,W] = per-row CDF over the columns, rows [H,W] = CDF of the row masses (constant along x; the shader reads rows[:,0]).
 * H <= 1024; row_mass_scratch [H] floats. */
int gs_light_tables(const float* base, int64_t H, int64_t W, float* pdf, float* rows, float* cols,
                    float* row_mass_scratch, gs_stream_t stream);
int gs_surface_points(const float* v_pos, const int32_t* tri, const int64_t* face_id, const float* r01,
                      int64_t n, float* out, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * mSDF open / close regularisers   (replaces geometry/gshell_tets_geometry.py:326-358: Huber (delta 1) of
 *   clamp(msdf, min=-eps) to -eps over ALL grid values, and of clamp(msdf_boundary, max=eps) to +eps over the boundary
 *   vertices of the triangles some view saw)
 *   gs_boundary_weight: weight [n_boundary] f32 WRITTEN: 1 where mesh vertex n_watertight + k belongs to a triangle with
 *     flags[t] != 0 (the rasteriser's per-triangle visibility flags, u8), else 0 -- the reference's vis_mask (:344-348).
 *   fwd: out2 [2] f32 WRITTEN = (open_w * sum_i huber, close_w * sum_j weight_j huber)  (weight NULL = no close term).
 *   bwd: g_out2_dev [2] device; g_msdf [N], g_boundary [n_boundary] WRITTEN (either may be NULL).
 * ---------------------------------------------------------------------------------- */
int gs_boundary_weight(const int32_t* tri, const uint8_t* flags, int64_t T, int64_t n_watertight,
                       int64_t n_boundary, float* weight, gs_stream_t stream);
int gs_msdf_reg_fwd(const float* msdf, int64_t N, const float* msdf_boundary, const float* weight,
                    int64_t n_boundary, float eps, float open_w, float close_w, float* out2,
                    gs_stream_t stream);
int gs_msdf_reg_bwd(const float* msdf, int64_t N, const float* msdf_boundary, const float* weight,
                    int64_t n_boundary, float eps, float open_w, float close_w,
                    const float* g_out2_dev, float* g_msdf, float* g_boundary, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * SDF sign-consistency regulariser   (replaces compute_sdf_reg_loss,
 *   geometry/gshell_tets_geometry.py:33-39, evaluated over ALL grid edges every iteration :361-362)
 *   sdf [N] f32, edges [E,2] i32 (the static sorted edge list of gs_mtets_topo).
 *   fwd: gs_sdf_reg_partials(E) per-block partial sums of the loss and of the crossing-edge count;
 *        loss = sum(part_loss) / sum(part_count)   (0 crossing edges -> the caller returns 0).
 *   bwd: g_sdf [N] ACCUMULATED with g_scalar * d loss / d sdf; count_dev = device scalar sum(part_count).
 * ---------------------------------------------------------------------------------- */
int64_t gs_sdf_reg_partials(int64_t E);
int gs_sdf_reg_fwd(const float* sdf, const int32_t* edges, int64_t E, float* part_loss,
                   float* part_count, gs_stream_t stream);
int gs_sdf_reg_bwd(const float* sdf, const int32_t* edges, int64_t E, const float* g_scalar_dev,
                   const float* count_dev, float* g_sdf, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Per-pixel geometric normal   (replaces the face-normal interpolation of render/render.py:243-248:
 *   dr.interpolate(face_normals, rast, [[i,i,i]]) == gather of normalize(cross(v1-v0, v2-v0)) by triangle id)
 *   v_pos [V,3], tri [T,3] i32, rast [B,H,W,4] -> out [B,H,W,3] (0 where empty).
 *   bwd: g_v_pos [V,3] ACCUMULATED.
 * ---------------------------------------------------------------------------------- */
int gs_face_normal_fwd(const float* v_pos, int64_t V, const int32_t* tri, int64_t T, const float* rast,
                       int64_t B, int64_t H, int64_t W, float* out, gs_stream_t stream);
int gs_face_normal_bwd(const float* v_pos, int64_t V, const int32_t* tri, int64_t T, const float* rast,
                       int64_t B, int64_t H, int64_t W, const float* g_out, float* g_v_pos,
                       gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * shade() buffer assembly + background composite   (replaces the torch tail of render/render.py:74, :105-112, :160-186,
 *   the composite :352-359 / :417-433 and the division after ou.bilateral_denoiser, optixutils/ops.py:145-147)
 *   Inputs per pixel: rast [P,4] (coverage = rast.w > 0), tex / tex_jitter [P,6] (kd, ks and their jittered taps),
 *   n_interp / n_jitter [P,3] (smooth normal, its jittered tap), mask_tap [P] (mask * jittered mask), n_shade / n_geo [P,3],
 *   depth [P,2], diffuse / specular [P,cw] (cw = 4: (sum w c, sum w) of the bilateral filter; cw = 3: raw radiance),
 *   msdf_image [P] or NULL, background [bg_views (1 or B),H,W,3].
 *   out [P, 44 | 45] = shaded z_grad normal geometric_normal kd ks kd_grad ks_grad normal_grad diffuse_light specular_light
 *   [msdf_image], each 4-channel buffer with alpha = 1, composited over (background, 0) / zero.  bwd: all g_* WRITTEN.
 * ---------------------------------------------------------------------------------- */
int gs_shade_assemble_fwd(const float* rast, const float* tex, const float* tex_jitter, const float* n_interp,
                          const float* n_jitter, const float* mask_tap, const float* n_shade, const float* n_geo,
                          const float* depth, const float* diffuse, const float* specular, int cw_channels,
                          const float* msdf_image, const float* background, int bg_views, int64_t B, int64_t H,
                          int64_t W, float* out, gs_stream_t stream);
int gs_shade_assemble_bwd(const float* rast, const float* tex, const float* tex_jitter, const float* n_interp,
                          const float* n_jitter, const float* mask_tap, const float* n_shade, const float* n_geo,
                          const float* depth, const float* diffuse, const float* specular, int cw_channels,
                          const float* msdf_image, int64_t B, int64_t H, int64_t W, const float* g_out, float* g_tex,
                          float* g_tex_jitter, float* g_n_interp, float* g_n_jitter, float* g_n_shade, float* g_n_geo,
                          float* g_diffuse, float* g_specular, float* g_msdf_image, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * SDF network, fused forward   (replaces MLP.forward + Embedding.forward, geometry/mlp.py:32-40,
 *   geometry/embedding.py:22-39, as called on the whole grid at geometry/gshell_tets_geometry.py:194)
 *   x [N,3] -> out [N];  d_hidden = 256, d_out = 1, Softplus(beta=100).
 *   packed (gs_sdf_mlp_packed_floats floats) = [Wt_0 | b_0 | Wt_1 | b_1 | ... | w_out(256) | b_out], where
 *   Wt_l is the TRANSPOSE of torch's Linear.weight (k-major [K_l, 256]); K_0 = E padded to a multiple of 8 with zero
 *   rows (E = 3 (2 n_freq + 1)); K_l = 256, plus the zero-padded E rows of the embedding for the skip layer
 *   (input order [h | emb], geometry/mlp.py:37).  skip_layer: 1-based index among the hidden layers' Linear
 *   modules counted from the first Linear as 0 (reference skip_in=[3] -> skip_layer = 4), or -1.
 *   ORACLE KERNEL (round 6): the exact-fp32 MFMA forward is what the fp16-pair kernels below are checked against; it is built into
 *   lib/variants/oracles.so only -- in the shipped library gs_sdf_mlp_fwd fails with a message naming the variant (gs_sdf_mlp_packed_floats works).
 * ---------------------------------------------------------------------------------- */
int64_t gs_sdf_mlp_packed_floats(int n_freq, int n_hidden, int skip_layer);
int gs_sdf_mlp_fwd(const float* x, int64_t N, const float* packed, int n_freq, int n_hidden,
                   int skip_layer, float* out, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * SDF network on the f16 matrix path with fp32-class accuracy ("h2": every operand is a pair of fp16 pieces
 *   v = hi + lo/2048, every product three v_mfma_f32_32x32x16_f16 with fp32 accumulation; csrc/mlp_h2.hip).
 *   Same function as gs_sdf_mlp_fwd (geometry/mlp.py:32-40 over the whole grid), ~4x faster; the exact-fp32 kernel
 *   above stays as its oracle.
 *   gs_sdf_mlp_h2_pack: weights / biases = HOST arrays of n_hidden + 2 DEVICE pointers in torch's own layout
 *   (Linear.weight [out,in] row-major, Linear.bias), hidden-producing layers first, the output layer last ->
 *   packed (gs_sdf_mlp_h2_packed_bytes bytes, 16-byte aligned) WRITTEN: fragment-major fp16 pairs + fp32 biases.
 *   One launch per optimisation step (the weights change every step).
 * ---------------------------------------------------------------------------------- */
int64_t gs_sdf_mlp_h2_packed_bytes(int n_freq, int n_hidden, int skip_layer);
int gs_sdf_mlp_h2_pack(const float* const* weights, const float* const* biases, int n_freq, int n_hidden,
                       int skip_layer, void* packed, uint32_t* status /* [2] device words, or NULL */,
                       gs_stream_t stream);
/* status (device, uint32 [2], zeroed by the caller; NULL = unchecked):
 *   [0] != 0  a non-finite signed distance left the network, or a weight lies beyond the fp16 range: the fp16-pair arithmetic
 *             has overflowed (activations >= 65 504 become inf in the split) -- the caller must fall back to gs_sdf_mlp_fwd
 *             (exact fp32; reference geometry/mlp.py:32-40 has no range limit);
 *   [1]       bits of max |three-product value - one-product value| over the rows of gs_sdf_mlp_h2_refine_rows;
 *   [2]       (gs_sdf_mlp_h2_refine_rows with tau > 0 only: the array then has THREE words) bits of
 *             max |new - old| / max(tau, |new|): the largest fraction of its sign margin that the first pass's error used up on
 *             any re-evaluated row -- near-surface rows (margin tau) and the audit sample of far rows (margin |sdf|) alike. */
int gs_sdf_mlp_fwd_h2(const float* x, int64_t N, const void* packed, int n_freq, int n_hidden,
                      int skip_layer, float* out, uint64_t* occ_bits /* [ceil(N/64)] sign bits WRITTEN, or NULL */,
                      uint32_t* status, gs_stream_t stream);
/* Two-pass evaluation of the full grid (same function, a third of the matrix work where the value cannot matter):
 *   1. gs_sdf_mlp_fwd_h1: ONE fp16 product per algorithmic product (operands rounded to 2^-11): out, sign bits.  Error ~1e-4.
 *   2. gs_mtets_flag_refine_rows (extraction section): flags both end points of every grid edge that changes sign or has an end
 *      point with |out| < tau;  gs_compact_rows turns the flags into a row list on the device.
 *   3. gs_sdf_mlp_h2_refine_rows: those rows again with the three-product arithmetic of gs_sdf_mlp_fwd_h2 -- bit-identical
 *      values (a row's arithmetic does not depend on its tile) --, written over out, sign bits corrected, status[1] =
 *      max |new - old| (the measured error of pass 1 on the rows that matter; the caller checks it against tau).
 *      AUDIT: the caller also flags a rotating pseudo-random sample of ALL rows (every k-th row, another residue class each call), so
 *      rows that were NOT selected are measured too: status[2] (see above) must stay below 1 / safety for the proof's hypothesis
 *      "error below the sign margin at every row" to be an observed fact on the unrefined rows as well, not an extrapolation.
 *   If the pass-1 error is below tau at every vertex, the result equals gs_sdf_mlp_fwd_h2's in every sign and at both end
 *   points of every sign-crossing edge -- all the reference consumes (gshell_tets.py:250, :277-290; gshell_tets_geometry.py:33-39). */
int gs_sdf_mlp_fwd_h1(const float* x, int64_t N, const void* packed, int n_freq, int n_hidden,
                      int skip_layer, float* out, uint64_t* occ_bits, uint32_t* status, gs_stream_t stream);
/* Kernel behind gs_sdf_mlp_fwd_h1: 0 = the shipped kernel (activations in LDS, weights through L1); 1 = activations register-resident across the
 * layers, weights staged through LDS once per 256 rows (k_h1r_fwd, round 5: correct, on par on MI355X) -- an ALTERNATE DESIGN that
 * exists in lib/variants/oracles.so only (GS_ORACLE_KERNELS): the shipped library answers impl == 1 with -1 and gs_last_error().  Same function and
 * arithmetic class (geometry/mlp.py:32-40 with fp16 operands and fp32 accumulation); sums are taken in a different order.  Select BEFORE
 * gs_sdf_mlp_h2_pack (the packer writes kernel 1's fragments only while it is selected).  impl < 0 only queries.  Returns the previous setting. */
int gs_sdf_mlp_h1_impl(int impl);
int gs_sdf_mlp_h2_refine_rows(const float* x, const int32_t* rows, int64_t cap, const int64_t* count_dev,
                              const void* packed, int n_freq, int n_hidden, int skip_layer, float* out,
                              uint64_t* occ_bits, uint32_t* status /* [2], or [3] when tau > 0 */, float tau,
                              gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * SDF network, gradients   (replaces autograd through geometry/mlp.py:32-40 as used by
 *   geometry/gshell_tets_geometry.py:194 [d loss / d sdf of the grid rows that carry gradient] and the eikonal term
 *   :302-324 [autograd.grad(create_graph=True) + second backward])          -- csrc/mlp_h2.hip
 *   Three launches over n (virtual) rows, all through HBM-resident fp32 planes owned by the caller:
 *     save_fwd : recompute the rows through the h2 chain and save every layer's output  A [n_hidden+1][Rpad][256], EMB [Rpad][48]
 *     bwd      : reverse chain -> D [n_hidden+1][Rpad][256] = d loss / d pre-activation; mode 1 also d loss / d x
 *     wgrad    : dW_l += D_l^T [A_{l-1} | EMB], db_l += column sums of D_l   (fp32 MFMA, float atomics across row strips)
 *   mode 1 (ROWS): rows[0..n) index x [N,3] (row-sparse backward).   mode 2 (EIK): x = n sample points; every sample is FOUR
 *   virtual rows (value, d/dx, d/dy, d/dz): forward-mode tangents ride the same GEMMs, out[64 t + 16 c + i] = df/dx_{c-1}
 *   of sample 16 t + i (c = 1..3), and ONE reverse pass over the virtual rows yields the parameter gradient of any loss of
 *   grad_x f (g_out = d loss / d out on the tangent rows, 0 on the value rows).  Rpad = gs_sdf_mlp_h2_rows_padded(mode, n).
 *   dW / db: HOST arrays of n_hidden + 2 DEVICE pointers (torch layout, output layer last), ACCUMULATED; db[last] untouched.
 *   mode 2 is the forward-mode formulation of the eikonal term, superseded in round 4 by reverse over reverse (gs_sdf_eikonal_rr_* below) and since
 *   round 6 an ORACLE formulation: its save_fwd / bwd instantiations exist in lib/variants/oracles.so only; the shipped library refuses mode 2.
 * ---------------------------------------------------------------------------------- */
int64_t gs_sdf_mlp_h2_rows_padded(int mode, int64_t n);
/*   Without a host sync (mode 1): gs_compact_rows turns d loss / d sdf [N] into (rows, g_rows, count) entirely on the device;
 *   n is then a CAPACITY (an upper bound on the rows with gradient, e.g. 2 x crossing edges) and n_dev -> count_dev[0];
 *   tiles past the device-side count exit.  n_dev = NULL: n is the exact count (the caller synchronised). */
int64_t gs_compact_rows_scratch_bytes(int64_t N);
int gs_compact_rows(const float* g, int64_t N, int64_t cap, void* scratch, int32_t* rows, float* g_rows /* may be NULL */,
                    int64_t* count_dev /* [2]: min(count, cap), overflow flag */, gs_stream_t stream);
/* The same over every `stride`-th float of g (element i = g[i * stride]): e.g. the triangle-id channel of a rasteriser output [B,H,W,4], stride 4 --
 * the covered-pixel list of a frame without a channel copy and without a host synchronisation (the caller reads count_dev when it needs the count). */
int gs_compact_rows_strided(const float* g, int64_t stride, int64_t N, int64_t cap, void* scratch, int32_t* rows,
                            float* g_rows, int64_t* count_dev, gs_stream_t stream);
int gs_sdf_mlp_h2_save_fwd(int mode, const float* x, const int32_t* rows, int64_t n, const int64_t* n_dev,
                           const void* packed, int n_freq, int n_hidden, int skip_layer, float* A_save,
                           float* EMB_save, float* out, gs_stream_t stream);
/* A_save / D_save are opaque scratch planes of layers * Rpad * 256 floats each, laid out [layer][32-row slab][256][32 rows]
 * (row index fastest: every access of the three kernels is a whole line); written by _save_fwd / _bwd, read by _bwd / _wgrad. */
int gs_sdf_mlp_h2_bwd(int mode, const float* g_out, const int32_t* rows, int64_t n, const int64_t* n_dev,
                      const void* packed, int n_freq, int n_hidden, int skip_layer, const float* A_save,
                      const float* EMB_save, float* D_save, float* g_x, gs_stream_t stream);
int gs_sdf_mlp_h2_wgrad(int mode, const float* g_out, int64_t n, const int64_t* n_dev, int n_freq, int n_hidden, int skip_layer,
                        const float* A_save, const float* EMB_save, const float* D_save,
                        float* const* dW, float* const* db,
                        int exact_fp32 /* 0: bf16-pair operands on the bf16 matrix path (default); 1: fp32 MFMA */,
                        gs_stream_t stream);
/* eikonal term on the output column of a mode-2 (value + 3 tangents) gs_sdf_mlp_h2_save_fwd pass of n samples:
 *   loss [1] WRITTEN = sum_i (|grad f(p_i)| - 1)^2  (geometry/gshell_tets_geometry.py:302-324);
 *   g_unit [rows_padded] WRITTEN = d loss / d (virtual-row outputs): multiply by the upstream scalar and hand it to
 *   gs_sdf_mlp_h2_bwd / _wgrad as g_out. */
int gs_sdf_eikonal_loss(const float* out_rows, int64_t n, int64_t rows_padded, float* loss,
                        float* g_unit, gs_stream_t stream);
/* The same term by REVERSE over reverse -- the formulation of the reference's autograd (autograd.grad(..., create_graph=True) followed by
 * a second backward, geometry/gshell_tets_geometry.py:302-324) -- at half the row passes of the tangent-row form above (default since round 4):
 *   _fwd: value pass over the n samples x [n,3], reverse chain with g_out = 1  ->  grad_f [n,3] WRITTEN = grad_x f,
 *         loss [1] WRITTEN = sum_i (|grad_f_i| - 1)^2, gbar [n,3] WRITTEN = d loss / d grad_f;
 *   _bwd: tangent pass in direction g_up * gbar (the adjoint of the reverse chain), reverse chain with the second-order source, one weight-gradient
 *         launch: dW / db (host arrays of n_hidden + 2 device pointers as for gs_sdf_mlp_h2_wgrad) ACCUMULATED with the gradient of g_up * loss;
 *         g_up = DEVICE scalar (the upstream gradient of the loss: no host read, no scaling launch).
 * Scratch owned by the caller, alive from _fwd to _bwd, with Rpad = gs_sdf_eikonal_rr_rows_padded(n), L = n_hidden + 1:
 *   A_all, D_all [L * 2 Rpad * 256] floats, EMB_all [2 Rpad * 48], g_all [2 Rpad]. */
int64_t gs_sdf_eikonal_rr_rows_padded(int64_t n);
int gs_sdf_eikonal_rr_fwd(const float* x, int64_t n, const void* packed, int n_freq, int n_hidden, int skip_layer,
                          float* A_all, float* EMB_all, float* D_all, float* g_all, float* grad_f, float* gbar, float* loss,
                          gs_stream_t stream);
int gs_sdf_eikonal_rr_bwd(int64_t n, const void* packed, int n_freq, int n_hidden, int skip_layer, float* A_all, float* EMB_all,
                          float* D_all, const float* g_all, const float* gbar, const float* g_up, float* const* dW,
                          float* const* db, int exact_fp32, gs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * G-FlexiCubes topology   (replaces the index machinery of GShellFlexiCubes.__call__,
 *   geometry/gshell_flexicubes.py:136-230: _identify_surf_cubes :334, _get_case_id :266, _identify_surf_edges :309,
 *   the edge-group tables of _compute_vd :406-485 and the quad gathering / splitting of _triangulate :493-522)
 *   Static per-grid tables come from the caller: edges [E,2] i32 (unique ORDERED cube edges, lexicographic),
 *   cube_edge [F,12] i32, ncubes [E] u8, inc [E,4] i32 (incident cube*12+e, ascending, -1 padded).
 *   gs_flexi_classify  : case_id / num_vd / n_ent [F] u8 (0 for non-surface cubes); scratch 2F bytes; regular grid res0*res1*res2 = F.
 *   gs_flexi_edge_flags: flags [E] u8: bit0 crossing, bit1 emits a quad (crossing & 4 cubes), bit2 s[first end] > 0.
 *   gs_flexi_entries   : edge-group entries at ent_base[c] (order: dual vertex, slot) and dual vertex ids vd_base[c] + j:
 *                        ent_vd / ent_edge / ent_cube / ent_e [n_ent] i32, vd_idx_map [F,12] i32 (caller pre-fills -1), vd_cube [n_vd].
 *   gs_flexi_quads     : faces [2Q,3] i64 (+ optional i32 copy) at 6*qrank[e]; vd_gamma [n_vd] f32 = normalised gamma per dual vertex.
 * ---------------------------------------------------------------------------------- */
int gs_flexi_classify(const float* s, const int32_t* cubes_fx8, int64_t F, int64_t res0, int64_t res1, int64_t res2,
                      uint8_t* scratch_2F, uint8_t* case_id, uint8_t* num_vd, uint8_t* n_ent, gs_stream_t stream);
int gs_flexi_edge_flags(const float* s, const int32_t* edges_ex2, const uint8_t* ncubes, int64_t E, uint8_t* flags,
                        gs_stream_t stream);
int gs_flexi_entries(const uint8_t* case_id, const uint8_t* num_vd, const int32_t* vd_base, const int32_t* ent_base,
                     const int32_t* cube_edge, int64_t F, int32_t* ent_vd, int32_t* ent_edge, int32_t* ent_cube,
                     int32_t* ent_e, int32_t* vd_idx_map, int32_t* vd_cube,
                     int32_t* vd_start /* [n_vd + 1] first entry of every dual vertex, or NULL */, int64_t n_vd,
                     int64_t n_entries, gs_stream_t stream);
int gs_flexi_quads(const uint8_t* flags, const int32_t* qrank, const int32_t* inc_ex4, const int32_t* vd_idx_map,
                   const float* vd_gamma, int64_t E, int64_t* faces, int32_t* faces_i32, gs_stream_t stream);


/* ------------------------------------------------------------------------------------
 * G-FlexiCubes float path   (replaces _compute_vd geometry/gshell_flexicubes.py:387-485, _compute_reg_loss :232-240, the
 *   boundary vertices of _triangulate_msdf :554-599, and the cumsum / masked-scatter ops behind the reference orderings)
 *   gs_flexi_ranks  : vd_base / ent_base [F] i32, qrank [E] i32 in the reference's orderings (dual vertices by (num_vd group,
 *                     cube, j), entries by (group, cube, j, slot), quads flipped-first by edge id) from three launches;
 *                     totals_dev [16] i64: [10] = n_vd, [11] = n_entries, [12] = n_quads; scratch gs_flexi_ranks_scratch_bytes.
 *   gs_flexi_vd_fwd : one thread per dual vertex over its contiguous entries -> vd [n_vd,3], nu_d, nu_d_sv [n_vd] (the in-place
 *                     index_add_ quirk of :476-477 reproduced), l_dev [n_entries].  beta [F,12], alpha [F,8] are the NORMALISED
 *                     weights (tanh / sigmoid stay with the caller).
 *   gs_flexi_vd_bwd : adjoint; g_x [N,3], g_s, g_nu [N], g_alpha [F,8] ACCUMULATED (float atomics), g_beta [F,12] slots WRITTEN --
 *                     the caller zero-fills all five.
 *   gs_flexi_cut_*  : boundary vertex per (cut triangle, edge): bverts [n,3] = interp_nonan(nu_d, vd), bnu [n] =
 *                     interp_nonan(nu_d_sv detached, nu_d_sv); bwd ACCUMULATES into g_vd, g_nu_d, g_nu_d_sv.
 * ---------------------------------------------------------------------------------- */
int64_t gs_flexi_ranks_scratch_bytes(int64_t F, int64_t E);
int gs_flexi_ranks(const uint8_t* num_vd, const uint8_t* n_ent, const uint8_t* flags, int64_t F, int64_t E,
                   void* scratch, int64_t* totals_dev, int32_t* vd_base, int32_t* ent_base, int32_t* qrank,
                   gs_stream_t stream);
int gs_flexi_vd_fwd(const float* x, const float* s, const float* nu, const float* beta_fx12, const float* alpha_fx8,
                    const int32_t* edges_ex2, const int32_t* ent_edge, const int32_t* ent_cube, const int32_t* ent_e,
                    const int32_t* vd_start, int64_t n_vd, float* vd, float* nu_d, float* nu_d_sv, float* l_dev,
                    gs_stream_t stream);
int gs_flexi_vd_bwd(const float* x, const float* s, const float* nu, const float* beta_fx12, const float* alpha_fx8,
                    const int32_t* edges_ex2, const int32_t* ent_edge, const int32_t* ent_cube, const int32_t* ent_e,
                    const int32_t* vd_start, int64_t n_vd, const float* g_vd, const float* g_nu_d,
                    const float* g_nu_d_sv, const float* g_l_dev, float* g_x, float* g_s, float* g_nu, float* g_beta,
                    float* g_alpha, gs_stream_t stream);
int gs_flexi_cut_fwd(const int64_t* pa, const int64_t* pb, int64_t n, const float* vd, const float* nu_d,
                     const float* nu_d_sv, float* bverts, float* bnu, gs_stream_t stream);
int gs_flexi_cut_bwd(const int64_t* pa, const int64_t* pb, int64_t n, const float* vd, const float* nu_d,
                     const float* nu_d_sv, const float* g_bverts, const float* g_bnu, float* g_vd, float* g_nu_d,
                     float* g_nu_d_sv, gs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GSHELL_HIP_H */

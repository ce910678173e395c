"""G-MarchingTets extractor -- drop-in for the reference's geometry/gshell_tets.py.

`GShell_Tets.__call__(pos_nx3, sdf_n, msdf_n, tet_fx4)` keeps the reference signature and
return tuple (gshell_tets.py:245, :426-443) but runs as hand-written HIP kernels
(gshell_amd/csrc/mtets.hip) through the C ABI in include/gshell_hip.h:

    (verts_aug, faces_aug, None, None, v_tng_aug, extra)

Differences that are deliberate (documented in DESIGN.md):
  * v_tng_aug / extra['v_tng_watertight'] back-propagate (gs_mtets_tangents_bwd: compute_tangents + auto_normals + the boundary
    interpolation, gshell_tets.py:9-78, :318-319, :375-380), although the reference's training path discards the tangents
    (gshell_tets_geometry.py:206-208, render.py:264-267).
  * faces are additionally available as int32 (`extra['faces_i32']`) for the rasteriser.
"""
import ctypes
import math

import torch

from .. import _lib
from .._lib import c_int64, c_void_p, check, ptr, stream


class TetTopology:
    """Static per-grid topology living in HBM (sorted unique edges, tet->edge table)."""

    def __init__(self, tet_fx4: torch.Tensor, num_verts: int, uv_tets: int = None):
        """`uv_tets`: the tet count the tangents' uv atlas is sized by (reference :301-309: `num_tets` of the WHOLE grid even when the valid tets were
        pre-filtered) -- default: this topology's own tet count."""
        if tet_fx4.dtype != torch.int64:
            tet_fx4 = tet_fx4.long()
        tet_fx4 = tet_fx4.contiguous()
        if not tet_fx4.is_cuda:
            raise _lib.GShellHipError(f"the tet grid must live in HBM (got device {tet_fx4.device}); the HIP extraction has no CPU fallback")
        self.device = tet_fx4.device
        self.N, self.F = int(num_verts), int(tet_fx4.shape[0])
        self._h = c_void_p(0)
        with torch.cuda.device(self.device):
            check(_lib.lib().gs_mtets_topo_create(ptr(tet_fx4, torch.int64, "tet_fx4"), c_int64(self.F), c_int64(self.N),
                                                  stream(), ctypes.byref(self._h)), "gs_mtets_topo_create")
        n, f, e = c_int64(), c_int64(), c_int64()
        self._edges_ptr, self._tet_ptr = c_void_p(), c_void_p()
        check(_lib.lib().gs_mtets_topo_info(self._h, ctypes.byref(n), ctypes.byref(f), ctypes.byref(e),
                                            ctypes.byref(self._edges_ptr), ctypes.byref(self._tet_ptr)))
        self.E = int(e.value)
        self._edges = None
        self.sign_epoch = 0          # bumped by every writer of the occupancy bits
        f_uv = self.F if uv_tets is None else int(uv_tets)
        nuv = int(math.ceil(math.sqrt((2 * f_uv + 1) // 2))) if f_uv > 0 else 1
        self.Nuv = nuv
        # torch.linspace(0, 1 - 1/N, N): the uv atlas axis of the reference's map_uv (gshell_tets.py:211-216)
        self.uv_lin = torch.linspace(0, 1 - (1 / nuv), nuv, dtype=torch.float32, device=self.device)

    @property
    def handle(self):
        return self._h

    def occ_bits_ptr(self) -> int:
        """Device address of the [ceil(N/64)] uint64 occupancy bits (written by k_occ_bits, or by the SDF network's epilogue)."""
        bits, words = c_void_p(), c_int64()
        check(_lib.lib().gs_mtets_occ_bits(self._h, ctypes.byref(bits), ctypes.byref(words)))
        return int(bits.value)

    def edges(self) -> torch.Tensor:
        """[E,2] int32, lexicographically sorted unique (min,max) grid edges (a copy)."""
        if self._edges is None:
            out = torch.empty((self.E, 2), dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                check(_lib.lib().gs_memcpy_d2d(ptr(out), self._edges_ptr, c_int64(out.numel() * 4), stream()))
            self._edges = out
        return self._edges

    def __del__(self):
        try:
            if self._h:
                _lib.lib().gs_mtets_topo_destroy(self._h)
                self._h = c_void_p(0)
        except Exception:
            pass


class _MarchingTetsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, sdf, msdf, topo: TetTopology, want_tangents: bool, presigned: bool = False):
        L = _lib.lib()
        dev = pos.device
        pos_c, sdf_c, msdf_c = pos.detach().contiguous().float(), sdf.detach().contiguous().float(), msdf.detach().contiguous().float()
        if pos_c.shape[0] != topo.N or sdf_c.numel() != topo.N or msdf_c.numel() != topo.N:
            raise _lib.GShellHipError(f"field sizes {tuple(pos_c.shape)}, {sdf_c.numel()}, {msdf_c.numel()} do not match the grid (N={topo.N})")
        counts = (c_int64 * 16)()
        with torch.cuda.device(dev):
            if presigned:      # sign bits already in the topology's array (fused geometry front end)
                check(L.gs_mtets_count_presigned(topo.handle, ptr(sdf_c, torch.float32, "sdf"), ptr(msdf_c, torch.float32, "msdf"), stream(), counts),
                      "gs_mtets_count_presigned")
            else:
                check(L.gs_mtets_count(topo.handle, ptr(pos_c, torch.float32, "pos"), ptr(sdf_c, torch.float32, "sdf"),
                                       ptr(msdf_c, torch.float32, "msdf"), stream(), counts), "gs_mtets_count")
            V, M1, M2, T, V_aug = counts[0], counts[1], counts[2], counts[9], counts[10]
            f32 = dict(dtype=torch.float32, device=dev)
            verts_aug = torch.empty((V_aug, 3), **f32)
            msdf_aug = torch.empty((V_aug,), **f32)
            verts_wt = torch.empty((V, 3), **f32)
            faces_wt = torch.empty((M1 + 2 * M2, 3), dtype=torch.int64, device=dev)
            faces_aug = torch.empty((T, 3), dtype=torch.int64, device=dev)
            faces_i32 = torch.empty((T, 3), dtype=torch.int32, device=dev)
            vert_ab = torch.empty((V, 2), dtype=torch.int32, device=dev)
            used_wt = torch.empty((V,), dtype=torch.uint8, device=dev)
            poly = torch.empty((3 * M1 + 4 * M2,), dtype=torch.int32, device=dev)
            cut_code = torch.empty((M1 + M2,), dtype=torch.uint8, device=dev)
            tet_id = torch.empty((M1 + M2,), dtype=torch.int32, device=dev)
            sign_code = torch.empty((M1 + M2,), dtype=torch.uint8, device=dev)
            grp_rank = torch.empty((M1 + M2,), dtype=torch.int32, device=dev)
            check(L.gs_mtets_fill(topo.handle, ptr(pos_c), ptr(sdf_c), ptr(msdf_c), ptr(verts_aug), ptr(msdf_aug), ptr(verts_wt),
                                  ptr(faces_wt), ptr(faces_aug), ptr(faces_i32), ptr(vert_ab), ptr(used_wt), ptr(poly),
                                  ptr(cut_code), ptr(tet_id), ptr(sign_code), ptr(grp_rank), stream()), "gs_mtets_fill")
            v_tng_aug = torch.zeros((V_aug, 3), **f32)
            scratch = None
            if want_tangents and V > 0:
                scratch = torch.empty((V, 7), **f32)          # per-vertex face-normal sum, face-tangent sum, face count: kept for the backward
                check(L.gs_mtets_tangents(c_int64(V), c_int64(M1), c_int64(M2), c_int64(topo.F), ptr(verts_wt), ptr(faces_wt),
                                          ptr(msdf_aug), ptr(poly), ptr(topo.uv_lin), c_int64(topo.Nuv), ptr(scratch),
                                          ptr(v_tng_aug), stream()), "gs_mtets_tangents")
        # (the tangent pass's tensors -- the OUTPUT v_tng_aug among them -- go through save_for_backward too: as plain attributes of ctx they formed an
        # output -> grad_fn -> ctx -> output cycle that only the cyclic GC frees, and escaped autograd's in-place-modification check)
        tng = (faces_wt, scratch, v_tng_aug, topo.uv_lin) if scratch is not None else ()
        ctx.save_for_backward(pos_c, sdf_c, msdf_c, verts_wt, msdf_aug, vert_ab, used_wt, poly, cut_code, *tng)
        ctx.dims = (topo.N, V, M1, M2)
        ctx.in_shapes = (pos.shape, sdf.shape, msdf.shape)
        # tangents: differentiable (compute_tangents + auto_normals + boundary interpolation, gshell_tets.py:9-78, :318-319, :375-380)
        ctx.tng = topo.Nuv if scratch is not None else None          # (an int; the tensors are saved above)
        ctx.mark_non_differentiable(faces_wt, faces_aug, faces_i32, tet_id)
        if scratch is None:
            ctx.mark_non_differentiable(v_tng_aug)
        ctx.set_materialize_grads(False)      # outputs nobody differentiates (verts_wt in training) arrive as None, not as zero tensors
        return verts_aug, msdf_aug, verts_wt, faces_aug, faces_wt, faces_i32, v_tng_aug, tet_id

    @staticmethod
    def backward(ctx, g_verts_aug, g_msdf_aug, g_verts_wt, g_faces_aug=None, g_faces_wt=None, g_faces_i32=None, g_tng_aug=None, *_unused):
        pos, sdf, msdf, verts_wt, msdf_aug, vert_ab, used_wt, poly, cut_code = ctx.saved_tensors[:9]
        N, V, M1, M2 = ctx.dims
        dev = pos.device
        flat = torch.zeros((5 * N,), dtype=torch.float32, device=dev)          # one fill, three views
        g_pos, g_sdf, g_msdf = flat[:3 * N].view(N, 3), flat[3 * N:4 * N], flat[4 * N:]
        if ctx.tng is None:
            g_tng_aug = None
        if V > 0 and not (g_verts_aug is None and g_msdf_aug is None and g_verts_wt is None and g_tng_aug is None):
            def prep(g):
                return None if g is None else g.contiguous().float()
            ga, gm, gw = prep(g_verts_aug), prep(g_msdf_aug), prep(g_verts_wt)
            g_mv = None
            if g_tng_aug is not None:
                faces_wt, acc, v_tng_aug, uv_lin = ctx.saved_tensors[9:13]
                Nuv = ctx.tng
                gt = prep(g_tng_aug)
                g_vw_t = torch.empty((V, 3), dtype=torch.float32, device=dev)
                g_mv = torch.empty((V,), dtype=torch.float32, device=dev)
                with torch.cuda.device(dev):
                    work = torch.empty((V, 9), dtype=torch.float32, device=dev)
                    check(_lib.lib().gs_mtets_tangents_bwd(c_int64(V), c_int64(M1), c_int64(M2), ptr(verts_wt), ptr(faces_wt), ptr(msdf_aug), ptr(poly),
                                                           ptr(uv_lin), c_int64(Nuv), ptr(acc), ptr(v_tng_aug), ptr(gt, torch.float32, "g_tng_aug"), ptr(work),
                                                           ptr(g_vw_t), ptr(g_mv), stream()), "gs_mtets_tangents_bwd")
                gw = g_vw_t if gw is None else gw + g_vw_t
            with torch.cuda.device(dev):
                scratch = torch.empty((V, 5), dtype=torch.float32, device=dev)
                check(_lib.lib().gs_mtets_bwd(c_int64(N), c_int64(V), c_int64(M1), c_int64(M2), ptr(pos), ptr(sdf), ptr(msdf),
                                              ptr(verts_wt), ptr(msdf_aug), ptr(vert_ab), ptr(used_wt), ptr(poly), ptr(cut_code),
                                              ptr(ga), ptr(gm), ptr(gw), ptr(g_mv), ptr(scratch), ptr(g_pos), ptr(g_sdf), ptr(g_msdf),
                                              stream()), "gs_mtets_bwd")
        ps, ss, ms = ctx.in_shapes
        return g_pos.reshape(ps), g_sdf.reshape(ss), g_msdf.reshape(ms), None, None, None


class GShell_Tets:
    """Same call surface as the reference class (gshell_tets.py:80, :245)."""

    def __init__(self, compute_tangents: bool = True):
        self.compute_tangents = compute_tangents
        self._topo_cache = {}

    def topology(self, tet_fx4: torch.Tensor, num_verts: int) -> TetTopology:
        key = (tet_fx4.data_ptr(), tuple(tet_fx4.shape), str(tet_fx4.device), int(num_verts))
        topo = self._topo_cache.get(key)
        if topo is None:
            topo = TetTopology(tet_fx4, num_verts)
            self._topo_cache = {key: topo}   # one grid at a time, like the reference
        return topo

    def __call__(self, pos_nx3, sdf_n, msdf_n, tet_fx4, output_watertight_template=True):
        if not output_watertight_template:
            return self._without_watertight_template(pos_nx3, sdf_n, msdf_n, tet_fx4)
        topo = self.topology(tet_fx4, pos_nx3.shape[0])
        # fused geometry front end: an sdf tensor that comes straight out of the SDF-network kernel carries the tag of the
        # occupancy bits its epilogue wrote into THIS topology; any later writer of those bits invalidates the tag (epoch)
        tag = getattr(sdf_n, '_gs_presigned', None)
        presigned = tag is not None and tag[0] is topo and tag[1] == topo.sign_epoch and tag[2] == sdf_n._version
        if not presigned:
            topo.sign_epoch += 1
        verts_aug, msdf_aug, verts_wt, faces_aug, faces_wt, faces_i32, v_tng_aug, tet_id = _MarchingTetsFn.apply(
            pos_nx3, sdf_n, msdf_n, topo, self.compute_tangents, presigned)
        V = verts_wt.shape[0]
        extra = {
            'n_verts_watertight': V,
            'vertices_watertight': verts_wt,
            'faces_watertight': faces_wt,
            'v_tng_watertight': v_tng_aug[:V],
            'msdf': msdf_aug,
            'msdf_watertight': msdf_aug[:V],
            'msdf_boundary': msdf_aug[V:],
            # extras of this implementation
            'faces_i32': faces_i32,
            'polygon_tet_id': tet_id,
        }
        return verts_aug, faces_aug, None, None, v_tng_aug, extra

    def _without_watertight_template(self, pos_nx3, sdf_n, msdf_n, tet_fx4):
        """output_watertight_template=False (reference gshell_tets.py:256-263, :436-441): tets whose four mSDF values are all <= 0 are dropped before anything
        else, so the edge set -- and with it the vertex numbering -- is that of the remaining tets, and `extra` carries the three mSDF entries only.  No call
        site of the reference passes False, so this is the plain route: the surviving tets form a topology of their own, built for this call (its edge list is
        sorted on the device every time; the static topology of the default mode is what makes that mode fast), and the same kernels run on it."""
        with torch.no_grad():
            keep = (msdf_n.detach().reshape(-1)[tet_fx4.reshape(-1)].reshape(-1, 4) > 0).sum(-1) > 0
            tets = tet_fx4[keep].contiguous()
        if tets.shape[0] == 0:        # nothing survives: the reference's gathers over empty index sets (:264-443)
            f32 = dict(dtype=torch.float32, device=pos_nx3.device)
            z3, z1 = torch.zeros((0, 3), **f32), torch.zeros((0,), **f32)
            extra = {'msdf': z1, 'msdf_watertight': z1, 'msdf_boundary': z1, 'faces_i32': torch.zeros((0, 3), dtype=torch.int32, device=pos_nx3.device)}
            return z3, torch.zeros((0, 3), dtype=torch.long, device=pos_nx3.device), None, None, z3, extra
        topo = TetTopology(tets, pos_nx3.shape[0], uv_tets=tet_fx4.shape[0])
        topo.sign_epoch += 1
        verts_aug, msdf_aug, verts_wt, faces_aug, faces_wt, faces_i32, v_tng_aug, tet_id = _MarchingTetsFn.apply(
            pos_nx3, sdf_n, msdf_n, topo, self.compute_tangents, False)
        V = verts_wt.shape[0]
        extra = {'msdf': msdf_aug, 'msdf_watertight': msdf_aug[:V], 'msdf_boundary': msdf_aug[V:],
                 # extras of this implementation (polygon_tet_id indexes the SURVIVING tets)
                 'faces_i32': faces_i32, 'polygon_tet_id': tet_id}
        return verts_aug, faces_aug, None, None, v_tng_aug, extra

    @torch.no_grad()
    def marching_from_auggrid(self, pos_nx3, sdf_n, tet_fx4, sorted_tet_edges_fx6x2, coeff_sdf_interp, verts_discretized,
                              midpoint_msdf_sign_n, occgrid):
        """Generative-decode extraction; same signature and 9-tuple as the reference
        (gshell_tets.py:446-629):

            (verts_aug, faces_aug, None, None, v_tng_aug, verts, valid_tet_gidx, msdf_vert_aug, msdf_vert)

        `sorted_tet_edges_fx6x2` must be the per-tet (min,max) edges in the base order
        01 02 03 12 13 23 -- the layout of the npz's 'tet_edges' -- because the static
        topology of this implementation is derived from `tet_fx4` in that order; it is
        checked once per grid.  Grids are cubic: coeff / msdf-sign [G,G,G], occgrid [G2,G2,G2]."""
        L = _lib.lib()
        dev = pos_nx3.device
        topo = self.topology(tet_fx4, pos_nx3.shape[0])
        if sorted_tet_edges_fx6x2 is not None and getattr(topo, "_edges_checked", None) != sorted_tet_edges_fx6x2.data_ptr():
            t = tet_fx4.long()
            a = t[:, [0, 0, 0, 1, 1, 2]]
            b = t[:, [1, 2, 3, 2, 3, 3]]
            want = torch.stack([torch.minimum(a, b), torch.maximum(a, b)], -1)
            if not torch.equal(sorted_tet_edges_fx6x2.reshape(-1, 6, 2).long(), want):
                raise _lib.GShellHipError("sorted_tet_edges_fx6x2 is not the (min,max) edge table of tet_fx4 in the order 01 02 03 12 13 23")
            topo._edges_checked = sorted_tet_edges_fx6x2.data_ptr()
        pos_c = pos_nx3.detach().contiguous().float()
        sdf_c = sdf_n.detach().contiguous().float().reshape(-1)
        vdisc = verts_discretized.detach().round().to(torch.int32).contiguous()
        coeff = coeff_sdf_interp.detach().contiguous().float()
        mgrid = midpoint_msdf_sign_n.detach().contiguous().float()
        occ = occgrid.detach().contiguous().float()
        G, G2 = int(coeff.shape[0]), int(occ.shape[0])
        if coeff.dim() != 3 or tuple(coeff.shape) != (G, G, G) or tuple(mgrid.shape) != (G, G, G) or tuple(occ.shape) != (G2, G2, G2):
            raise _lib.GShellHipError(f"cubic grids expected, got {tuple(coeff.shape)}, {tuple(mgrid.shape)}, {tuple(occ.shape)}")
        if pos_c.shape[0] != topo.N or sdf_c.numel() != topo.N or tuple(vdisc.shape) != (topo.N, 3):
            raise _lib.GShellHipError("field sizes do not match the grid")
        if getattr(topo, "_vdisc_checked", None) != (verts_discretized.data_ptr(), G, G2):
            lo, hi = int(vdisc.min()), int(vdisc.max())
            if lo < 0 or hi >= G or 2 * hi >= G2:       # the reference would raise an index error here
                raise IndexError(f"verts_discretized range [{lo},{hi}] does not fit grids of {G}^3 / {G2}^3 cells")
            topo._vdisc_checked = (verts_discretized.data_ptr(), G, G2)
        counts = (c_int64 * 16)()
        with torch.cuda.device(dev):
            topo.sign_epoch += 1       # this pass rewrites the occupancy bits too
            check(L.gs_mtets_aug_count(topo.handle, ptr(sdf_c, torch.float32, "sdf"), ptr(vdisc, torch.int32, "verts_discretized"),
                                       ptr(mgrid), c_int64(G), stream(), counts), "gs_mtets_aug_count")
            V, M1, M2, T, V_aug = counts[0], counts[1], counts[2], counts[9], counts[10]
            f32 = dict(dtype=torch.float32, device=dev)
            verts_aug = torch.empty((V_aug, 3), **f32)
            msdf_aug = torch.empty((V_aug,), **f32)
            verts_wt = torch.empty((V, 3), **f32)
            faces_wt = torch.empty((M1 + 2 * M2, 3), dtype=torch.int64, device=dev)
            faces_aug = torch.empty((T, 3), dtype=torch.int64, device=dev)
            vert_ab = torch.empty((V, 2), dtype=torch.int32, device=dev)
            poly = torch.empty((3 * M1 + 4 * M2,), dtype=torch.int32, device=dev)
            cut_code = torch.empty((M1 + M2,), dtype=torch.uint8, device=dev)
            tet_id = torch.empty((M1 + M2,), dtype=torch.int32, device=dev)
            sign_code = torch.empty((M1 + M2,), dtype=torch.uint8, device=dev)
            grp_rank = torch.empty((M1 + M2,), dtype=torch.int32, device=dev)
            bnd_w = torch.empty((3 * M1 + 4 * M2, 2), **f32)
            check(L.gs_mtets_aug_fill(topo.handle, ptr(pos_c), ptr(sdf_c), ptr(vdisc), ptr(coeff), ptr(mgrid), c_int64(G), ptr(occ),
                                      c_int64(G2), ptr(verts_aug), ptr(msdf_aug), ptr(verts_wt), ptr(faces_wt), ptr(faces_aug),
                                      ptr(None), ptr(vert_ab), ptr(poly), ptr(cut_code), ptr(tet_id), ptr(sign_code), ptr(grp_rank),
                                      ptr(bnd_w), stream()), "gs_mtets_aug_fill")
            v_tng_aug = torch.zeros((V_aug, 3), **f32)
            if self.compute_tangents and V > 0:
                scratch = torch.empty((V, 7), **f32)
                check(L.gs_mtets_aug_tangents(c_int64(V), c_int64(M1), c_int64(M2), ptr(verts_wt), ptr(faces_wt), ptr(bnd_w),
                                              ptr(poly), ptr(topo.uv_lin), c_int64(topo.Nuv), ptr(scratch), ptr(v_tng_aug),
                                              stream()), "gs_mtets_aug_tangents")
        return verts_aug, faces_aug, None, None, v_tng_aug, verts_wt, tet_id.long(), msdf_aug, msdf_aug[:V]

"""GShellTetsGeometry with the reference's class surface (geometry/gshell_tets_geometry.py:45-384): owns the tet grid,
the SDF network / mSDF / deformation parameters and the shadow-ray context; `getMesh` -> `render` -> `tick`.
Parameter names (`sdf`, `msdf`, `deform`, `sdf_net.*`) match the reference's state_dict."""
import numpy as np
import torch
import torch.nn.functional as F

from .. import _lib
from .._lib import c_float, c_int64, check, ptr, stream
from ..render import mesh, optixutils as ou, regularizer, render
from .gshell_tets import GShell_Tets
from .mlp import MLP, check_forward_status, eikonal_sq_sum, forward_row_sharded, forward_row_sparse_backward


class _SdfRegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sdf, edges_i32):
        L = _lib.lib()
        s = sdf.detach().reshape(-1).contiguous().float()
        E = edges_i32.shape[0]
        nb = L.gs_sdf_reg_partials(c_int64(E))
        part = torch.empty((2, nb), dtype=torch.float32, device=s.device)
        with torch.cuda.device(s.device):
            check(L.gs_sdf_reg_fwd(ptr(s, torch.float32, "sdf"), ptr(edges_i32, torch.int32, "edges"), c_int64(E), ptr(part[0]), ptr(part[1]), stream()),
                  "gs_sdf_reg_fwd")
        tot = part.sum(dim=1)                               # [loss sum, crossing count], stays on the device
        ctx.save_for_backward(s, edges_i32, tot)
        ctx.shape = sdf.shape
        return tot[0] / tot[1].clamp_min(1.0)

    @staticmethod
    def backward(ctx, g):
        s, edges_i32, tot = ctx.saved_tensors
        g_sdf = torch.zeros_like(s)
        gs_ = g.detach().reshape(1).contiguous().float()
        cnt = tot[1:2].contiguous()
        with torch.cuda.device(s.device):
            check(_lib.lib().gs_sdf_reg_bwd(ptr(s), ptr(edges_i32), c_int64(edges_i32.shape[0]), ptr(gs_), ptr(cnt), ptr(g_sdf), stream()),
                  "gs_sdf_reg_bwd")
        return g_sdf.reshape(ctx.shape), None


def compute_sdf_reg_loss(sdf, all_edges):
    """Sign-consistency BCE over the grid edges whose end points disagree in sign (reference :33-39), as one fused
    HIP pass over the static edge list.  `all_edges`: [E,2] int32 (or int64, converted)."""
    if all_edges.dtype != torch.int32:
        all_edges = all_edges.int()
    return _SdfRegFn.apply(sdf, all_edges.contiguous())


class _MsdfRegFn(torch.autograd.Function):
    """(open_w * sum_i huber(clamp(msdf_i, min=-eps) + eps),  close_w * sum_j w_j huber(clamp(b_j, max=eps) - eps)) as one kernel
    each way (gs_msdf_reg_*): the mSDF regularisers of the reference's tick (:326-358)."""

    @staticmethod
    def forward(ctx, msdf, boundary, weight, eps, open_w, close_w):
        m = msdf.detach().reshape(-1).contiguous().float()
        b = boundary.detach().reshape(-1).contiguous().float()
        w = None if weight is None else weight.detach().reshape(-1).contiguous().float()
        out = torch.empty(2, dtype=torch.float32, device=m.device)
        with torch.cuda.device(m.device):
            check(_lib.lib().gs_msdf_reg_fwd(ptr(m, torch.float32, "msdf"), c_int64(m.numel()), ptr(b), ptr(w), c_int64(b.numel()), c_float(eps),
                                             c_float(open_w), c_float(close_w), ptr(out), stream()), "gs_msdf_reg_fwd")
        ctx.save_for_backward(m, b, w)
        ctx.k = (float(eps), float(open_w), float(close_w), msdf.shape, boundary.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        m, b, w = ctx.saved_tensors
        eps, open_w, close_w, sm, sb = ctx.k
        g_c = g.detach().contiguous().float()
        g_m = torch.empty_like(m) if ctx.needs_input_grad[0] else None
        g_b = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(m.device):
            check(_lib.lib().gs_msdf_reg_bwd(ptr(m), c_int64(m.numel()), ptr(b), ptr(w), c_int64(b.numel()), c_float(eps), c_float(open_w),
                                             c_float(close_w), ptr(g_c), ptr(g_m), ptr(g_b), stream()), "gs_msdf_reg_bwd")
        return (None if g_m is None else g_m.reshape(sm)), (None if g_b is None else g_b.reshape(sb)), None, None, None, None


class _LossTerms:
    """tick's loss terms by how they combine across view-sharded ranks; a group is summed when it is asked for"""

    def __init__(self, groups, total):
        self._groups, self._total = groups, total

    def __getitem__(self, k):
        return self._total(self._groups[k])

    def get(self, k, default=None):
        return self[k] if k in self._groups else default

    def __setitem__(self, k, v):
        self._groups[k] = [v]

    def add(self, k, t):
        self._groups[k] = list(self._groups[k]) + [t]


def boundary_weight(tri_i32, flags_u8, n_watertight, n_boundary):
    """visible_boundary_weight as ONE kernel (gs_boundary_weight: plain stores of 1.0, no atomics, no index arithmetic in ATen)."""
    w = torch.empty(n_boundary, dtype=torch.float32, device=tri_i32.device)
    with torch.cuda.device(tri_i32.device):
        check(_lib.lib().gs_boundary_weight(ptr(tri_i32, torch.int32, "tri"), ptr(flags_u8, torch.uint8, "flags"), c_int64(tri_i32.shape[0]),
                                            c_int64(n_watertight), c_int64(n_boundary), ptr(w), stream()), "gs_boundary_weight")
    return w


def visible_boundary_weight(tri, tri_flags, n_watertight, n_boundary):
    """[n_boundary] float 0/1: 1 where the boundary vertex (mesh vertex n_watertight + k) belongs to a flagged triangle.
    Equals the reference's `vis_mask` (gshell_tets_geometry.py:344-348: unique triangle ids -> their vertices -> boolean mask)
    as a scatter-max of the per-triangle flags, i.e. without data-dependent sizes."""
    idx = tri.reshape(-1) - n_watertight
    ok = idx >= 0
    src = (tri_flags.reshape(-1, 1) != 0).expand(-1, 3).reshape(-1) & ok
    # corners that are not boundary vertices are parked on 4096 spare slots: sending them all to ONE address serialises
    # 3.10^5 atomics there (8.5 ms on MI355X; scatter_reduce_('amax') likewise)
    spare = 4096
    park = n_boundary + (torch.arange(idx.numel(), device=tri.device) & (spare - 1))
    w = torch.zeros(n_boundary + spare, dtype=torch.float32, device=tri.device)
    w.scatter_add_(0, torch.where(ok, idx, park), src.to(torch.float32))
    return (w[:n_boundary] > 0).to(torch.float32)


def sample_points(v_pos, faces, n, generator=None):
    """Area-weighted uniform surface samples; stands in for kaolin.ops.mesh.sample_points (third-party, reference
    geometry/gshell_tets_geometry.py:236): face ~ area, (u,v) = (sqrt(r1), r2) -> (1-u, u(1-v), uv)."""
    v0, v1, v2 = v_pos[faces[:, 0]], v_pos[faces[:, 1]], v_pos[faces[:, 2]]
    area = torch.linalg.cross(v1 - v0, v2 - v0).norm(dim=-1)
    area = torch.where(torch.isfinite(area), area, torch.zeros_like(area)) + 1e-20
    fid = torch.multinomial(area, n, replacement=True, generator=generator)
    r = torch.rand(n, 2, device=v_pos.device, generator=generator)
    u, v = r[:, 0:1].sqrt(), r[:, 1:2]
    return (1 - u) * v0[fid] + u * (1 - v) * v1[fid] + u * v * v2[fid], fid


def sample_points_detached(v_pos, faces_i32, n, generator=None):
    """sample_points without a graph (the reference detaches the samples before use, :303): area kernel, one cumsum, one rand
    (whose generator is the reproducibility contract of view-sharded jobs) and ONE kernel that inverts the area CDF and forms the
    points -- torch.multinomial spent 0.12 ms renormalising its input row alone."""
    v = v_pos.detach().contiguous().float()
    T = faces_i32.shape[0]
    area = torch.empty(T, dtype=torch.float32, device=v.device)
    out = torch.empty((n, 3), dtype=torch.float32, device=v.device)
    fid = torch.empty(n, dtype=torch.int64, device=v.device)
    with torch.cuda.device(v.device):
        check(_lib.lib().gs_tri_area(ptr(v, torch.float32, "v_pos"), ptr(faces_i32, torch.int32, "faces"), c_int64(T), ptr(area), stream()), "gs_tri_area")
        cdf = torch.cumsum(area, dim=0)
        r = torch.rand(n, 3, device=v.device, generator=generator)       # (u, v, face) per sample
        check(_lib.lib().gs_surface_points_cdf(ptr(v), ptr(faces_i32), ptr(cdf), c_int64(T), ptr(r), c_int64(n), ptr(out), ptr(fid), stream()),
              "gs_surface_points_cdf")
    return out, fid


class GShellTetsGeometry(torch.nn.Module):
    def __init__(self, grid_res, scale, FLAGS, offset=None, tet_init_file=None, extract_from_generative=False, tet_grid=None):
        super().__init__()
        self.FLAGS, self.grid_res, self.scale = FLAGS, grid_res, scale
        self.gshell_tets = GShell_Tets(compute_tangents=False)       # tangents are dead on the training path (render.py:264-267)
        self.boxscale = torch.tensor(FLAGS.boxscale).view(1, 3).cuda()
        with torch.no_grad():
            self.optix_ctx = ou.OptiXContext()
            if tet_grid is not None:        # (vertices [N,3] float, indices [F,4] int) given directly (synthetic grids)
                verts, indices = tet_grid
            else:
                tets = np.load('data/tets/{}_tets.npz'.format(grid_res) if tet_init_file is None else tet_init_file)
                verts, indices = tets['vertices'], tets['indices']
            self.verts = torch.as_tensor(verts, dtype=torch.float32).cuda()
            self.verts = (self.verts - self.verts.mean(dim=0)) * scale * self.boxscale
            self.indices = torch.as_tensor(indices).long().cuda()
            self.generate_edges()
            if extract_from_generative:
                # cells of the 2x denser cubic grid that stores per-edge features (reference :70-78); the on-disk
                # 'tet_edges' table is re-derived from the indices (same (min,max) pairs, order 01 02 03 12 13 23)
                raw = torch.as_tensor(verts, dtype=torch.float32).cuda()
                self.original_verts = raw.clone()
                uniq = raw.reshape(-1).unique()
                dx = (uniq[1] - uniq[0]) / 2.0
                self.verts_discretized = ((raw - raw.min()) / dx).round()   # lattice coordinates; round() guards 3.9999 -> 3
                t = self.indices
                a, b = t[:, [0, 0, 0, 1, 1, 2]], t[:, [1, 2, 3, 2, 3, 3]]
                self.sorted_tetedges = torch.stack([torch.minimum(a, b), torch.maximum(a, b)], -1)
            else:
                self.original_verts = None
            self.offset = 0.0 if offset is None else torch.tensor(offset).cuda().view(1, 3)

        if self.FLAGS.use_sdf_mlp:
            self.sdf = torch.nn.Parameter(torch.zeros_like(self.verts[:, 0]), requires_grad=True)   # placeholder (reference :91)
            self.sdf_net = MLP(skip_in=FLAGS.skip_in, n_freq=FLAGS.n_freq, n_hidden=FLAGS.n_hidden, d_hidden=FLAGS.d_hidden,
                               use_float16=FLAGS.use_float16).cuda()
            opt = torch.optim.Adam(self.sdf_net.parameters(), lr=1e-3)
            target = (self.verts / self.boxscale).norm(dim=1, keepdim=True) - FLAGS.sphere_init_norm
            for _ in range(FLAGS.sdf_mlp_pretrain_steps):
                loss = (self.sdf_net(self.verts) - target).pow(2).mean()
                opt.zero_grad()
                loss.backward()
                opt.step()
        else:
            if not FLAGS.sphere_init:
                sdf = torch.rand_like(self.verts[:, 0]) - 0.1
            else:
                sdf = (self.verts / self.boxscale).norm(dim=1) - 0.5
            self.sdf = torch.nn.Parameter(sdf.clone().detach(), requires_grad=True)

        if getattr(FLAGS, "use_msdf_mlp", False):
            raise NotImplementedError("use_msdf_mlp is off in every reference config")
        msdf = (torch.rand_like(self.verts[:, 0]) - 0.01).clamp(-1, 1)
        self.msdf = torch.nn.Parameter(msdf.clone().detach(), requires_grad=True)
        self.deform = torch.nn.Parameter(torch.zeros_like(self.verts), requires_grad=True)
        self.clamp_deform()

    @torch.no_grad()
    def generate_edges(self):
        # sorted unique (min,max) grid edges (reference :141-156): the extractor's static topology already holds exactly this
        # list; it is built on the device at the first use (`all_edges` below), not in the constructor
        self._all_edges = None
        self.max_displacement = 1.0 / self.grid_res * self.scale / 2.1

    @property
    def all_edges(self):
        if self._all_edges is None:
            with torch.no_grad():
                self._all_edges = self.gshell_tets.topology(self.indices, self.verts.shape[0]).edges()   # [E,2] int32, sorted unique (min,max)
        return self._all_edges

    @torch.no_grad()
    def getAABB(self):
        return torch.min(self.verts, dim=0).values, torch.max(self.verts, dim=0).values

    @torch.no_grad()
    def clamp_deform(self):
        if not self.FLAGS.use_tanh_deform:
            self.deform.data[:] = self.deform.clamp(-1.0, 1.0)
        self.msdf.data[:] = self.msdf.clamp(-2.0, 2.0)

    def getMesh_from_augmented_grid_withocc(self, material, sdf_sign, sdf_coeff, msdf_sign, occgrid):
        """Mesh of one generated cubic grid (reference :167-189; caller eval_gmeshdiffusion_generated_samples.py:180)."""
        v_deformed = self.verts + self.max_displacement * self.deform
        sdf = self.sdf_net(v_deformed) if self.FLAGS.use_sdf_mlp else self.sdf
        verts, faces, uvs, uv_idx, v_tng, _v_wt, _tet_gidx, v_msdf, _m_wt = self.gshell_tets_full.marching_from_auggrid(
            v_deformed, sdf_sign, self.indices, self.sorted_tetedges, sdf_coeff, self.verts_discretized, msdf_sign, occgrid)
        imesh = mesh.Mesh(verts, faces, v_tex=uvs, t_tex_idx=uv_idx, material=material)
        imesh = mesh.auto_normals(imesh)
        imesh = mesh.compute_tangents(imesh, v_tng=v_tng)
        return {'imesh': imesh, 'sdf': sdf, 'v_msdf': v_msdf}

    @property
    def gshell_tets_full(self):
        """Extractor with tangents enabled (the decode path returns them); shares the static topology."""
        ext = getattr(self, "_gshell_tets_full", None)
        if ext is None:
            ext = GShell_Tets(compute_tangents=True)
            ext._topo_cache = self.gshell_tets._topo_cache
            self._gshell_tets_full = ext
        return ext

    def _sdf_values(self, v_deformed):
        """SDF of every grid vertex; backward only through the rows that receive gradient (geometry/mlp.py).  In a
        view-sharded job the rows are split over the ranks and the values all-gathered (FLAGS.shard_mlp_rows)."""
        if not self.FLAGS.use_sdf_mlp:
            return self.sdf
        shard = getattr(self.FLAGS, "view_shard", None)
        if shard is not None and shard.world > 1 and getattr(self.FLAGS, "shard_mlp_rows", False):
            return forward_row_sharded(self.sdf_net, v_deformed, shard)
        # single GPU: the kernel's epilogue also writes the extraction's occupancy bits (fused geometry front end)
        sink = self.gshell_tets.topology(self.indices, self.verts.shape[0]) if hasattr(self, 'gshell_tets') else self._refine_edges()
        return forward_row_sparse_backward(self.sdf_net, v_deformed, sign_sink=sink)

    def _refine_edges(self):
        """Edge set of the two-pass SDF forward for a geometry without a TetTopology (overridden by G-FlexiCubes); None = one pass."""
        return None

    def _recover_forward(self, todo):
        """What check_forward_status asked for after an extraction's host sync: switch this network's evaluation mode (reference fp32
        GEMMs, geometry/mlp.py:32-40, have neither the fp16 range limit nor a one-product pass); the caller then re-evaluates."""
        import warnings
        if todo == "fp32":
            warnings.warn("SDF network: fp16-pair arithmetic overflowed; switching this network to torch fp32 ops")
            self.sdf_net._gs_precision = "torch"
        else:
            warnings.warn(f"SDF network: the one-product pass used up {self.sdf_net.__dict__.get('_gs_two_pass_margin_used'):.3f} of the sign margin of a "
                          f"re-evaluated row (max deviation {self.sdf_net.__dict__.get('_gs_two_pass_maxdev'):.3e}; refined rows + audit sample of all "
                          "rows), more than 1 / 4: this network is evaluated in one pass from now on")
            self.sdf_net._gs_one_pass = True

    def getMesh(self, material):
        v_deformed = self.verts + self.max_displacement * self.deform
        # SDF of every grid vertex; backward only through the rows that receive gradient (see geometry/mlp.py)
        v_grid = v_deformed                       # the network's input (before the offset): a re-evaluation must see the identical tensor
        sdf = self._sdf_values(v_grid)
        msdf = self.msdf
        v_deformed = v_deformed + self.offset
        verts, faces, uvs, uv_idx, v_tng, extra = self.gshell_tets(v_deformed, sdf, msdf, self.indices)
        if self.FLAGS.use_sdf_mlp:
            # status words of the fused forward pass; the extraction's count has just synchronised the stream, so this costs no stall
            todo = check_forward_status(self.sdf_net)
            if todo is not None:
                # an activation or weight beyond the fp16 range, or a one-product pass that is off by too much: from now on this network is
                # evaluated by plain torch fp32 ops (counted in mlp.FALLBACKS) / in one pass
                self._recover_forward(todo)
                sdf = self._sdf_values(v_grid)
                verts, faces, uvs, uv_idx, v_tng, extra = self.gshell_tets(v_deformed, sdf, msdf, self.indices)
                check_forward_status(self.sdf_net)
        if self.FLAGS.use_sdf_mlp and getattr(self.FLAGS, "sync_free_rows", False):
            # d loss / d sdf is non-zero only at end points of sign-crossing edges (extraction backward + sign regulariser): a bound
            # that lets the row-sparse backward size its planes without waiting for the exact count (geometry/mlp.py).  Measured on
            # MI355X (r02): 31.0 ms / iteration with it, 30.8 without (3x larger planes, three more launches) -- off by default.
            self.sdf_net._gs_rows_bound = 2 * int(extra['n_verts_watertight']) + 128
        imesh = mesh.Mesh(verts, faces, v_tex=uvs, t_tex_idx=uv_idx, material=material)
        imesh.t_pos_idx_i32 = extra['faces_i32']
        with torch.no_grad():
            ou.optix_build_bvh(self.optix_ctx, imesh.v_pos.contiguous(), imesh.faces_i32(), rebuild=1)
        imesh = mesh.auto_normals(imesh)
        self.last_mesh_sizes = (int(imesh.v_pos.shape[0]), int(imesh.t_pos_idx.shape[0]))
        out = {'imesh': imesh, 'sdf': sdf, 'msdf': extra['msdf'], 'msdf_watertight': extra['msdf_watertight'],
               'msdf_boundary': extra['msdf_boundary'], 'n_verts_watertight': extra['n_verts_watertight']}
        if getattr(self.FLAGS, "visualize_watertight", False):
            wt = mesh.Mesh(extra['vertices_watertight'], extra['faces_watertight'], material=material)
            out['imesh_watertight'] = mesh.auto_normals(wt)
        return out

    def _launch_eikonal(self, pts):
        """sum_i (|grad f(p_i)| - 1)^2 over this rank's share of the surface samples (reference :302-324), launched as soon as the
        samples exist.  It does not depend on the rendering, so FLAGS.eikonal_side_stream can put it on a SIDE STREAM (autograd
        replays the backward there too): the chain kernels (matrix pipe, HBM planes) then share the chip with the render pass's
        VALU-bound stages.  Measured on MI355X: round 2, 35.0 ms / iteration with the side stream against 34.7 without; round 6, on the
        14.7 ms iteration, 14.49 (mean of 5 runs) with it against 14.75 without.  OFF by default all the same: while k_h2_fwd / k_h2_bwd
        (the chain) run on another queue, packed-fp32 arithmetic of the rasteriser's k_rast_small returned wrong products now and then
        (single samples with a wrong depth in 13 - 71 % of the frames of tools/raster_race_probe6.py: 6 of 25 chain-test runs failed on
        their visible-triangle sets; none of 40 without the side stream).  The rasteriser is immune since (raster.hip is compiled without
        packed-fp32 instructions: 0 of 1000 frames, 0 of 25 runs), but what makes a co-resident kernel a victim is not understood, and
        the render pass's other kernels were not audited -- so nothing of the product runs two kernels at once.  Record:
        profiles/r06_two_queue_probes.txt, DESIGN.md 5.4.
        -> (sum, number of samples of the GLOBAL batch, stream or None)"""
        FL = self.FLAGS
        shard = getattr(FL, "view_shard", None)
        n_total = pts.shape[0]
        if shard is not None and shard.world > 1:      # identical sample set on every rank (seeded): rank r takes samples r, r + world, ...
            pts = pts[shard.rank::shard.world].contiguous()
        if getattr(FL, "eikonal_side_stream", False) and pts.is_cuda:
            main = torch.cuda.current_stream()
            side = getattr(self, "_side_stream", None)
            if side is None:
                side = self._side_stream = torch.cuda.Stream(device=pts.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                s = eikonal_sq_sum(self.sdf_net, pts)
            pts.record_stream(side)
            return s, n_total, side
        return eikonal_sq_sum(self.sdf_net, pts), n_total, None

    def render(self, glctx, target, lgt, opt_material, bsdf=None, denoiser=None, shadow_scale=1.0, use_uv=False, _with_eikonal=False):
        d = self.getMesh(opt_material)
        opt_mesh = d['imesh']
        if opt_mesh.v_pos.size(0) != 0 and opt_mesh.t_pos_idx.size(0) != 0:
            ns = getattr(self.FLAGS, 'noise_stream', None)       # seeded by (iteration) on every rank of a view-sharded job
            gen = ns.generator('eikonal', opt_mesh.v_pos.device) if ns is not None else None
            if opt_mesh.v_pos.is_cuda and hasattr(opt_mesh, 'faces_i32'):
                d['sampled_pts'] = sample_points_detached(opt_mesh.v_pos, opt_mesh.faces_i32().contiguous(), 50000, generator=gen)[0]
            else:
                d['sampled_pts'] = sample_points(opt_mesh.v_pos, opt_mesh.t_pos_idx, 50000, generator=gen)[0]
        else:
            d['sampled_pts'] = None
        if _with_eikonal and self.FLAGS.use_sdf_mlp and self.FLAGS.use_eikonal and d['sampled_pts'] is not None:
            d['eikonal'] = self._launch_eikonal(d['sampled_pts'].detach())
        d['buffers'] = render.render_mesh(self.FLAGS, glctx, opt_mesh, target['mvp'], target['campos'], lgt, target['resolution'], spp=target['spp'],
                                          msaa=True, background=target['background'], bsdf=bsdf, use_uv=use_uv, optix_ctx=self.optix_ctx,
                                          denoiser=denoiser, shadow_scale=shadow_scale, extra_dict={'msdf': d['msdf']})
        if getattr(self.FLAGS, "visualize_watertight", False):
            d['buffers_watertight'] = render.render_mesh(self.FLAGS, glctx, d['imesh_watertight'], target['mvp'], target['campos'], lgt,
                                                         target['resolution'], spp=target['spp'], msaa=True, background=target['background'],
                                                         bsdf=bsdf, use_uv=use_uv, optix_ctx=self.optix_ctx, denoiser=denoiser,
                                                         shadow_scale=shadow_scale, extra_dict={'msdf': d['msdf']})
        return d

    def _frame_sum_weights(self, dev, n_px, with_msdf, with_light, with_img=False):
        """constant weight vectors over regularizer.frame_sums' nine sums: (image terms, regulariser terms); cached on the device"""
        FL = self.FLAGS
        key = (str(dev), n_px, with_msdf, with_light, with_img, FL.lambda_diffuse, FL.lambda_kd, FL.lambda_ks, FL.lambda_nrm)
        cache = self.__dict__.setdefault('_fs_weights', {})
        if key not in cache:
            w_img = [1.0 / n_px, 0.5 / n_px if with_msdf else 0.0, 0.5 / n_px if with_msdf else 0.0, 0, 0, 0, 0, 0, 0]
            w_reg = [0, 0, 0, FL.lambda_diffuse / n_px if with_light else 0.0, 0, 0, FL.lambda_kd / n_px, FL.lambda_ks / (3 * n_px),
                     FL.lambda_nrm / (3 * n_px)]
            if with_img:       # image_loss is the mean over B*H*W*3 elements
                w_img.append(1.0 / (3 * n_px))
                w_reg.append(0.0)
            cache[key] = (torch.tensor(w_img, dtype=torch.float32, device=dev), torch.tensor(w_reg, dtype=torch.float32, device=dev))
        return cache[key]

    def tick(self, glctx, target, lgt, opt_material, loss_fn, iteration, denoiser):
        FL = self.FLAGS
        t_iter = iteration / FL.iter
        shadow_ramp = min(iteration / 1000, 1.0)
        if denoiser is not None:
            denoiser.set_influence(shadow_ramp)
        d = self.render(glctx, target, lgt, opt_material, denoiser=denoiser, shadow_scale=shadow_ramp, _with_eikonal=True)
        buffers = d['buffers']
        dev = buffers['shaded'].device

        # ---- image losses (reference :275-285)
        color_ref = target['img']
        gt_mask = color_ref[..., 3:]
        # pixel sums of the alpha MSE, the mSDF image terms and the image-space regularisers in one fused pass
        stacked = getattr(buffers, 'stacked', None)
        # a loss object of this package names its (loss, tonemapper): the colour term then rides the same pass (fs[9])
        img_spec = getattr(loss_fn, 'gs_spec', None) if getattr(FL, "fused_image_loss", True) else None
        if stacked is None or 'shaded' not in stacked[1]:
            img_spec = None
        fs = regularizer.frame_sums(stacked, color_ref, img_spec) if (stacked is not None and getattr(FL, "fused_frame_sums", True)) else None
        if fs is None:
            img_spec = None
        n_px = float(gt_mask.numel())
        shard = getattr(FL, "view_shard", None)
        world = shard.world if shard is not None else 1
        use_fs_msdf = fs is not None and 'msdf_image' in buffers
        has_spec = fs is not None and 'diffuse_light' in buffers and 'specular_light' in buffers
        if fs is not None:
            # every term that is LINEAR in the nine frame sums comes out of two dot products with constant weight vectors (the scalar
            # algebra of the reference's tick is ~140 five-microsecond launches per iteration when written operator by operator)
            w_img, w_reg = self._frame_sum_weights(dev, n_px, use_fs_msdf, 'diffuse_light' in buffers, img_spec is not None)
            img_loss = torch.dot(fs, w_img)
        else:
            img_loss = F.mse_loss(buffers['shaded'][..., 3:], gt_mask)
        if img_spec is None:
            img_loss = img_loss + loss_fn(buffers['shaded'][..., 0:3] * gt_mask, color_ref[..., 0:3] * gt_mask)
        if not use_fs_msdf:
            msdf_img = buffers['msdf_image']
            img_loss = img_loss + 5e-1 * F.l1_loss(msdf_img.clamp(min=0) * (gt_mask == 0).float(), torch.zeros_like(gt_mask))
            img_loss = img_loss + 5e-1 * F.l1_loss(msdf_img.clamp(max=0) * (gt_mask == 1).float(), torch.ones_like(gt_mask))
        depth_loss = torch.zeros((), device=dev)         # use_depth is off in every reference config

        glob = []            # terms that do not depend on the local views (identical on every rank)
        presharded = []      # terms whose SUM over ranks is the single-GPU term (weight 1 in the sharded loss)
        per_view = []        # per-view means besides img_loss

        # ---- eikonal on the SDF network at surface samples (reference :302-324)
        if d.get('eikonal') is not None:
            eik_sum, n_total, side = d['eikonal']
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
            if FL.eikonal_scale is None:
                eik_coeff = 3e-1 if iteration < 500 else (1e-1 if iteration < 2000 else 1e-2)
            else:
                eik_coeff = FL.eikonal_scale
            (presharded if world > 1 else glob).append(eik_sum * (eik_coeff / n_total))

        # ---- mSDF open / close regularisers (reference :326-358), one kernel each way
        if FL.use_mesh_msdf_reg and (FL.msdf_reg_open_scale > 0 or FL.msdf_reg_close_scale != 0):
            regscale = (64 / self.grid_res) ** 3
            vis_w = None
            if FL.msdf_reg_close_scale != 0:
                # boundary vertices of the triangles seen by ANY view (reference :344-348), without the reference's two
                # data-dependent compactions (unique ids -> index list -> boolean mask): the rasteriser's per-triangle flags
                # are scattered onto the boundary vertices and the Huber terms are summed under that 0/1 weight -- no host sync
                with torch.no_grad():
                    nwt = d['n_verts_watertight']
                    imesh = d['imesh']
                    tri_i32 = imesh.faces_i32() if hasattr(imesh, 'faces_i32') else imesh.t_pos_idx.int()
                    flags = getattr(buffers, 'visible_flags', None)
                    if flags is None:
                        flags = torch.zeros(tri_i32.size(0), dtype=torch.uint8, device=dev)
                        flags[buffers['visible_triangles']] = 1
                    if world > 1:     # union over the views of the global batch
                        fl32 = flags.to(torch.int32)
                        shard.all_reduce_max(fl32)
                        flags = fl32.to(torch.uint8)
                    vis_w = boundary_weight(tri_i32.contiguous(), flags.to(torch.uint8).contiguous(), int(nwt), d['msdf_boundary'].size(0))
            two = _MsdfRegFn.apply(d['msdf'], d['msdf_boundary'], vis_w, 1e-3, max(FL.msdf_reg_open_scale, 0.0) * regscale,
                                   FL.msdf_reg_close_scale * regscale)
            glob.append(two.sum())

        sdf_weight = FL.sdf_regularizer - (FL.sdf_regularizer - 0.01) * min(1.0, 4.0 * t_iter)
        glob.append(compute_sdf_reg_loss(d['sdf'], self.all_edges) * sdf_weight)

        if fs is not None:
            per_view.append(torch.dot(fs, w_reg))        # diffuse monochrome term + material smoothness
            if has_spec and world > 1:
                # the specular / diffuse energy ratio is a ratio of GLOBAL-batch means (regularizer.py:43-51): all-reduce the two
                # luma sums (no_grad) and back-propagate the linearisation  d(S4/S5) = ds4_r / S5 - S4 / S5^2 ds5_r  per rank;
                # the values sum to the global ratio over the ranks, the gradients sum to its gradient
                with torch.no_grad():
                    S = torch.stack((fs[4], fs[5]))
                    shard.all_reduce_sum(S)
                n_glob = n_px * world
                ratio = fs[4] / S[1] - (S[0] / (S[1] * S[1])) * (fs[5] - fs[5].detach())
                clamped = fs[4] / (1e-3 * n_glob)
                presharded.append(torch.where(S[1] / n_glob >= 1e-3, ratio, clamped) * FL.lambda_specular)
            elif has_spec:
                per_view.append(fs[4] / fs[5].clamp_min(1e-3 * n_px) * FL.lambda_specular)      # = mean / clamp(mean, 1e-3)
        else:
            if 'diffuse_light' in buffers:
                per_view.append(regularizer.shading_loss(buffers['diffuse_light'], buffers['specular_light'], color_ref, FL.lambda_diffuse,
                                                         FL.lambda_specular))
            per_view.append(regularizer.material_smoothness_grad(buffers['kd_grad'], buffers['ks_grad'], buffers['normal_grad'],
                                                                 lambda_kd=FL.lambda_kd, lambda_ks=FL.lambda_ks, lambda_nrm=FL.lambda_nrm))
        if FL.lambda_chroma != 0:
            per_view.append(regularizer.chroma_loss(buffers['kd'], color_ref, FL.lambda_chroma))

        def total(ts):
            acc = None
            for t in ts:
                acc = t if acc is None else acc + t
            return acc if acc is not None else torch.zeros((), device=dev)

        reg_loss = total(glob + per_view + presharded)
        # decomposition for view-sharded training: per-view means, terms that do not depend on the local views (identical on
        # every rank), and terms already split over the ranks (eikonal samples, linearised energy ratio); summed on demand
        self.last_terms = _LossTerms({'per_view': [img_loss] + per_view, 'global': glob, 'presharded': presharded}, total)
        return img_loss, depth_loss, reg_loss

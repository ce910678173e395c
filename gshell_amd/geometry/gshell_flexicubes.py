"""G-FlexiCubes extractor -- drop-in for the reference's geometry/gshell_flexicubes.py (class GShellFlexiCubes,
`construct_voxel_grid` :103-134 and `__call__` :136-230; the non-training, grad_func=None path every G-Shell script runs).

    verts, faces, L_dev, extra = GShellFlexiCubes()(x_nx3, s_n, nu_n, cube_fx8, res, beta_fx12, alpha_fx8, gamma_f)

Index / topology work runs as HIP kernels (gshell_amd/csrc/flexi.hip) over a STATIC per-grid edge table, so no
`torch.unique` / stable sort / python loop over `num_vd` groups runs per call; the prefix ranks behind the reference's
orderings are three scan launches, and the floating-point part (weighted zero crossings, dual vertices, nu_d with the
reference's in-place quirk, L_dev, the mSDF cut interpolation) and its adjoint are one kernel each
(gshell_amd/csrc/flexi_float.hip, wrapped by _FlexiVdFn / _FlexiCutFn).  Only the tanh / sigmoid normalisation of the
per-cube weights stays in torch.  Output orderings are bit-identical to the reference (tests/test_flexi_gpu.py against
goldens minted from the real reference).

Not implemented (never reached by the reference's scripts): training=True quad fans, output_tetmesh, grad_func.
"""
import torch

from .. import _lib
from .._lib import c_int64, check, ptr, stream

# local cube edge e joins corners _CUBE_EDGES[e] in this orientation (reference :88-89)
_CUBE_EDGES = [[0, 1], [1, 5], [4, 5], [0, 4], [2, 3], [3, 7], [6, 7], [2, 6], [2, 0], [3, 1], [7, 5], [6, 4]]
# mSDF cut of a triangle: ids 0-2 corners, 3-5 boundary points on edges (01, 12, 20); index = m0*4 + m1*2 + m2 (reference
# geometry/flexicubes_table.py:794-812)
_CUT_N = [0, 1, 1, 2, 1, 2, 2, 1]
_CUT_CFG = [[0, 0, 0, 0, 0, 0], [4, 2, 5, 0, 0, 0], [3, 1, 4, 0, 0, 0], [3, 1, 2, 3, 2, 5], [0, 3, 5, 0, 0, 0], [0, 3, 4, 0, 4, 2],
            [0, 1, 4, 0, 4, 5], [0, 1, 2, 0, 0, 0]]


class FlexiTopology:
    """Static per-grid tables in HBM (built once with torch sort/unique; the per-call path never sorts)."""

    def __init__(self, cube_fx8, num_verts, res):
        dev = cube_fx8.device
        cubes = cube_fx8.long()
        self.F, self.N = int(cubes.shape[0]), int(num_verts)
        self.res = [int(res)] * 3 if not isinstance(res, (list, tuple)) else [int(r) for r in res]
        ce = torch.tensor(_CUBE_EDGES, device=dev)
        pairs = cubes[:, ce]
        key = pairs[..., 0] * self.N + pairs[..., 1]
        ukey, inv, counts = torch.unique(key.reshape(-1), return_inverse=True, return_counts=True)
        self.E = int(ukey.numel())
        self.edges = torch.stack([ukey // self.N, ukey % self.N], -1).int().contiguous()
        self.cube_edge = inv.reshape(-1, 12).int().contiguous()
        self.ncubes = counts.to(torch.uint8).contiguous()
        order = torch.sort(inv, stable=True).indices
        start = torch.cumsum(counts, 0) - counts
        inc = torch.full((self.E, 4), -1, dtype=torch.int32, device=dev)
        grp = inv[order]
        inc[grp, torch.arange(order.numel(), device=dev) - start[grp]] = order.int()
        self.inc = inc.contiguous()
        self.cubes_i32 = cubes.int().contiguous()
        self.corner_of_edge = ce                                   # [12,2]


class _FlexiVdFn(torch.autograd.Function):
    """Dual vertices / nu_d / nu_d_stopvgd / L_dev from the entry tables (gs_flexi_vd_fwd / bwd)."""

    @staticmethod
    def forward(ctx, x, s, nu, beta, alpha, topo, ent_edge, ent_cube, ent_e, vd_start, n_vd, n_entries):
        L = _lib.lib()
        dev = x.device
        t = [v.detach().contiguous().float() for v in (x, s, nu, beta, alpha)]
        vd = torch.empty((n_vd, 3), dtype=torch.float32, device=dev)
        nu_d, nu_d_sv = torch.empty(n_vd, dtype=torch.float32, device=dev), torch.empty(n_vd, dtype=torch.float32, device=dev)
        l_dev = torch.empty(n_entries, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(L.gs_flexi_vd_fwd(*[ptr(v, torch.float32) for v in t], ptr(topo.edges, torch.int32), ptr(ent_edge), ptr(ent_cube), ptr(ent_e), ptr(vd_start),
                                    c_int64(n_vd), ptr(vd), ptr(nu_d), ptr(nu_d_sv), ptr(l_dev), stream()), "gs_flexi_vd_fwd")
        ctx.t, ctx.tables, ctx.n_vd = t, (topo.edges, ent_edge, ent_cube, ent_e, vd_start), n_vd
        ctx.shapes = (x.shape, s.shape, nu.shape, beta.shape, alpha.shape)
        return vd, nu_d, nu_d_sv, l_dev

    @staticmethod
    def backward(ctx, g_vd, g_nu_d, g_nu_d_sv, g_l):
        L = _lib.lib()
        t, (edges, ent_edge, ent_cube, ent_e, vd_start) = ctx.t, ctx.tables
        dev = t[0].device
        z = lambda ref: torch.zeros_like(ref)
        g_x, g_s, g_nu, g_beta, g_alpha = z(t[0]), z(t[1]), z(t[2]), z(t[3]), z(t[4])
        f = lambda g, n: (torch.zeros(n, dtype=torch.float32, device=dev) if g is None else g.contiguous().float())
        gv, gn, gs_, gl = f(g_vd, (ctx.n_vd, 3)), f(g_nu_d, ctx.n_vd), f(g_nu_d_sv, ctx.n_vd), f(g_l, ent_edge.shape[0])
        with torch.cuda.device(dev):
            check(L.gs_flexi_vd_bwd(*[ptr(v) for v in t], ptr(edges), ptr(ent_edge), ptr(ent_cube), ptr(ent_e), ptr(vd_start), c_int64(ctx.n_vd), ptr(gv),
                                    ptr(gn), ptr(gs_), ptr(gl), ptr(g_x), ptr(g_s), ptr(g_nu), ptr(g_beta), ptr(g_alpha), stream()), "gs_flexi_vd_bwd")
        sh = ctx.shapes
        return (g_x.reshape(sh[0]), g_s.reshape(sh[1]), g_nu.reshape(sh[2]), g_beta.reshape(sh[3]), g_alpha.reshape(sh[4])) + (None,) * 7


class _FlexiCutFn(torch.autograd.Function):
    """Boundary vertices of the mSDF cut: bverts = interp_nonan(nu_d, vd), bnu = interp_nonan(nu_d_sv detached, nu_d_sv)."""

    @staticmethod
    def forward(ctx, vd, nu_d, nu_d_sv, pa, pb):
        L = _lib.lib()
        t = [v.detach().contiguous().float() for v in (vd, nu_d, nu_d_sv)]
        n = pa.shape[0]
        bverts = torch.empty((n, 3), dtype=torch.float32, device=vd.device)
        bnu = torch.empty(n, dtype=torch.float32, device=vd.device)
        with torch.cuda.device(vd.device):
            check(L.gs_flexi_cut_fwd(ptr(pa, torch.int64), ptr(pb, torch.int64), c_int64(n), ptr(t[0]), ptr(t[1]), ptr(t[2]), ptr(bverts), ptr(bnu), stream()),
                  "gs_flexi_cut_fwd")
        ctx.t, ctx.idx = t, (pa, pb)
        return bverts, bnu

    @staticmethod
    def backward(ctx, g_bverts, g_bnu):
        L = _lib.lib()
        t, (pa, pb) = ctx.t, ctx.idx
        n = pa.shape[0]
        dev = t[0].device
        g_vd, g_nu_d, g_nu_sv = torch.zeros_like(t[0]), torch.zeros_like(t[1]), torch.zeros_like(t[2])
        gb = torch.zeros((n, 3), dtype=torch.float32, device=dev) if g_bverts is None else g_bverts.contiguous().float()
        gn = torch.zeros(n, dtype=torch.float32, device=dev) if g_bnu is None else g_bnu.contiguous().float()
        with torch.cuda.device(dev):
            check(L.gs_flexi_cut_bwd(ptr(pa), ptr(pb), c_int64(n), ptr(t[0]), ptr(t[1]), ptr(t[2]), ptr(gb), ptr(gn), ptr(g_vd), ptr(g_nu_d), ptr(g_nu_sv),
                                     stream()), "gs_flexi_cut_bwd")
        return g_vd, g_nu_d, g_nu_sv, None, None


class GShellFlexiCubes:
    def __init__(self, device="cuda", qef_reg_scale=1e-3, weight_scale=0.99):
        self.device = device
        self.weight_scale = weight_scale
        self.qef_reg_scale = qef_reg_scale
        self.cube_edges = torch.tensor(_CUBE_EDGES, dtype=torch.long, device=device).reshape(-1)
        self._topo_cache = {}

    # ---- grid ------------------------------------------------------------------------------------------
    def construct_voxel_grid(self, res):
        """Vertices in [-0.5, 0.5]^3 and the 8 corner indices of every cube, in the reference's layout (:103-134): vertex
        index = lexicographic rank of (x,y,z); cube (i,j,k) has number (i*res_y + j)*res_z + k; corner c = (c&1, (c>>1)&1, c>>2)."""
        r = [res] * 3 if isinstance(res, int) else list(res)
        dev = self.device
        ax = [torch.arange(n + 1, device=dev) for n in r]
        i, j, k = torch.meshgrid(*ax, indexing="ij")
        verts = torch.stack([i / r[0], j / r[1], k / r[2]], -1).reshape(-1, 3).float() - 0.5
        ci, cj, ck = torch.meshgrid(*[torch.arange(n, device=dev) for n in r], indexing="ij")
        base = torch.stack([ci, cj, ck], -1).reshape(-1, 1, 3)
        corner = torch.tensor([[c & 1, (c >> 1) & 1, c >> 2] for c in range(8)], device=dev)[None]
        p = base + corner
        cubes = (p[..., 0] * (r[1] + 1) + p[..., 1]) * (r[2] + 1) + p[..., 2]
        return verts, cubes

    def topology(self, cube_fx8, num_verts, res):
        key = (cube_fx8.data_ptr(), tuple(cube_fx8.shape), int(num_verts), str(res))
        t = self._topo_cache.get(key)
        if t is None:
            t = FlexiTopology(cube_fx8, num_verts, res)
            self._topo_cache = {key: t}
        return t

    # ---- extraction ----------------------------------------------------------------------------------------
    def __call__(self, x_nx3, s_n, nu_n, cube_fx8, res, beta_fx12=None, alpha_fx8=None, gamma_f=None, training=False, output_tetmesh=False,
                 grad_func=None):
        if output_tetmesh or grad_func is not None:
            raise NotImplementedError("output_tetmesh (the reference itself raises, gshell_flexicubes.py:224) / grad_func (the non-differentiable QEF mode) are never "
                                      "used by the G-Shell scripts (SURVEY.md 3.4)")
        L = _lib.lib()
        dev = x_nx3.device
        topo = self.topology(cube_fx8, x_nx3.shape[0], res)
        F, E, N = topo.F, topo.E, topo.N
        s1 = s_n.reshape(-1)
        nu1 = nu_n.reshape(-1)
        s_c = s1.detach().contiguous().float()
        i32 = dict(dtype=torch.int32, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)

        with torch.cuda.device(dev), torch.no_grad():
            case_id, num_vd, n_ent = torch.empty(F, **u8), torch.empty(F, **u8), torch.empty(F, **u8)
            scratch = torch.empty(2 * F, **u8)
            check(L.gs_flexi_classify(ptr(s_c, torch.float32, "s"), ptr(topo.cubes_i32), c_int64(F), c_int64(topo.res[0]), c_int64(topo.res[1]),
                                      c_int64(topo.res[2]), ptr(scratch), ptr(case_id), ptr(num_vd), ptr(n_ent), stream()), "gs_flexi_classify")
            flags = torch.empty(E, **u8)
            check(L.gs_flexi_edge_flags(ptr(s_c), ptr(topo.edges), ptr(topo.ncubes), c_int64(E), ptr(flags), stream()), "gs_flexi_edge_flags")
            # prefix ranks in the reference's orderings (device scans: count / scan / apply) + the one host sync for the sizes
            vd_base32, ent_base32, qrank = torch.empty(F, **i32), torch.empty(F, **i32), torch.empty(E, **i32)
            rscratch = torch.empty(max(int(L.gs_flexi_ranks_scratch_bytes(c_int64(F), c_int64(E))), 8) // 4 + 2, **i32)
            totals = torch.empty(16, dtype=torch.int64, device=dev)
            check(L.gs_flexi_ranks(ptr(num_vd), ptr(n_ent), ptr(flags), c_int64(F), c_int64(E), ptr(rscratch), ptr(totals), ptr(vd_base32), ptr(ent_base32),
                                   ptr(qrank), stream()), "gs_flexi_ranks")
            n_vd, n_entries, n_quads = (int(v) for v in totals[10:13].tolist())
        if n_vd == 0:           # no surface: the reference returns a 3-tuple (:193-202)
            return (torch.zeros((0, 3), device=dev), torch.zeros((0, 3), dtype=torch.long, device=dev), torch.zeros((0,), device=dev))

        with torch.cuda.device(dev), torch.no_grad():
            ent_vd, ent_edge, ent_cube, ent_e = (torch.empty(n_entries, **i32) for _ in range(4))
            vd_idx_map = torch.full((F, 12), -1, **i32)
            vd_cube = torch.empty(n_vd, **i32)
            vd_start = torch.empty(n_vd + 1, **i32)
            check(L.gs_flexi_entries(ptr(case_id), ptr(num_vd), ptr(vd_base32), ptr(ent_base32), ptr(topo.cube_edge),
                                     c_int64(F), ptr(ent_vd), ptr(ent_edge), ptr(ent_cube), ptr(ent_e), ptr(vd_idx_map), ptr(vd_cube), ptr(vd_start),
                                     c_int64(n_vd), c_int64(n_entries), stream()), "gs_flexi_entries")

        # ---- normalised weights (:242-264)
        ws = self.weight_scale
        beta = torch.ones((F, 12), device=dev) if beta_fx12 is None else torch.tanh(beta_fx12) * ws + 1
        alpha = torch.ones((F, 8), device=dev) if alpha_fx8 is None else torch.tanh(alpha_fx8) * ws + 1
        gamma = torch.ones((F,), device=dev) if gamma_f is None else torch.sigmoid(gamma_f) * ws + (1 - ws) / 2

        # ---- dual vertices, nu_d, nu_d_stopvgd, L_dev (:387-485, :232-240): one kernel, one thread per dual vertex
        vd, nu_d, nu_d_sv, L_dev = _FlexiVdFn.apply(x_nx3, s1, nu1, beta, alpha, topo, ent_edge, ent_cube, ent_e, vd_start, n_vd, n_entries)
        nu_d, nu_d_sv = nu_d[:, None], nu_d_sv[:, None]

        # ---- quads -> triangles (:487-522)
        with torch.cuda.device(dev), torch.no_grad():
            vd_gamma = gamma.detach()[vd_cube.long()].contiguous().float()
            faces = torch.empty((2 * n_quads, 3), dtype=torch.int64, device=dev)
            check(L.gs_flexi_quads(ptr(flags), ptr(qrank), ptr(topo.inc), ptr(vd_idx_map), ptr(vd_gamma), c_int64(E), ptr(faces), ptr(None),
                                   stream()), "gs_flexi_quads")

        if training:
            # ---- training=True (:523-551; no call site of the G-Shell scripts sets it): every quad becomes a fan of four triangles around a centre vertex -- the
            # midpoints of its two diagonals weighted by the products of the opposite gammas.  The quads are read back from the kernel's two triangles per quad
            # ((q0,q1,q2),(q0,q2,q3) or (q0,q1,q3),(q3,q1,q2): the second starts with q0 only in the first form); the rest is the reference's own arithmetic in its
            # own order, differentiable w.r.t. vd, nu_d and -- unlike the two-triangle split -- gamma.
            with torch.no_grad():
                f0, f1 = faces[0::2], faces[1::2]
                first = f1[:, 0] == f0[:, 0]
                quads = torch.where(first[:, None], torch.stack([f0[:, 0], f0[:, 1], f0[:, 2], f1[:, 2]], -1), torch.stack([f0[:, 0], f0[:, 1], f1[:, 2], f0[:, 2]], -1))
            g = gamma[vd_cube.long()][quads]                                                  # [Q,4], ONE use of gamma per dual vertex (:421)
            g02, g13 = g[:, 0:1] * g[:, 2:3], g[:, 1:2] * g[:, 3:4]
            vq, nq, nsq = vd[quads], nu_d[quads], nu_d_sv[quads]                             # [Q,4,3], [Q,4,1], [Q,4,1]
            mid = lambda t, a, b: (t[:, a:a + 1] + t[:, b:b + 1]) / 2
            wsum = (g02 + g13) + 1e-8
            vd_c = ((mid(vq, 0, 2) * g02.unsqueeze(-1) + mid(vq, 1, 3) * g13.unsqueeze(-1)) / wsum.unsqueeze(-1)).squeeze(1)
            nu_c = ((mid(nq, 0, 2) * g02.unsqueeze(-1) + mid(nq, 1, 3) * g13.unsqueeze(-1)) / wsum.unsqueeze(-1)).squeeze(1)
            nus_c = ((mid(nsq, 0, 2) * g02.unsqueeze(-1).detach() + mid(nsq, 1, 3) * g13.unsqueeze(-1).detach()) / wsum.unsqueeze(-1).detach()).squeeze(1)
            centre = torch.arange(quads.shape[0], device=dev) + n_vd
            vd, nu_d, nu_d_sv = torch.cat([vd, vd_c]), torch.cat([nu_d, nu_c]), torch.cat([nu_d_sv, nus_c])
            faces = torch.cat([quads[:, [0, 1, 1, 2, 2, 3, 3, 0]].reshape(-1, 4, 2), centre.reshape(-1, 1, 1).repeat(1, 4, 1)], -1).reshape(-1, 3)
            n_vd = int(vd.shape[0])                                                          # the cut numbers its vertices after ALL of these (:577)

        # ---- mSDF cut (:554-599)
        extra = {'n_verts_watertight': n_vd, 'vertices_watertight': vd, 'faces_watertight': faces, 'msdf_watertight': nu_d}
        with torch.no_grad():
            mocc = (nu_d.reshape(-1) >= 0)[faces]
            msum = mocc.sum(-1)
            uncut_mask, cut_mask = msum == 3, (msum < 3) & (msum > 0)
            uncut, cut = faces[uncut_mask], faces[cut_mask]
        if uncut.shape[0] == 0:         # reference quirk: nothing survives un-cut -> the UNCUT mesh is returned (:566-567)
            extra.update(msdf=nu_d, msdf_boundary=nu_d[:1].detach() * 0.0)
            return vd, faces, L_dev, extra
        pa, pb = cut[:, [0, 1, 2]].reshape(-1), cut[:, [1, 2, 0]].reshape(-1)

        bverts, bnu = _FlexiCutFn.apply(vd, nu_d.reshape(-1), nu_d_sv.reshape(-1), pa.contiguous(), pb.contiguous())
        bnu = bnu[:, None]
        with torch.no_grad():
            mc = mocc[cut_mask].long()
            cfg = mc[:, 0] * 4 + mc[:, 1] * 2 + mc[:, 2]
            idx_map = torch.cat([cut, n_vd + torch.arange(cut.shape[0] * 3, device=dev).reshape(-1, 3)], -1)
            if getattr(self, "_cut_tabs", None) is None or self._cut_tabs[0].device != cfg.device:
                self._cut_tabs = (torch.tensor(_CUT_N, device=dev), torch.tensor(_CUT_CFG, device=dev))
            cut_n, cut_cfg = self._cut_tabs[0][cfg], self._cut_tabs[1][cfg]
            one, two = cut_n == 1, cut_n == 2
            faces_open = torch.cat([uncut, torch.gather(idx_map[one], 1, cut_cfg[one][:, :3]).reshape(-1, 3),
                                    torch.gather(idx_map[two], 1, cut_cfg[two][:, :6]).reshape(-1, 3)])
        extra.update(msdf=torch.cat([nu_d_sv, bnu]), msdf_boundary=bnu)
        return torch.cat([vd, bverts]), faces_open, L_dev, extra

"""CPU oracle for G-FlexiCubes (TEST INFRASTRUCTURE -- checker only).

A torch-CPU restatement of the reference's geometry/gshell_flexicubes.py:136-230 (`GShellFlexiCubes.__call__`, the
non-training, grad_func=None path that every G-Shell script runs), written in the formulation the HIP path
uses: a STATIC per-grid edge table (sorted unique ordered edges, cube->edge ids, incident cube-edges per edge) and prefix
ranks instead of the per-call `torch.unique(..., return_counts)` (:317) / stable `torch.sort` (:496) / python loop over
`num_vd` groups (:406).  Autograd through this restatement is the gradient oracle.

Parity pin: tests/test_oracle_flexi.py checks this file against tests/golden/flexi_*.npz minted from the REAL reference by
oracle/make_golden_flexi.py (faces bit-exact, floats 1e-6, gradients 1e-5).
Case tables: oracle/flexi_tables.npz (packed by tools/gen_flexi_tables.py; data of the Dual Marching Cubes algorithm).

Reference quirks reproduced on purpose: inside is NEGATIVE here (`occ = s < 0`, :315,:339); the in-place
`index_add_` at :476-477 makes the returned nu_d = A/B + A' (A' = the same sum with detached weights) and
nu_d_stopvgd = nu_d / B; `mocc = nu >= 0` (:556); no uncut face -> the UNCUT mesh is returned (:566-567); empty surface ->
a 3-tuple (:193-202).
"""
import os

import numpy as np
import torch

_T = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "flexi_tables.npz"))
DMC = torch.tensor(_T["dmc"].astype(np.int64))            # [256,4,7] cube-edge ids of each dual vertex, -1 padded
NUM_VD = torch.tensor(_T["num_vd"].astype(np.int64))      # [256]
CHECK = torch.tensor(_T["check"].astype(np.int64))        # [256,5]  (ambiguous?, dx, dy, dz, inverted case)
CUT_N = torch.tensor(_T["cut_n"].astype(np.int64))        # [8]
CUT_CFG = torch.tensor(_T["cut_cfg"].astype(np.int64))    # [8,6]
# local cube edge e joins corners CUBE_EDGES[e] in THIS orientation (reference :88-89)
CUBE_EDGES = torch.tensor([[0, 1], [1, 5], [4, 5], [0, 4], [2, 3], [3, 7], [6, 7], [2, 6], [2, 0], [3, 1], [7, 5], [6, 4]])


def construct_voxel_grid(res):
    """Vertices ([ (res+1)^3, 3 ] in [-0.5, 0.5]) and cube corner indices of the reference's grid (:103-134): vertex index
    = lexicographic rank of (x, y, z); cube (i,j,k) is cube number (i*res + j)*res + k; corner c = (c&1, (c>>1)&1, c>>2)."""
    n = res + 1
    i, j, k = torch.meshgrid(torch.arange(n), torch.arange(n), torch.arange(n), indexing="ij")
    verts = torch.stack([i, j, k], -1).reshape(-1, 3).float() / res - 0.5
    ci, cj, ck = torch.meshgrid(torch.arange(res), torch.arange(res), torch.arange(res), indexing="ij")
    base = torch.stack([ci, cj, ck], -1).reshape(-1, 1, 3)
    corner = torch.tensor([[c & 1, (c >> 1) & 1, c >> 2] for c in range(8)])[None]
    p = base + corner
    cubes = (p[..., 0] * n + p[..., 1]) * n + p[..., 2]
    return verts, cubes


def build_topology(cubes, num_verts):
    pairs = cubes[:, CUBE_EDGES]                                        # [F,12,2] ordered pairs
    key = pairs[..., 0] * num_verts + pairs[..., 1]
    ukey, inv, counts = torch.unique(key.reshape(-1), return_inverse=True, return_counts=True)
    E = ukey.numel()
    edges = torch.stack([ukey // num_verts, ukey % num_verts], -1)
    order = torch.sort(inv, stable=True).indices                       # cube*12+e ascending inside each edge
    start = torch.cumsum(counts, 0) - counts
    inc = torch.full((E, 4), -1, dtype=torch.long)
    pos = torch.arange(order.numel()) - start[inv[order]]
    inc[inv[order], pos] = order
    return {"edges": edges, "cube_edge": inv.reshape(-1, 12), "ncubes": counts, "inc": inc}


def _interp(wa, wb, xa, xb):
    """zero crossing of the linear function with end values (wa, wb): reference _linear_interp (:346-357)."""
    return (xa * wb - xb * wa) / (wb - wa)


def extract(x, s, nu, cubes, res, beta=None, alpha=None, gamma=None, topo=None, weight_scale=0.99, training=False):
    N, F = x.shape[0], cubes.shape[0]
    s1 = s.reshape(-1)
    nu1 = nu.reshape(-1)
    if topo is None:
        topo = build_topology(cubes, N)
    edges, cube_edge, ncubes, inc = topo["edges"], topo["cube_edge"], topo["ncubes"], topo["inc"]
    occ = s1 < 0
    occ8 = occ[cubes]
    cnt = occ8.sum(-1)
    surf = (cnt > 0) & (cnt < 8)
    if int(surf.sum()) == 0:
        return torch.zeros((0, 3)), torch.zeros((0, 3), dtype=torch.long), torch.zeros((0,))
    dt = x.dtype
    beta = torch.ones(F, 12, dtype=dt) if beta is None else torch.tanh(beta) * weight_scale + 1
    alpha = torch.ones(F, 8, dtype=dt) if alpha is None else torch.tanh(alpha) * weight_scale + 1
    gamma = torch.ones(F, dtype=dt) if gamma is None else torch.sigmoid(gamma) * weight_scale + (1 - weight_scale) / 2

    # ---- case ids with the C16/C19 ambiguity fix-up (:266-306)
    case_raw = (occ8.long() * (2 ** torch.arange(8))).sum(-1)
    chk = CHECK[case_raw]
    rr = [res, res, res] if not isinstance(res, (list, tuple)) else list(res)
    cid = torch.arange(F)
    ijk = torch.stack([cid // (rr[1] * rr[2]), (cid // rr[2]) % rr[1], cid % rr[2]], -1)
    adj = ijk + chk[:, 1:4]
    inside = ((adj >= 0) & (adj < torch.tensor(rr))).all(-1)
    adj_id = ((adj[:, 0] * rr[1] + adj[:, 1]) * rr[2] + adj[:, 2]).clamp(0, F - 1)
    adj_amb = surf[adj_id] & (CHECK[case_raw[adj_id], 0] == 1)
    invert = surf & (chk[:, 0] == 1) & inside & adj_amb
    case = torch.where(invert, chk[:, 4], case_raw)
    num_vd = torch.where(surf, NUM_VD[case], torch.zeros_like(case))

    # ---- crossing edges -> rank; zero crossings
    ea, eb = edges[:, 0], edges[:, 1]
    cross = occ[ea] != occ[eb]

    # ---- dual vertices: ordered by (num_vd group ascending, cube, j); entries by (group, cube, j, slot)
    vd_base = torch.zeros(F, dtype=torch.long)
    total = 0
    ent_cube, ent_j, ent_slot = [], [], []
    for n in range(1, 5):
        sel = torch.nonzero(num_vd == n).reshape(-1)
        if sel.numel() == 0:
            continue
        vd_base[sel] = total + torch.arange(sel.numel()) * n
        total += sel.numel() * n
        c = sel[:, None, None].expand(-1, n, 7)
        j = torch.arange(n)[None, :, None].expand(sel.numel(), -1, 7)
        k = torch.arange(7)[None, None, :].expand(sel.numel(), n, -1)
        valid = DMC[case[sel]][:, :n] >= 0
        ent_cube.append(c[valid]); ent_j.append(j[valid]); ent_slot.append(k[valid])
    ent_cube, ent_j, ent_slot = torch.cat(ent_cube), torch.cat(ent_j), torch.cat(ent_slot)
    n_vd = total
    ent_e = DMC[case[ent_cube], ent_j, ent_slot]                          # local cube edge
    ent_vd = vd_base[ent_cube] + ent_j
    ent_edge = cube_edge[ent_cube, ent_e]                                 # global edge id
    a_id, b_id = edges[ent_edge, 0], edges[ent_edge, 1]
    al = alpha[ent_cube[:, None], CUBE_EDGES[ent_e]]                      # [n_ent,2] corner weights in edge orientation
    ca, cb = (s1[a_id] * al[:, 0])[:, None], (s1[b_id] * al[:, 1])[:, None]
    ue = _interp(ca, cb, x[a_id], x[b_id])
    nu_e = _interp(ca, cb, nu1[a_id, None], nu1[b_id, None])
    nu_e_sv = _interp(ca.detach(), cb.detach(), nu1[a_id, None], nu1[b_id, None])
    bt = beta[ent_cube, ent_e][:, None]
    beta_sum = torch.zeros(n_vd, 1, dtype=dt).index_add(0, ent_vd, bt)
    vd = torch.zeros(n_vd, 3, dtype=dt).index_add(0, ent_vd, ue * bt) / beta_sum
    nu_d = torch.zeros(n_vd, 1, dtype=dt).index_add(0, ent_vd, nu_e * bt) / beta_sum
    nu_d = nu_d.index_add(0, ent_vd, nu_e_sv * bt.detach())               # the reference's in-place quirk (:476-477)
    nu_d_sv = nu_d / beta_sum.detach()
    zc = _interp(s1[a_id, None], s1[b_id, None], x[a_id], x[b_id])        # un-weighted zero crossing of each entry's edge
    dist = (zc - vd[ent_vd]).norm(dim=-1)
    n_edges = torch.zeros(n_vd, dtype=dt).index_add(0, ent_vd, torch.ones_like(dist))
    mean_l2 = torch.zeros(n_vd, dtype=dt).index_add(0, ent_vd, dist) / n_edges
    L_dev = (dist - mean_l2[ent_vd]).abs()
    vd_cube = torch.zeros(n_vd, dtype=torch.long).index_put((ent_vd,), ent_cube)      # the cube of each dual vertex (all of a vertex's entries name the same cube)
    vd_gamma = gamma[vd_cube]                                                         # ONE use of gamma per dual vertex (:421): its gradient counts (training=True)
    vd_idx_map = torch.full((F, 12), -1, dtype=torch.long)
    vd_idx_map[ent_cube, ent_e] = ent_vd

    # ---- quads around crossing edges shared by 4 cubes, split by gamma (:487-522, non-training branch)
    qe = torch.nonzero(cross & (ncubes == 4)).reshape(-1)
    q = vd_idx_map.reshape(-1)[inc[qe]]                                   # [Q,4] in ascending cube order
    flip = s1[edges[qe, 0]] > 0
    quads = torch.cat([q[flip][:, [0, 1, 3, 2]], q[~flip][:, [2, 3, 1, 0]]])
    g = vd_gamma[quads]
    if training:
        # :523-551: a centre vertex per quad, the diagonals' midpoints weighted by the products of the opposite gammas; four fan triangles per quad
        g02, g13 = g[:, 0:1] * g[:, 2:3], g[:, 1:2] * g[:, 3:4]                         # [Q,1]
        vq, nq, nsq = vd[quads], nu_d[quads], nu_d_sv[quads]                           # [Q,4,3], [Q,4,1], [Q,4,1]
        mid = lambda t, a, b: (t[:, a:a + 1] + t[:, b:b + 1]) / 2
        wsum = (g02 + g13) + 1e-8
        vd_c = ((mid(vq, 0, 2) * g02.unsqueeze(-1) + mid(vq, 1, 3) * g13.unsqueeze(-1)) / wsum.unsqueeze(-1)).squeeze(1)
        nu_c = ((mid(nq, 0, 2) * g02.unsqueeze(-1) + mid(nq, 1, 3) * g13.unsqueeze(-1)) / wsum.unsqueeze(-1)).squeeze(1)
        nus_c = ((mid(nsq, 0, 2) * g02.unsqueeze(-1).detach() + mid(nsq, 1, 3) * g13.unsqueeze(-1).detach()) / wsum.unsqueeze(-1).detach()).squeeze(1)
        centre = torch.arange(quads.shape[0]) + n_vd
        vd, nu_d, nu_d_sv = torch.cat([vd, vd_c]), torch.cat([nu_d, nu_c]), torch.cat([nu_d_sv, nus_c])
        faces = torch.cat([quads[:, [0, 1, 1, 2, 2, 3, 3, 0]].reshape(-1, 4, 2), centre.reshape(-1, 1, 1).repeat(1, 4, 1)], -1).reshape(-1, 3)
        n_vd_dual, n_vd = n_vd, int(vd.shape[0])                                        # the cut numbers its vertices after ALL of these (:577)
    else:
        first = (g[:, 0] * g[:, 2]) > (g[:, 1] * g[:, 3])
        faces = torch.where(first[:, None], quads[:, [0, 1, 2, 0, 2, 3]], quads[:, [0, 1, 3, 3, 1, 2]]).reshape(-1, 3)
        n_vd_dual = n_vd

    # ---- mSDF cut of the triangles (:554-599)
    mocc = (nu_d.detach() >= 0).reshape(-1)[faces]
    msum = mocc.sum(-1)
    uncut, cut = faces[msum == 3], faces[(msum < 3) & (msum > 0)]
    extra = {"n_verts_watertight": n_vd, "vertices_watertight": vd, "faces_watertight": faces, "msdf_watertight": nu_d}
    if uncut.shape[0] == 0:
        extra.update(msdf=nu_d, msdf_boundary=nu_d[:1].detach() * 0.0)
        return vd, faces, L_dev, extra
    pa, pb = cut[:, [0, 1, 2]].reshape(-1), cut[:, [1, 2, 0]].reshape(-1)

    def interp_nonan(wa, wb, xa, xb):      # _linear_interp_nonan (:359-373): zero weights where the denominator vanishes
        den = wb - wa
        ok = den.abs() > 0
        safe = torch.where(ok, den, torch.ones_like(den))
        return xa * torch.where(ok, wb / safe, torch.zeros_like(den)) + xb * torch.where(ok, -wa / safe, torch.zeros_like(den))
    bverts = interp_nonan(nu_d[pa], nu_d[pb], vd[pa], vd[pb])
    bnu = interp_nonan(nu_d_sv[pa].detach(), nu_d_sv[pb].detach(), nu_d_sv[pa], nu_d_sv[pb])
    verts_open = torch.cat([vd, bverts])
    nus_open = torch.cat([nu_d_sv, bnu])
    mc = mocc[(msum < 3) & (msum > 0)].long()
    cfg = mc[:, 0] * 4 + mc[:, 1] * 2 + mc[:, 2]
    idx_map = torch.cat([cut, n_vd + torch.arange(cut.shape[0] * 3).reshape(-1, 3)], -1)
    ntri = CUT_N[cfg]
    one, two = ntri == 1, ntri == 2
    faces_open = torch.cat([uncut, torch.gather(idx_map[one], 1, CUT_CFG[cfg[one]][:, :3]).reshape(-1, 3),
                            torch.gather(idx_map[two], 1, CUT_CFG[cfg[two]][:, :6]).reshape(-1, 3)])
    extra.update(msdf=nus_open, msdf_boundary=bnu)
    with torch.no_grad():
        # Forward-error scale of each boundary vertex (a length; multiply by a few float32 eps).  x = x_a + w (x_b - x_a) with
        # w = nu_a / (nu_a - nu_b); nu_d of a dual vertex is a sum over its <= 7 entries (twice: the index_add_ quirk), exact only
        # to eps * A with A = sum |nu_e beta| (1 + 1 / sum beta) -- far more than eps |nu_d| when the terms cancel.  So
        #   |dx| <~ eps |x_a - x_b| (|nu_b| A_a + |nu_a| A_b) / (nu_b - nu_a)^2.
        # Large (i) on edges whose end points lie on the SAME side of the cut (every edge of a cut triangle gets a vertex; those
        # extrapolate and no face references them) and (ii) where both values are noise around zero.
        A = torch.zeros(n_vd_dual, 1, dtype=dt).index_add(0, ent_vd, (nu_e_sv * bt).abs().detach())
        A = (A * (1.0 + 1.0 / beta_sum.detach())).reshape(-1)
        A = torch.cat([A, torch.zeros(n_vd - n_vd_dual, dtype=dt)])                    # training: centre vertices (averages of four dual vertices) carry no estimate of their own
        wa, wb = nu_d[pa].reshape(-1), nu_d[pb].reshape(-1)
        den2 = (wb - wa) ** 2
        amp = torch.where(den2 > 0, (wb.abs() * A[pa] + wa.abs() * A[pb]) / den2.clamp_min(1e-38), torch.zeros_like(den2))
        extra["boundary_cond"] = amp * (vd[pa] - vd[pb]).norm(dim=-1)
    return verts_open, faces_open, L_dev, extra

"""CPU oracle for G-MarchingTets (TEST INFRASTRUCTURE -- checker only).

A torch-CPU restatement of the algorithm in the reference's
geometry/gshell_tets.py:245-443 (`GShell_Tets.__call__`), written in the
formulation the HIP path uses (static sorted edge list + prefix ranks instead of
per-call `torch.unique`), so it doubles as the executable spec of the kernels.
Autograd through this restatement is the gradient oracle.

Parity pin: `tests/test_oracle_mtets.py` checks this file against the golden
fixtures minted from the *real* reference by oracle/make_golden_mtets.py
(faces bit-exact, float outputs bit-exact on CPU, grads to 1e-5).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (gshell_amd/) never does.
"""
import numpy as np
import torch

# ---- case tables (data of the algorithm; reference gshell_tets.py:82-181) ------------
# Local tet edge k joins tet corners EDGE_CORNERS[k]           (ref :178 base_tet_edges)
EDGE_CORNERS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))
# Triangles of the watertight surface per sign pattern (local edge ids)   (ref :82-99)
TRI_TABLE = (
    (), (1, 0, 2), (4, 0, 3), (1, 4, 2, 1, 3, 4), (3, 1, 5), (2, 3, 0, 2, 5, 3), (1, 4, 0, 1, 5, 4), (4, 2, 5),
    (4, 5, 2), (4, 1, 0, 4, 5, 1), (3, 2, 0, 3, 5, 2), (1, 3, 5), (4, 1, 2, 4, 3, 1), (3, 0, 4), (2, 0, 1), ())
# Polygon (3- or 4-gon) boundary loop per sign pattern (local edge ids)   (ref :101-118)
POLY_TABLE = (
    (), (1, 0, 2), (4, 0, 3), (1, 3, 4, 2), (3, 1, 5), (2, 5, 3, 0), (1, 5, 4, 0), (4, 2, 5),
    (4, 5, 2), (4, 5, 1, 0), (3, 5, 2, 0), (1, 3, 5), (4, 3, 1, 2), (3, 0, 4), (2, 0, 1), ())
# mSDF cut of a triangle: ids 0-2 = polygon corners, 3-5 = boundary points on edges
# (c0c1, c1c2, c2c0); index = m0*4 + m1*2 + m2                        (ref :121-138)
CUT_TRI = ((), (4, 2, 5), (3, 1, 4), (3, 1, 2, 3, 2, 5), (0, 3, 5), (0, 3, 4, 0, 4, 2), (0, 1, 4, 0, 4, 5), (0, 1, 2))
# mSDF cut of a quad: ids 0-3 corners, 4-7 boundary points on (c0c1, c1c2, c2c3, c3c0);
# index = m0*8 + m1*4 + m2*2 + m3                                      (ref :140-175)
CUT_QUAD = (
    (), (6, 3, 7), (5, 2, 6), (5, 2, 7, 3, 7, 2), (4, 1, 5), (4, 1, 5, 4, 5, 7, 5, 6, 7, 7, 6, 3),
    (4, 1, 2, 6, 4, 2), (4, 1, 2, 7, 4, 2, 7, 2, 3), (0, 4, 7), (0, 4, 6, 3, 0, 6),
    (0, 4, 5, 0, 5, 2, 0, 2, 6, 0, 6, 7), (0, 4, 5, 0, 5, 2, 0, 2, 3), (0, 1, 5, 7, 0, 5),
    (0, 1, 5, 0, 5, 6, 0, 6, 3), (0, 1, 2, 0, 2, 6, 0, 6, 7), (0, 1, 2, 0, 2, 3))


def _pad(table, width, device=None):
    return torch.tensor([list(r) + [0] * (width - len(r)) for r in table], dtype=torch.long, device=device)


def build_topology(tets: torch.Tensor):
    """Static per-grid data: sorted unique edge list and tet->edge ids.

    Vertex ids of the extracted mesh are ranks of sign-crossing edges in the
    lexicographic (min,max) order that `torch.unique(dim=0)` produces in the
    reference (gshell_tets.py:266-276); ranking over this static list is
    equivalent and needs no per-call sort.
    """
    t = tets.long()
    ec = torch.tensor(EDGE_CORNERS, device=t.device)
    a, b = t[:, ec[:, 0]], t[:, ec[:, 1]]
    key = torch.minimum(a, b) * (int(t.max()) + 1) + torch.maximum(a, b)      # [F,6]
    ukey, inv = torch.unique(key.reshape(-1), return_inverse=True)
    n = int(t.max()) + 1
    edges = torch.stack([ukey // n, ukey % n], -1)                            # [E,2] sorted
    return {"edges": edges, "tet_edge": inv.reshape(-1, 6)}


def _edge_weights(xa, xb):
    """Zero crossing of a linear function with end values xa, xb (ref :278-285)."""
    x1 = xb * -1.0
    d = xa + x1
    den = torch.sign(d) * (d.abs() + 1e-12)
    den = torch.where(den == 0, torch.full_like(den, 1e-12), den)
    return x1 / den, xa / den


def extract(pos, sdf, msdf, tets, topo=None, with_tangents=True, output_watertight_template=True, _uv_tets=None):
    """Returns dict with the reference's outputs (names follow ref :426-443).  output_watertight_template=False (ref :260-263): tets whose four mSDF values
    are all <= 0 are dropped BEFORE anything else -- the edge set, hence the vertex numbering, is that of the remaining tets -- and the three
    `*_watertight` mesh entries are not returned (ref :436-441)."""
    if not output_watertight_template:
        keep = (msdf.reshape(-1)[tets.reshape(-1)].reshape(-1, 4) > 0).sum(-1) > 0        # ref :256-258
        if int(keep.sum()) == 0:                                                           # the reference's gathers over empty index sets
            e3, e1 = pos.new_zeros((0, 3)), pos.new_zeros((0,))
            return {"verts_aug": e3, "faces_aug": torch.zeros((0, 3), dtype=torch.long), "v_tng_aug": e3, "msdf": e1, "msdf_watertight": e1, "msdf_boundary": e1}
        out = extract(pos, sdf, msdf, tets[keep], None, with_tangents, True, _uv_tets=tets.shape[0])      # ref :301, :309: the uv atlas is sized by the WHOLE grid
        for k in ("n_verts_watertight", "vertices_watertight", "faces_watertight", "v_tng_watertight"):
            out["_" + k] = out.pop(k)               # kept under a private name (tests address the referenced rows through them)
        return out
    if topo is None:
        topo = build_topology(tets)
    edges, tet_edge = topo["edges"], topo["tet_edge"]
    dev = pos.device       # the restatement is plain torch: it also runs on the GPU box at the full BASELINE grid sizes
    sdf = (sdf if sdf.dtype == torch.float64 else sdf.float()).reshape(-1)     # float64: the arbiter runs of oracle/make_golden_chain.py
    msdf = msdf.reshape(-1)
    F = tets.shape[0]
    occ = sdf > 0                                                   # ref :250 (strict)
    occ4 = occ[tets.reshape(-1)].reshape(F, 4).long()
    code = occ4[:, 0] + 2 * occ4[:, 1] + 4 * occ4[:, 2] + 8 * occ4[:, 3]     # ref :296-297
    ntri = torch.tensor([len(r) // 3 for r in TRI_TABLE], device=dev)[code]

    ea, eb = edges[:, 0], edges[:, 1]
    cross = occ[ea] != occ[eb]
    vid_of_edge = torch.cumsum(cross.long(), 0) - 1
    vid_of_edge = torch.where(cross, vid_of_edge, torch.full_like(vid_of_edge, -1))
    va, vb = ea[cross], eb[cross]                                   # ref :276 interp_v
    V = int(va.shape[0])
    wa, wb = _edge_weights(sdf[va], sdf[vb])
    verts = pos[va] * wa[:, None] + pos[vb] * wb[:, None]           # ref :286
    mv = msdf[va] * wa + msdf[vb] * wb                              # ref :289
    mv_sg = msdf[va] * wa.detach() + msdf[vb] * wb.detach()         # ref :290

    tet1 = torch.nonzero(ntri == 1).reshape(-1)                     # tet order preserved
    tet2 = torch.nonzero(ntri == 2).reshape(-1)
    M1, M2 = int(tet1.shape[0]), int(tet2.shape[0])
    tri_t, poly_t = _pad(TRI_TABLE, 6, dev), _pad(POLY_TABLE, 4, dev)
    vid1 = vid_of_edge[tet_edge[tet1]]                              # [M1,6]
    vid2 = vid_of_edge[tet_edge[tet2]]
    faces_wt = torch.cat([
        torch.gather(vid1, 1, tri_t[code[tet1]][:, :3]).reshape(-1, 3),
        torch.gather(vid2, 1, tri_t[code[tet2]][:, :6]).reshape(-1, 3)], 0)   # ref :313-316

    poly1 = torch.gather(vid1, 1, poly_t[code[tet1]][:, :3])        # [M1,3] polygon corners
    poly2 = torch.gather(vid2, 1, poly_t[code[tet2]][:, :4])        # [M2,4]

    v_tng = _tangents(verts, faces_wt, tet1, tet2, F if _uv_tets is None else int(_uv_tets)) if with_tangents else torch.zeros_like(verts)

    def boundary(poly):
        a = poly
        b = torch.roll(poly, -1, dims=1)                            # loop edges (ck, ck+1)
        ma, mb = mv[a], mv[b]
        x1 = mb * -1.0
        den = ma + x1
        nz = ((torch.sign(ma) + torch.sign(mb)).abs() != 2) & (den.abs() > 1e-12)   # ref :346-355
        safe = torch.where(nz, den, torch.ones_like(den))
        w_a = torch.where(nz, x1 / safe, torch.zeros_like(den))
        w_b = torch.where(nz, ma / safe, torch.zeros_like(den))
        p = verts[a] * w_a[..., None] + verts[b] * w_b[..., None]   # ref :368-373
        tg = v_tng[a] * w_a[..., None] + v_tng[b] * w_b[..., None]
        ms = mv_sg[a] * w_a.detach() + mv_sg[b] * w_b.detach()     # ref :383-384
        return p.reshape(-1, 3), tg.reshape(-1, 3), ms.reshape(-1)

    p1, t1, m1 = boundary(poly1)
    p2, t2, m2 = boundary(poly2)
    verts_aug = torch.cat([verts, p1, p2], 0)
    v_tng_aug = torch.cat([v_tng, t1, t2], 0)
    msdf_aug = torch.cat([mv_sg, m1, m2], 0)

    mocc1 = (mv[poly1] > 0).long()                                  # ref :330-331
    mocc2 = (mv[poly2] > 0).long()
    ci1 = mocc1[:, 0] * 4 + mocc1[:, 1] * 2 + mocc1[:, 2]           # ref :396-399 (flipped powers)
    ci2 = mocc2[:, 0] * 8 + mocc2[:, 1] * 4 + mocc2[:, 2] * 2 + mocc2[:, 3]
    loc1 = torch.cat([poly1, V + torch.arange(3 * M1, device=dev).reshape(-1, 3)], 1)             # ref :402
    loc2 = torch.cat([poly2, V + 3 * M1 + torch.arange(4 * M2, device=dev).reshape(-1, 4)], 1)    # ref :403
    cut1, cut2 = _pad(CUT_TRI, 6, dev), _pad(CUT_QUAD, 12, dev)
    n1 = torch.tensor([len(r) // 3 for r in CUT_TRI], device=dev)[ci1]
    n2 = torch.tensor([len(r) // 3 for r in CUT_QUAD], device=dev)[ci2]
    groups = []
    for k in (1, 2):                                                # ref :409-416 group order
        sel = n1 == k
        groups.append(torch.gather(loc1[sel], 1, cut1[ci1[sel]][:, :3 * k]).reshape(-1, 3))
    for k in (1, 2, 3, 4):
        sel = n2 == k
        groups.append(torch.gather(loc2[sel], 1, cut2[ci2[sel]][:, :3 * k]).reshape(-1, 3))
    faces_aug = torch.cat(groups, 0)

    used = torch.zeros(verts_aug.shape[0], dtype=torch.bool, device=dev)        # ref :419-423
    used[faces_aug.reshape(-1)] = True
    verts_aug = torch.where(used[:, None], verts_aug, torch.zeros_like(verts_aug))

    return {
        "verts_aug": verts_aug, "faces_aug": faces_aug, "v_tng_aug": v_tng_aug,
        "n_verts_watertight": V, "vertices_watertight": verts, "faces_watertight": faces_wt,
        "v_tng_watertight": v_tng, "msdf": msdf_aug, "msdf_watertight": mv_sg,
        "msdf_boundary": msdf_aug[V:], "used": used, "tet1": tet1, "tet2": tet2,
    }


def _tangents(verts, faces, tet1, tet2, F, return_cond=False):
    """Smooth normals + uv tangents of the watertight mesh (ref :9-78, :210-239, :301-319).
    return_cond: also the per-vertex condition number of the result w.r.t. the ORDER of the two scatter sums (the reference
    accumulates them with scatter_add_, i.e. float atomics in arbitrary order on a GPU): (sum m(t_f) / |sum t_f| + sum m(n_f) /
    |sum n_f|) / |t - (t.n) n|, m() = the magnitude of a term including its own rounding -- the factor by which float32 round-off
    of the terms and of the partial sums is amplified in the unit tangent (infinite where every face at the vertex is degenerate).

    Reference quirk (ref :319): compute_tangents is called with t_tex_idx = faces, so
    the uv of mesh vertex v is entry v of the atlas table built by map_uv, i.e. corner
    (v % 4) of atlas cell (v // 4) on an N x N grid, N = ceil(sqrt(F)) -- the per-face
    uv_idx is never used."""
    if faces.shape[0] == 0:
        return torch.zeros_like(verts)
    N = int(np.ceil(np.sqrt((2 * F + 1) // 2)))
    lin = torch.linspace(0, 1 - (1 / N), N, dtype=torch.float32)
    pad = 0.9 / N

    def uv_of(v):
        cell, k = torch.div(v, 4, rounding_mode="trunc"), v % 4
        tx, ty = lin[cell % N], lin[torch.div(cell, N, rounding_mode="trunc")]
        u = torch.where((k == 1) | (k == 2), tx + pad, tx)
        w = torch.where(k >= 2, ty + pad, ty)
        return torch.stack([u, w], -1)
    i0, i1, i2 = faces[:, 0], faces[:, 1], faces[:, 2]
    uv0, uv1, uv2 = uv_of(i0), uv_of(i1), uv_of(i2)
    p0, p1, p2 = verts[i0], verts[i1], verts[i2]
    fn = torch.linalg.cross(p1 - p0, p2 - p0)
    nrm = torch.zeros_like(verts)
    for idx in (i0, i1, i2):
        nrm = nrm.index_add(0, idx, fn)
    d = (nrm * nrm).sum(-1, keepdim=True)
    nrm = torch.where(d > 1e-20, nrm, torch.tensor([0.0, 0.0, 1.0]))
    nrm = nrm / torch.sqrt(torch.clamp((nrm * nrm).sum(-1, keepdim=True), min=1e-20))
    e1, e2 = uv1 - uv0, uv2 - uv0
    q1, q2 = p1 - p0, p2 - p0
    nom = q1 * e2[:, 1:2] - q2 * e1[:, 1:2]
    den = e1[:, 0:1] * e2[:, 1:2] - e1[:, 1:2] * e2[:, 0:1]
    tang = nom / torch.where(den > 0.0, torch.clamp(den, min=1e-6), torch.clamp(den, max=-1e-6))
    tsum = torch.zeros_like(verts)
    cnt = torch.zeros_like(verts)
    for idx in (i0, i1, i2):
        tsum = tsum.index_add(0, idx, tang)
        cnt = cnt.index_add(0, idx, torch.ones_like(tang))
    t = tsum / cnt

    def nz(x):
        return x / torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=1e-20))
    t = nz(t)
    out = nz(t - (t * nrm).sum(-1, keepdim=True) * nrm)
    if not return_cond:
        return out
    with torch.no_grad():
        # magnitudes of what is summed, INCLUDING the rounding of each term: a face normal e1 x e2 is exact only to eps |e1| |e2|
        # (a degenerate face leaves a residue of that size whose sign depends on whether the products were contracted to fma --
        # torch does on the CPU and on CUDA, numpy and -ffp-contract=off code do not), a face tangent to eps (|q1| |e2y| + |q2| |e1y|) / |den|
        den_c = torch.where(den > 0.0, torch.clamp(den, min=1e-6), torch.clamp(den, max=-1e-6)).abs().reshape(-1)
        m_t = (q1.norm(dim=-1) * e2[:, 1].abs() + q2.norm(dim=-1) * e1[:, 1].abs()) / den_c
        m_n = q1.norm(dim=-1) * q2.norm(dim=-1)
        a_t, a_n, s_n = torch.zeros(verts.shape[0]), torch.zeros(verts.shape[0]), torch.zeros_like(verts)
        for idx in (i0, i1, i2):
            a_t = a_t.index_add(0, idx, m_t)
            a_n = a_n.index_add(0, idx, m_n)
            s_n = s_n.index_add(0, idx, fn)
        tiny = 1e-30
        perp = (t - (t * nrm).sum(-1, keepdim=True) * nrm).norm(dim=-1)
        cond = (a_t / tsum.norm(dim=-1).clamp_min(tiny) + a_n / s_n.norm(dim=-1).clamp_min(tiny)) / perp.clamp_min(tiny)
    return out, cond


def extract_from_auggrid(pos, sdf, tets, verts_disc, coeff_grid, msdf_grid, occgrid, topo=None, with_tangents=True):
    """Generative-decode extraction (ref gshell_tets.py:446-629, `marching_from_auggrid`).

    Same topology / numbering as `extract`; per-edge quantities are looked up in cubic
    grids at canonical edge midpoints.  `verts_disc` [N,3] holds integer cells (the
    reference stores them as floats, gshell_tets_geometry.py:72-78).  No autograd.
    Parity pin: tests/test_oracle_mtets.py against goldens minted from the real
    reference by oracle/make_golden_auggrid.py."""
    if topo is None:
        topo = build_topology(tets)
    edges, tet_edge = topo["edges"], topo["tet_edge"]
    dev = pos.device       # the restatement is plain torch: it also runs on the GPU box at the full BASELINE grid sizes
    sdf = sdf.float().reshape(-1)
    vd = verts_disc.float()
    F = tets.shape[0]
    occ = sdf > 0                                                   # ref :454
    occ4 = occ[tets.reshape(-1)].reshape(F, 4).long()
    code = occ4[:, 0] + 2 * occ4[:, 1] + 4 * occ4[:, 2] + 8 * occ4[:, 3]     # ref :459-460
    ntri = torch.tensor([len(r) // 3 for r in TRI_TABLE], device=dev)[code]
    ea, eb = edges[:, 0], edges[:, 1]
    cross = occ[ea] != occ[eb]                                      # ref :470
    vid_of_edge = torch.cumsum(cross.long(), 0) - 1
    vid_of_edge = torch.where(cross, vid_of_edge, torch.full_like(vid_of_edge, -1))
    va, vb = ea[cross], eb[cross]
    V = int(va.shape[0])
    canon = (vd[va] + vd[vb]) / 2.0                                 # ref :478-479
    mid = torch.stack([vd[va], vd[vb]], 1).mean(dim=1).long()       # ref :481
    c = coeff_grid[mid[:, 0], mid[:, 1], mid[:, 2]].reshape(-1, 1).clamp(0, 1)      # ref :483
    verts = pos[vb] * c + pos[va] * (1 - c)                         # ref :484
    mv = msdf_grid[mid[:, 0], mid[:, 1], mid[:, 2]]                 # ref :486

    tet1 = torch.nonzero(ntri == 1).reshape(-1)
    tet2 = torch.nonzero(ntri == 2).reshape(-1)
    M1, M2 = int(tet1.shape[0]), int(tet2.shape[0])
    tri_t, poly_t = _pad(TRI_TABLE, 6, dev), _pad(POLY_TABLE, 4, dev)
    vid1 = vid_of_edge[tet_edge[tet1]]
    vid2 = vid_of_edge[tet_edge[tet2]]
    faces_wt = torch.cat([
        torch.gather(vid1, 1, tri_t[code[tet1]][:, :3]).reshape(-1, 3),
        torch.gather(vid2, 1, tri_t[code[tet2]][:, :6]).reshape(-1, 3)], 0)   # ref :513-516
    poly1 = torch.gather(vid1, 1, poly_t[code[tet1]][:, :3])        # ref :533-538
    poly2 = torch.gather(vid2, 1, poly_t[code[tet2]][:, :4])
    v_tng = _tangents(verts, faces_wt, tet1, tet2, F) if with_tangents else torch.zeros_like(verts)

    def boundary(poly):
        a, b = poly, torch.roll(poly, -1, dims=1)
        m0, m1 = canon[a], canon[b]                                 # ref :539-540
        loc = (torch.stack([m0, m1], 2).mean(dim=2) * 2.0).long()   # ref :543-544
        cg = occgrid[loc[..., 0], loc[..., 1], loc[..., 2]] * 0.5 + 0.5          # ref :547-548
        order = (torch.sign(m0 - m1) * torch.tensor([16.0, 4.0, 1.0])).sum(-1)     # ref :556-558
        first = order > 0                                           # ref :559 (descending sort of [o,-o])
        w_a = torch.where(first, cg, 1 - cg)
        w_b = torch.where(first, 1 - cg, cg)
        p = verts[a] * w_a[..., None] + verts[b] * w_b[..., None]   # ref :584-585
        tg = v_tng[a] * w_a[..., None] + v_tng[b] * w_b[..., None]  # ref :591-592
        return p.reshape(-1, 3), tg.reshape(-1, 3)

    p1, t1 = boundary(poly1)
    p2, t2 = boundary(poly2)
    verts_aug = torch.cat([verts, p1, p2], 0)
    v_tng_aug = torch.cat([v_tng, t1, t2], 0)
    msdf_aug = torch.cat([mv, torch.zeros(verts_aug.shape[0] - V)], 0)        # ref :597-600

    mocc1 = (mv[poly1] > 0).long()                                  # ref :540-541
    mocc2 = (mv[poly2] > 0).long()
    ci1 = mocc1[:, 0] * 4 + mocc1[:, 1] * 2 + mocc1[:, 2]           # ref :602-606
    ci2 = mocc2[:, 0] * 8 + mocc2[:, 1] * 4 + mocc2[:, 2] * 2 + mocc2[:, 3]
    loc1 = torch.cat([poly1, V + torch.arange(3 * M1, device=dev).reshape(-1, 3)], 1)             # ref :608
    loc2 = torch.cat([poly2, V + 3 * M1 + torch.arange(4 * M2, device=dev).reshape(-1, 4)], 1)    # ref :609
    cut1, cut2 = _pad(CUT_TRI, 6, dev), _pad(CUT_QUAD, 12, dev)
    n1 = torch.tensor([len(r) // 3 for r in CUT_TRI], device=dev)[ci1]
    n2 = torch.tensor([len(r) // 3 for r in CUT_QUAD], device=dev)[ci2]
    groups = []
    for k in (1, 2):                                                # ref :614-621
        sel = n1 == k
        groups.append(torch.gather(loc1[sel], 1, cut1[ci1[sel]][:, :3 * k]).reshape(-1, 3))
    for k in (1, 2, 3, 4):
        sel = n2 == k
        groups.append(torch.gather(loc2[sel], 1, cut2[ci2[sel]][:, :3 * k]).reshape(-1, 3))
    faces_aug = torch.cat(groups, 0)
    return {
        "verts_aug": verts_aug, "faces_aug": faces_aug, "v_tng_aug": v_tng_aug,
        "vertices_watertight": verts, "faces_watertight": faces_wt,
        "valid_tet_gidx": torch.cat([tet1, tet2], 0), "msdf": msdf_aug, "msdf_watertight": mv,
    }

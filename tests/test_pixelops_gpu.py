"""Per-pixel / per-vertex ops: HIP path vs golden vectors from the real reference and vs the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import pixel_oracle as po
from oracle import scenes

pytestmark = pytest.mark.gpu
DEV = "cuda"
G = os.path.join(os.path.dirname(__file__), "golden")


def test_shading_normal_golden():
    from gshell_amd.render import renderutils as ru
    g = np.load(os.path.join(G, "pixelops_shading_normal.npz"))
    wgt = torch.tensor(g["w"], device=DEV)
    for tag in ("nopert", "pert"):
        for two_sided in (True, False):
            key = f"{tag}_{int(two_sided)}"
            leaves = {k: torch.tensor(g[f"in_{k}"], device=DEV).requires_grad_(True) for k in ("pos", "view_pos", "smooth_nrm", "smooth_tng", "geom_nrm")}
            pn = torch.tensor(g["in_perturbed_nrm"], device=DEV).requires_grad_(True) if tag == "pert" else None
            out = ru.prepare_shading_normal(leaves["pos"], leaves["view_pos"], pn, leaves["smooth_nrm"], leaves["smooth_tng"], leaves["geom_nrm"],
                                            two_sided_shading=two_sided, opengl=True)
            (out * wgt).sum().backward()
            # tolerance: north_star 1e-4 relative fp32
            np.testing.assert_allclose(out.detach().cpu().numpy(), g[f"out_{key}"], rtol=1e-4, atol=1e-6)
            for k, v in leaves.items():
                ref = g[f"g_{k}_{key}"]
                np.testing.assert_allclose(v.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref).max()), err_msg=f"{k} {key}")
            if pn is not None:
                ref = g[f"g_perturbed_nrm_{key}"]
                np.testing.assert_allclose(pn.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref).max()))


def test_shading_normal_large_random():
    from gshell_amd.render import renderutils as ru
    gen = torch.Generator().manual_seed(3)
    B, H, W = 2, 64, 48
    ins = [torch.randn(B, H, W, 3, generator=gen) for _ in range(4)]
    vp = torch.randn(B, 1, 1, 3, generator=gen) * 3
    wgt = torch.randn(B, H, W, 3, generator=gen)
    ref_in = [t.clone().requires_grad_(True) for t in ins]
    ref = po.prepare_shading_normal(ref_in[0], vp, None, ref_in[1], ref_in[2], ref_in[3])
    (ref * wgt).sum().backward()
    dev_in = [t.to(DEV).requires_grad_(True) for t in ins]
    out = ru.prepare_shading_normal(dev_in[0], vp.to(DEV), None, dev_in[1], dev_in[2], dev_in[3])
    (out * wgt.to(DEV)).sum().backward()
    assert torch.allclose(out.cpu(), ref.detach(), rtol=1e-4, atol=1e-6)
    for a, b in zip(dev_in, ref_in):
        assert (a.grad.cpu() - b.grad).abs().max() <= 1e-4 * b.grad.abs().max() + 1e-6


@pytest.mark.parametrize("loss", ["l1", "mse", "smape", "relmse"])
@pytest.mark.parametrize("tm", ["none", "log_srgb"])
def test_image_loss(loss, tm):
    from gshell_amd.render import renderutils as ru
    gen = torch.Generator().manual_seed(11)
    img = torch.rand(2, 33, 47, 3, generator=gen) * 1.6 - 0.1
    tgt = torch.rand(2, 33, 47, 3, generator=gen) * 1.6
    a_ref = img.clone().requires_grad_(True)
    v_ref = po.image_loss(a_ref, tgt, loss, tm)
    v_ref.backward()
    a = img.to(DEV).requires_grad_(True)
    v = ru.image_loss(a, tgt.to(DEV), loss=loss, tonemapper=tm)
    v.backward()
    assert torch.allclose(v.cpu(), v_ref.detach(), rtol=1e-5)
    assert (a.grad.cpu() - a_ref.grad).abs().max() <= 1e-4 * a_ref.grad.abs().max()
    if tm == "none" and loss != "smape":   # the reference kernel and its python twin agree here: golden from the twin
        g = np.load(os.path.join(G, "pixelops_image_loss.npz"))
        gi = torch.tensor(g["in_img"]).clamp(0, 65535).to(DEV)
        v = ru.image_loss(gi, torch.tensor(g["in_target"], device=DEV), loss=loss, tonemapper=tm)
        assert np.isclose(float(v), float(po.image_loss(gi.cpu(), torch.tensor(g["in_target"]), loss, tm, twin=True)), rtol=1e-5)


def test_auto_normals_golden_and_random():
    from gshell_amd.render import mesh
    g = np.load(os.path.join(G, "pixelops_auto_normals.npz"))
    v = torch.tensor(g["in_verts"], device=DEV).requires_grad_(True)
    m = mesh.auto_normals(mesh.Mesh(v, torch.tensor(g["in_tri"], device=DEV).long()))
    (m.v_nrm * torch.tensor(g["w"], device=DEV)).sum().backward()
    np.testing.assert_allclose(m.v_nrm.detach().cpu().numpy(), g["out"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(v.grad.cpu().numpy(), g["g_verts"], rtol=1e-4, atol=1e-5)
    assert m.t_nrm_idx is m.t_pos_idx
    verts, tri = scenes.grid_sheet(40, seed=5)
    vr = torch.tensor(verts).requires_grad_(True)
    w = torch.randn(vr.shape, generator=torch.Generator().manual_seed(1))
    ref = po.auto_normals(vr, torch.tensor(tri).long())
    (ref * w).sum().backward()
    vd = torch.tensor(verts, device=DEV).requires_grad_(True)
    out = mesh.auto_normals(mesh.Mesh(vd, torch.tensor(tri, device=DEV).long())).v_nrm
    (out * w.to(DEV)).sum().backward()
    assert torch.allclose(out.cpu(), ref.detach(), rtol=1e-4, atol=1e-6)
    assert (vd.grad.cpu() - vr.grad).abs().max() <= 1e-4 * vr.grad.abs().max()


def test_texture_linear_clamp():
    from gshell_amd.render import rast as dr
    gen = torch.Generator().manual_seed(2)
    tex = torch.rand(2, 24, 31, 3, generator=gen)
    uv = po.pixel_grid(31, 24)[None].repeat(2, 1, 1, 1) + torch.randn(2, 24, 31, 2, generator=gen) * 0.02
    w = torch.rand(2, 24, 31, 3, generator=gen)
    t_ref = tex.clone().requires_grad_(True)
    ref = po.texture_linear_clamp(t_ref, uv)
    (ref * w).sum().backward()
    t = tex.to(DEV).requires_grad_(True)
    out = dr.texture(t, uv.to(DEV), filter_mode='linear', boundary_mode='clamp')
    (out * w.to(DEV)).sum().backward()
    assert torch.allclose(out.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(t.grad.cpu(), t_ref.grad, rtol=1e-4, atol=1e-5)


def test_sdf_reg_loss():
    from gshell_amd import grid
    from gshell_amd.geometry.gshell_tets import GShell_Tets
    from gshell_amd.geometry.gshell_tets_geometry import compute_sdf_reg_loss
    verts, tets = grid.bcc_grid(10)
    topo = GShell_Tets().topology(tets.to(DEV), verts.shape[0])
    edges = topo.edges()
    gen = torch.Generator().manual_seed(0)
    sdf = (verts.norm(dim=1) - 0.3 + 0.05 * torch.randn(verts.shape[0], generator=gen))[:, None]   # [N,1] like the MLP output
    sdf[:5] = 0.0                                                                                   # sign(0) == 0 counts as crossing
    s_ref = sdf.clone().requires_grad_(True)
    l_ref = po.sdf_reg_loss(s_ref, edges.cpu().long())
    l_ref.backward()
    s = sdf.to(DEV).requires_grad_(True)
    l = compute_sdf_reg_loss(s, edges)
    (l * 1.0).backward()
    assert abs(float(l) - float(l_ref)) <= 1e-5 * abs(float(l_ref))
    assert (s.grad.cpu() - s_ref.grad).abs().max() <= 1e-4 * s_ref.grad.abs().max()
    # no crossing edge: loss 0, zero gradient, no NaN
    s2 = torch.ones(verts.shape[0], device=DEV, requires_grad=True)
    l2 = compute_sdf_reg_loss(s2, edges)
    l2.backward()
    assert float(l2) == 0.0 and float(s2.grad.abs().max()) == 0.0


@pytest.mark.parametrize("n_hidden,skip_in", [(6, [3]), (2, []), (3, [0])])
def test_fused_sdf_mlp_forward(n_hidden, skip_in):
    """Fused MFMA forward vs the plain torch module on the same device and on CPU (fp32)."""
    from gshell_amd.geometry.mlp import MLP, forward_row_sparse_backward, fused_forward
    torch.manual_seed(0)
    net = MLP(skip_in=skip_in, n_freq=6, n_hidden=n_hidden, d_hidden=256)
    with torch.no_grad():
        for p in net.parameters():          # non-trivial biases
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.05)
    x = torch.rand(1000 + 37, 3) * 1.4 - 0.7
    import copy
    ref = net(x).detach()
    ref64 = copy.deepcopy(net).double()(x.double()).detach()[:, 0]      # before .to(DEV), which moves `net` itself
    netd = net.to(DEV)
    xd = x.to(DEV)
    err_torch = float((ref[:, 0].double() - ref64).abs().max())          # what ANY fp32 evaluation order costs (torch CPU SGEMM)
    from gshell_amd import _lib
    for precision in ("fp32", "h2"):
        if precision == "fp32":      # the exact-fp32 MFMA kernel of round 1: an oracle kernel (lib/variants/oracles.so); the shipped library refuses it
            with pytest.raises(_lib.GShellHipError, match="oracle"):
                fused_forward(netd, xd, precision)
            with _lib.use_variant("oracles"):
                out = fused_forward(netd, xd, precision)
        else:
            out = fused_forward(netd, xd, precision)
        assert out.shape == (x.shape[0], 1)
        # fp32: k-ordered fma chain on the matrix core.  h2: fp16-pair operands (2^-22), three MFMAs per product, fp32
        # accumulate.  Both must sit at fp32 round-off of a 256-term dot product chain -- measured against float64 and
        # against the error of torch's own fp32 evaluation of the same network.
        err = float((out.cpu()[:, 0].double() - ref64).abs().max())
        assert err <= 2e-6 * float(ref64.abs().max()) + 4.0 * err_torch, (precision, err, err_torch)
        assert float((out.cpu() - ref).abs().max() / ref.abs().max()) < 2e-5
    assert torch.equal(fused_forward(netd, xd), fused_forward(netd, xd, "h2"))       # h2 is the default path
    xg = xd.clone().requires_grad_(True)
    y = forward_row_sparse_backward(netd, xg)
    g = torch.zeros_like(y)
    g[::7] = 1.0
    y.backward(g)
    xr = x.clone().requires_grad_(True)
    net.cpu()(xr).backward(g.cpu())
    assert (xg.grad.cpu() - xr.grad).abs().max() <= 1e-4 * xr.grad.abs().max()


@pytest.mark.parametrize("with_light", [True, False])
def test_frame_sums_match_the_torch_terms(with_light):
    """gs_frame_sums_fwd/bwd vs the term-by-term torch formulation (regularizer.shading_loss / material_smoothness_grad, the
    alpha MSE and the two mSDF image terms of tick): sums to 1e-5 relative, gradients w.r.t. every buffer to 1e-4 relative."""
    import torch.nn.functional as F
    from gshell_amd.render import regularizer as R
    g = torch.Generator(device="cuda").manual_seed(3 + with_light)
    B, H, W = 2, 37, 41
    keys = ['shaded', 'z_grad', 'kd', 'kd_grad', 'ks_grad', 'normal_grad'] + (['diffuse_light', 'specular_light'] if with_light else []) + ['msdf_image']
    sizes = [4, 4, 4, 4, 4, 4] + ([4, 4] if with_light else []) + [1]
    stacked = torch.rand(B, H, W, sum(sizes), device="cuda", generator=g)
    o = sum(sizes[:-1])
    stacked[..., o] = stacked[..., o] * 2 - 1                           # mSDF image takes both signs
    stacked[0, 0, :5, o] = 0.0                                          # ... and exact zeros (clamp / abs kinks)
    if with_light:
        od = sum(sizes[:keys.index('diffuse_light')])
        stacked[0, 1, :7, od:od + 8] = 0.0                              # unlit pixels: logsrgb at its knee
        stacked[0, 2, :7, od:od + 3] *= 30.0
    stacked.requires_grad_(True)
    ref = torch.rand(B, H, W, 4, device="cuda", generator=g)
    ref[..., 3] = (ref[..., 3] > 0.5).float()                           # alpha is exactly 0 or 1 on most pixels
    ref[1, 3, :9, 3] = 0.37                                             # ... and fractional on antialiased edges
    w = torch.rand(9, device="cuda", generator=g) + 0.5
    fs = R.frame_sums((stacked, keys, sizes), ref)
    (fs * w).sum().backward()
    got_g = stacked.grad.clone()
    stacked.grad = None
    buf = dict(zip(keys, torch.split(stacked, sizes, dim=-1)))
    m = ref[..., 3:]
    n = float(m.numel())
    zero = torch.zeros((), device="cuda")
    want = [F.mse_loss(buf['shaded'][..., 3:], m) * n,
            F.l1_loss(buf['msdf_image'].clamp(min=0) * (m == 0).float(), torch.zeros_like(m)) * n,
            F.l1_loss(buf['msdf_image'].clamp(max=0) * (m == 1).float(), torch.ones_like(m)) * n]
    if with_light:
        d, s = R.luma(buf['diffuse_light']), R.luma(buf['specular_light'])
        want += [(R._log_srgb((d + s) * m) - R._log_srgb(R.value(ref) * m)).abs().mean() * n, s.mean() * n, d.mean() * n]
    else:
        want += [zero, zero, zero]
    want += [(buf['kd_grad'][..., :3].sum(-1) / 3 * buf['kd_grad'][..., -1]).sum(),
             (buf['ks_grad'][..., :-1] * buf['ks_grad'][..., -1:]).sum(), (buf['normal_grad'][..., :-1] * buf['normal_grad'][..., -1:]).sum()]
    want = torch.stack([x.reshape(()) for x in want])
    assert torch.allclose(fs, want, rtol=1e-5, atol=1e-4), (fs, want)
    (want * w).sum().backward()
    scale = float(stacked.grad.abs().max())
    assert float((got_g - stacked.grad).abs().max()) <= 1e-4 * scale


@pytest.mark.parametrize("loss,tm", [("l1", "log_srgb"), ("mse", "log_srgb"), ("smape", "none"), ("relmse", "none"), ("mse", "none")])
def test_frame_sums_colour_term_equals_image_loss(loss, tm):
    """gs_frame_sums_img_*: the tenth sum is the element sum of renderutils.image_loss(shaded rgb * m, reference rgb * m) (pinned to
    the reference's loss twins in test_image_loss_*), the other nine sums are unchanged, and the gradient of the frame is the sum of
    the two separate gradients."""
    from gshell_amd.render import regularizer as R, renderutils as ru
    g = torch.Generator(device="cuda").manual_seed(11)
    B, H, W = 2, 29, 35
    keys, sizes = ['shaded', 'kd_grad', 'ks_grad', 'normal_grad', 'diffuse_light', 'specular_light', 'msdf_image'], [4, 4, 4, 4, 4, 4, 1]
    stacked = (torch.rand(B, H, W, sum(sizes), device="cuda", generator=g) * 1.5).requires_grad_(True)
    with torch.no_grad():
        stacked[0, 0, :6, 0:3] = 0.0                                     # tonemapper knee / clamp edge
        stacked[0, 1, :6, 0:3] *= 40.0
    ref = torch.rand(B, H, W, 4, device="cuda", generator=g)
    ref[..., 3] = (ref[..., 3] > 0.4).float()
    ref[1, 2, :9, 3] = 0.41
    w = torch.rand(10, device="cuda", generator=g) + 0.5
    spec = (ru._LOSS[loss], ru._TONEMAP[tm])
    fs10 = R.frame_sums((stacked, keys, sizes), ref, spec)
    (fs10 * w).sum().backward()
    got = stacked.grad.clone()
    stacked.grad = None
    fs9 = R.frame_sums((stacked, keys, sizes), ref)
    m = ref[..., 3:]
    img = ru.image_loss(stacked[..., 0:3] * m, ref[..., 0:3] * m, loss=loss, tonemapper=tm) * (3.0 * m.numel())
    assert torch.equal(fs10[:9], fs9)
    assert abs(float(fs10[9]) - float(img)) <= 1e-5 * abs(float(img)) + 1e-6
    ((fs9 * w[:9]).sum() + img * w[9]).backward()
    assert float((got - stacked.grad).abs().max()) <= 1e-5 * float(stacked.grad.abs().max())


def test_sdf_net_torch_formulation_fast_paths_match_plain_modules():
    """geometry/mlp.py swaps in a split-K Linear and HIP softplus kernels (value / gradient / gradient of the gradient) for
    large row counts.  Against the plain nn.Linear / nn.Softplus modules: outputs 1e-6, eikonal-style double-backward
    gradients 1e-4 relative (fp32, different summation order)."""
    import gshell_amd.geometry.mlp as M
    torch.manual_seed(0)
    net = M.MLP(n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3]).cuda()
    x = (torch.rand(70001, 3, device="cuda") - 0.5).requires_grad_(True)

    def run():
        y = net(x)
        g = torch.autograd.grad(y.sum(), x, create_graph=True)[0]
        loss = ((g.pow(2).sum(-1).sqrt() - 1) ** 2).mean() + y.pow(2).mean()
        return y.detach(), g.detach(), torch.autograd.grad(loss, list(net.parameters()))
    y1, g1, p1 = run()
    saved = M._linear, M._softplus
    M._linear, M._softplus = (lambda m, h: m(h)), (lambda m, h: m(h))
    try:
        y0, g0, p0 = run()
    finally:
        M._linear, M._softplus = saved
    assert torch.allclose(y1, y0, rtol=1e-5, atol=1e-6)
    assert float((g1 - g0).abs().max()) <= 1e-4 * float(g0.abs().max())
    for a, b in zip(p1, p0):
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-9


@pytest.mark.parametrize("N,nb,T", [(5000, 700, 1500), (300001, 40000, 90000), (64, 0, 10)])
def test_msdf_regularisers_match_the_reference_formulation(N, nb, T):
    """gs_boundary_weight + gs_msdf_reg_fwd/bwd vs the reference's operator sequence (geometry/gshell_tets_geometry.py:326-358:
    clamp -> huber_loss(reduction='sum') against -eps / +eps, the close term over the boundary vertices of visible triangles).
    Values 2e-6 relative (fp32 sums in a different order), gradients 1e-6 (fp32 v + eps against the float64 reference)."""
    import torch.nn.functional as F
    from gshell_amd.geometry.gshell_tets_geometry import _MsdfRegFn, boundary_weight, visible_boundary_weight
    g = torch.Generator().manual_seed(N)
    msdf = (torch.randn(N, generator=g) * 0.8)
    msdf[:8] = torch.tensor([-1e-3, 1e-3, 0.0, -2.0, 2.0, -0.9995, 0.9995, 1.5])
    nwt = 1000
    bnd = torch.randn(nb, 1, generator=g) * 0.7
    tri = torch.randint(0, nwt + max(nb, 1), (T, 3), generator=g).int()
    flags = (torch.rand(T, generator=g) > 0.6).to(torch.uint8)
    open_w, close_w, eps = 0.37, 1.3, 1e-3
    # reference formulation, float64 on the CPU
    m64 = msdf.double().requires_grad_(True)
    b64 = bnd.double().requires_grad_(True)
    vis = torch.zeros(nb, dtype=torch.bool)
    vt = tri[flags.bool()].reshape(-1).long() - nwt
    vt = vt[(vt >= 0) & (vt < nb)]
    vis[vt] = True
    e = torch.full((1,), eps, dtype=torch.float64)
    want_open = open_w * F.huber_loss(m64.clamp(min=-e), -e.expand(N), reduction='sum')
    bm = b64.reshape(-1)[vis]
    want_close = close_w * F.huber_loss(bm.clamp(max=e), e.expand(bm.numel()), reduction='sum') if bm.numel() else torch.zeros((), dtype=torch.float64)
    (2.0 * want_open + 3.0 * want_close).backward()
    # HIP
    md, bd = msdf.to(DEV).requires_grad_(True), bnd.to(DEV).requires_grad_(True)
    w = boundary_weight(tri.to(DEV), flags.to(DEV), nwt, nb)
    assert torch.equal(w.cpu() > 0, vis)
    if nb:
        assert torch.equal(w, visible_boundary_weight(tri.to(DEV).long(), flags.to(DEV).int(), nwt, nb))
    two = _MsdfRegFn.apply(md, bd, w, eps, open_w, close_w)
    (2.0 * two[0] + 3.0 * two[1]).backward()
    assert abs(float(two[0]) - float(want_open)) <= 2e-6 * abs(float(want_open)) + 1e-12
    assert abs(float(two[1]) - float(want_close)) <= 2e-6 * abs(float(want_close)) + 1e-12
    assert torch.allclose(md.grad.cpu().double(), m64.grad, rtol=1e-6, atol=1e-9)
    if nb:
        assert torch.allclose(bd.grad.cpu().double(), b64.grad, rtol=1e-6, atol=1e-9)


def test_surface_sampling_kernels_match_the_torch_formulation():
    """gs_tri_area / gs_surface_points_cdf vs torch expressions: areas 1e-6 relative (degenerate and non-finite triangles included),
    face ids = the inversion of the area CDF at the drawn numbers, points 1e-6; the draw consumes the generator deterministically
    (ranks of a view-sharded job agree) and faces come out in proportion to their area."""
    from gshell_amd import _lib
    from gshell_amd._lib import c_int64, check, ptr, stream
    from gshell_amd.geometry.gshell_tets_geometry import sample_points, sample_points_detached
    g = torch.Generator(device="cuda").manual_seed(21)
    V, T, n = 5000, 9000, 20000
    v = torch.randn(V, 3, device="cuda", generator=g)
    tri = torch.randint(0, V, (T, 3), device="cuda", generator=g).int()
    tri[5] = tri[5, 0]                                    # degenerate
    v[tri[7, 1].long()] = float("inf")                    # non-finite area -> weight 1e-20
    vf = v.clone()
    v0, v1, v2 = vf[tri[:, 0].long()], vf[tri[:, 1].long()], vf[tri[:, 2].long()]
    area_ref = torch.linalg.cross(v1 - v0, v2 - v0).norm(dim=-1)
    area_ref = torch.where(torch.isfinite(area_ref), area_ref, torch.zeros_like(area_ref)) + 1e-20
    area = torch.empty(T, device="cuda")
    check(_lib.lib().gs_tri_area(ptr(vf), ptr(tri), c_int64(T), ptr(area), stream()), "gs_tri_area")
    assert torch.allclose(area, area_ref, rtol=2e-6, atol=0)
    finite = torch.isfinite(vf).all(dim=1)
    ok_tri = finite[tri.long()].all(dim=1)
    tri_ok = tri[ok_tri].contiguous()
    gen_a, gen_b = torch.Generator(device="cuda").manual_seed(5), torch.Generator(device="cuda").manual_seed(5)
    pts_a, fid_a = sample_points_detached(vf, tri_ok, n, generator=gen_a)
    # the product draws r = rand(n, 3) = (u, v, face) from the generator and inverts the area CDF in the kernel: replay it with torch ops
    r = torch.rand(n, 3, device="cuda", generator=gen_b)
    a0, a1, a2 = vf[tri_ok[:, 0].long()], vf[tri_ok[:, 1].long()], vf[tri_ok[:, 2].long()]
    area_ok = torch.linalg.cross(a1 - a0, a2 - a0).norm(dim=-1) + 1e-20
    cdf = torch.cumsum(area_ok, dim=0)
    fid_b = torch.searchsorted(cdf, r[:, 2] * cdf[-1], right=True).clamp(max=tri_ok.shape[0] - 1)
    same = fid_a == fid_b                                  # an ulp in an area / in the scan can move a draw to the neighbouring face
    assert float(same.float().mean()) > 0.999
    u, w = r[:, 0:1].sqrt(), r[:, 1:2]
    pts_b = (1 - u) * a0[fid_b] + u * (1 - w) * a1[fid_b] + u * w * a2[fid_b]
    assert torch.allclose(pts_a[same], pts_b[same], rtol=1e-5, atol=1e-6)
    assert torch.equal(torch.rand(3, device="cuda", generator=gen_a), torch.rand(3, device="cuda", generator=gen_b))    # same generator consumption
    # faces are drawn in proportion to their area (kaolin.ops.mesh.sample_points' contract): 16 groups of faces, 5 sigma
    grp = torch.arange(tri_ok.shape[0], device="cuda") * 16 // tri_ok.shape[0]
    share = torch.zeros(16, device="cuda").index_add_(0, grp, area_ok) / area_ok.sum()
    freq = torch.zeros(16, device="cuda").index_add_(0, grp[fid_a], torch.ones(n, device="cuda")) / n
    assert float(((freq - share).abs() / (share * (1 - share) / n).sqrt()).max()) < 5.0
    # the autograd formulation used outside the training path agrees in distribution as well (same contract, torch.multinomial)
    pts_c, fid_c = sample_points(vf, tri_ok.long(), n, generator=gen_b)
    freq_c = torch.zeros(16, device="cuda").index_add_(0, grp[fid_c], torch.ones(n, device="cuda")) / n
    assert float(((freq_c - share).abs() / (share * (1 - share) / n).sqrt()).max()) < 5.0


def test_depth_zgrad_equals_the_reference_expression_bit_for_bit():
    """ru.depth_zgrad (one launch) against the reference's operator chain (render/render.py:276-279) on random values, values at and
    below eps, negative w, infinities and NaN: bit-identical."""
    from gshell_amd.render import renderutils as ru
    g = torch.Generator().manual_seed(21)
    n = 5000
    c = torch.randn(2, 50, 50, 4, generator=g)
    d = torch.randn(2, 50, 50, 8, generator=g) * 0.01
    flat_c, flat_d = c.view(-1, 4), d.view(-1, 8)
    flat_c[0] = torch.tensor([0.0, 0.0, 1e-6, 1e-6])
    flat_c[1] = torch.tensor([0.0, 0.0, 0.0, 0.0])
    flat_c[2] = torch.tensor([0.0, 0.0, float("nan"), 1.0])
    flat_c[3] = torch.tensor([0.0, 0.0, 1.0, float("nan")])
    flat_c[4] = torch.tensor([0.0, 0.0, float("inf"), 2.0])
    flat_c[5] = torch.tensor([0.0, 0.0, 0.5, -3.0])
    flat_d[6, 2:4] = torch.tensor([float("inf"), float("nan")])
    c, d = c.to(DEV), d.to(DEV)
    eps = 0.00001
    z0 = torch.clamp(c[..., 2:3], min=eps) / torch.clamp(c[..., 3:4], min=eps)
    z1 = torch.clamp(c[..., 2:3] + torch.abs(d[..., 2:3]), min=eps) / torch.clamp(c[..., 3:4] + torch.abs(d[..., 3:4]), min=eps)
    ref = torch.cat((z0, torch.abs(z1 - z0)), dim=-1)
    out = ru.depth_zgrad(c, d, eps)
    assert out.shape == ref.shape
    same = (out.view(torch.int32) == ref.view(torch.int32)) | (torch.isnan(out) & torch.isnan(ref))
    assert bool(same.all()), int((~same).sum())

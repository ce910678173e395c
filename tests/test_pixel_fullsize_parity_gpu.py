"""Pixel stages (R2 rasterise, R3 interpolate, R4 bilinear taps, R5 hash grid, R7 antialias) against the oracle AT A BASELINE CONFIG SIZE:
512 x 512 frames of the mesh the tet-res128 / tet-res256 workload extracts (4 10^4 / 2.3 10^5 triangles, micro triangles of ~1 pixel, depth
ties along shared edges), one view from the bench orbit and one from INSIDE the garment, where thousands of triangles straddle the eye
plane (the near-clip / workgroup-per-triangle path `k_rast_large`).

The oracle side: oracle/raster_c.c (the C restatement of raster_oracle.rasterize_ids, pinned to the python loop bit for bit by
tests/test_raster_oracle.py) for the integer decisions; raster_oracle / pixel_oracle / hashgrid_oracle torch float32 on the CPU for
everything differentiable.  PARITY UNPINNED for the nvdiffrast / tiny-cuda-nn semantics themselves (sources absent: see the oracle headers).

Bar: ids bit-exact; every float buffer within 1e-4 (relative to the buffer's scale) with AT MOST 2 pixels per buffer outside, each outlier
printed with its cause; gradients 1e-4 relative to their maximum."""
import numpy as np
import pytest
import torch

from oracle import hashgrid_oracle as ho
from oracle import pixel_oracle as po
from oracle import raster_oracle as ro

pytestmark = pytest.mark.gpu
DEV = "cuda"
H = W = 512


def _outside(a, b, tol=1e-4):
    """per-pixel bool: some channel differs by more than tol * max(|b|) + tol * |b|."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = float(b.abs().max()) or 1.0
    dev = (a - b).abs() - tol * b.abs()
    return (dev > tol * scale).reshape(*dev.shape[:3], -1).any(-1), scale


@pytest.fixture(scope="module", params=[128, 256])
def scene(request):
    from gshell_amd import workload
    from gshell_amd.render import renderutils as ru
    res = request.param
    torch.manual_seed(0)
    tr = workload.build(res=res, n_samples=2, batch=1, train_res=(H, W), fit_steps=150)
    with torch.no_grad():
        d = tr.geometry.getMesh(tr.mat)
    m = d['imesh']
    mvp0, _ = workload.views([0], DEV)                      # the bench orbit (radius 2.2)
    mvp1, _ = workload.views([5], DEV, radius=0.22)         # inside the skirt: the surface passes through the eye plane
    mvp = torch.cat([mvp0, mvp1])
    clip = ru.xfm_points(m.v_pos[None], mvp)
    tri = m.faces_i32().contiguous()
    clip_c, tri_c = clip.cpu(), tri.cpu().long()
    ids_ref = torch.tensor(ro.rasterize_ids_c(clip_c.numpy(), tri.cpu().numpy(), H, W))
    return dict(res=res, tr=tr, d=d, mesh=m, mvp=mvp, clip=clip, tri=tri, clip_c=clip_c, tri_c=tri_c, ids_ref=ids_ref)


def test_xfm_rasterise_ids_bit_exact_and_rast_within_1e4(scene):
    from gshell_amd.render import rast as dr
    s = scene
    T = s['tri'].shape[0]
    assert T > (30000 if s['res'] == 128 else 150000)
    # R1 at this size: the clip-space positions themselves
    ref_clip = ro.xfm_points(s['mesh'].v_pos.detach().cpu()[None], s['mvp'].cpu())
    assert float((s['clip_c'] - ref_clip).abs().max()) <= 1e-5 * float(ref_clip.abs().max())
    # ... and bit for bit the kernel-order restatement the end-to-end oracle chains rasterise from (mesh.cu:43-46 without contraction)
    assert torch.equal(s['clip_c'], ro.xfm_points_kernel_order(s['mesh'].v_pos.detach().cpu()[None], s['mvp'].cpu()))
    front = s['clip_c'][:, s['tri_c'].reshape(-1), 3].reshape(2, -1, 3) > 1e-6
    n_clip = (front.any(-1) & ~front.all(-1)).sum(-1)
    assert int(n_clip[0]) == 0 and int(n_clip[1]) > 200, n_clip     # view 1 exercises the near-clip path on hundreds of triangles
    rast, db, vis = dr.rasterize(None, s['clip'], s['tri'], (H, W), return_visible=True)
    ids = (rast[..., 3].long() - 1).cpu()
    ids_ref = s['ids_ref']
    cov = (ids_ref >= 0).float().mean(dim=(1, 2))
    print(f"\n  tet-res{s['res']}: T={T}, coverage per view {cov.tolist()}, near-clipped triangles {n_clip.tolist()}, "
          f"visible triangles {int((torch.bincount(ids_ref[ids_ref >= 0], minlength=T) > 0).sum())}")
    assert float(cov[0]) > 0.05 and float(cov[1]) > 0.3
    assert torch.equal(ids, ids_ref), f"{int((ids != ids_ref).sum())} pixels carry a different triangle id"
    want = torch.zeros(T, dtype=torch.uint8)
    want[torch.unique(ids_ref[ids_ref >= 0])] = 1
    assert torch.equal(vis.cpu(), want)
    rast_ref, db_ref = ro.rast_from_ids(s['clip_c'], s['tri_c'], ids_ref)
    for name, a, b in (("rast.uv", rast[..., :2], rast_ref[..., :2]), ("rast.z/w", rast[..., 2:3], rast_ref[..., 2:3]), ("rast_db", db, db_ref)):
        bad, scale = _outside(a, b)
        print(f"  {name}: scale {scale:.3g}, pixels outside 1e-4: {int(bad.sum())} of {bad.numel()}")
        if name == "rast_db":
            # d(u,v)/d(pixel) of a triangle seen edge-on grows without bound; the float32 oracle and kernel agree to 1e-4 of the VALUE there
            rel = ((a.cpu() - b).abs() <= 1e-3 * b.abs() + 1e-4)
            bad = ~rel.all(-1)
            print(f"  rast_db: pixels outside rtol 1e-3 + 1e-4: {int(bad.sum())}")
        for (bb, y, x) in torch.nonzero(bad)[:4].tolist():
            print(f"    outlier view {bb} pixel ({y},{x}) tri {int(ids_ref[bb, y, x])}: {a[bb, y, x].tolist()} vs {b[bb, y, x].tolist()}")
        assert int(bad.sum()) <= 2, name
    s['rast'], s['rast_db'], s['rast_ref'], s['db_ref'] = rast, db, rast_ref, db_ref


def test_rasterise_backward_at_512(scene):
    from gshell_amd.render import rast as dr
    s = scene
    wgt = torch.rand(2, H, W, 2, generator=torch.Generator().manual_seed(1))
    p_ref = s['clip_c'].clone().requires_grad_(True)
    rast_ref, _ = ro.rast_from_ids(p_ref, s['tri_c'], s['ids_ref'])
    # view 1's near-clipped triangles have vertices at w ~ 0: their barycentric gradients are unbounded, weight the regular view only
    wgt[1] *= (s['clip_c'][1, s['tri_c'][s['ids_ref'][1].clamp(min=0)], 3].min(-1).values > 0.05)[..., None].float()
    (rast_ref[..., :2] * wgt).sum().backward()
    p = s['clip'].detach().clone().requires_grad_(True)
    rast, _ = dr.rasterize(None, p, s['tri'], (H, W))
    (rast[..., :2] * wgt.to(DEV)).sum().backward()
    g, g_ref = p.grad.cpu(), p_ref.grad
    err = float((g - g_ref).abs().max() / g_ref.abs().max())
    print(f"\n  rasterise backward, tet-res{s['res']}: max error / max gradient = {err:.2e}")
    assert err <= 1e-4


def test_interpolate_groups_and_derivatives_at_512(scene):
    from gshell_amd.render import rast as dr
    s = scene
    m, d = s['mesh'], s['d']
    rast_ref, db_ref = ro.rast_from_ids(s['clip_c'], s['tri_c'], s['ids_ref'])
    rast_d, db_d = rast_ref.to(DEV), db_ref.to(DEV)
    attrs = [m.v_pos.detach(), m.v_nrm.detach(), d['msdf'].detach().reshape(-1, 1)]
    gen = torch.Generator().manual_seed(2)
    wg = [torch.rand(2, H, W, a.shape[1], generator=gen) for a in attrs]
    a_d = [a.clone().requires_grad_(True) for a in attrs]
    r_d = rast_d.clone().requires_grad_(True)
    outs = dr.interpolate_groups(a_d, r_d, s['tri'])
    sum((o * w.to(DEV)).sum() for o, w in zip(outs, wg)).backward()
    a_c = [a.cpu().clone().requires_grad_(True) for a in attrs]
    r_c = rast_ref.clone().requires_grad_(True)
    refs = [ro.interpolate(a[None], r_c, s['tri_c']) for a in a_c]
    sum((o * w).sum() for o, w in zip(refs, wg)).backward()
    for name, o, r in zip(("gb_pos", "gb_normal", "msdf_image"), outs, refs):
        bad, scale = _outside(o, r)
        print(f"\n  {name}: scale {scale:.3g}, pixels outside 1e-4: {int(bad.sum())}")
        assert int(bad.sum()) <= 2, name
    for name, x, y in zip(("d/d v_pos", "d/d v_nrm", "d/d msdf"), a_d, a_c):
        err = float((x.grad.cpu() - y.grad).abs().max() / y.grad.abs().max())
        print(f"  {name}: {err:.2e}")
        assert err <= 1e-4, name
    err = float((r_d.grad.cpu()[..., :2] - r_c.grad[..., :2]).abs().max() / r_c.grad[..., :2].abs().max())
    assert err <= 1e-4, err
    # the depth guide's input: clip-space position with screen-space derivatives (render.py:262-264)
    out, da = dr.interpolate(s['clip'].detach(), rast_d, s['tri'], rast_db=db_d, diff_attrs='all')
    out_ref, da_ref = ro.interpolate(s['clip_c'], rast_ref, s['tri_c'], db_ref)
    bad, _ = _outside(out, out_ref)
    assert int(bad.sum()) <= 2
    front = (s['clip_c'][:, :, 3][torch.arange(2)[:, None, None], s['tri_c'][s['ids_ref'].clamp(min=0)].amin(-1)] > 0.05)
    ok = ((da.cpu() - da_ref).abs() <= 1e-3 * da_ref.abs() + 1e-4 * float(da_ref[front].abs().max())).all(-1)
    print(f"  clip-space derivatives: pixels outside: {int((~ok & front).sum())}")
    assert int((~ok & front).sum()) <= 2


def test_antialias_at_512(scene):
    from gshell_amd.render import rast as dr
    s = scene
    rast_ref, _ = ro.rast_from_ids(s['clip_c'], s['tri_c'], s['ids_ref'])
    rast_d = rast_ref.to(DEV)
    V = s['clip'].shape[1]
    opp_ref = ro.tri_adjacency_sorted(s['tri_c'].numpy())
    topo = dr.AATopology(s['tri'], V)
    assert np.array_equal(topo.opp.cpu().numpy(), opp_ref), "triangle adjacency differs"
    C = 6
    gen = torch.Generator().manual_seed(3)
    color = torch.rand(2, H, W, C, generator=gen) * (s['ids_ref'] >= 0)[..., None].float() + 0.1 * torch.rand(2, H, W, C, generator=gen)
    wgt = torch.rand(2, H, W, C, generator=gen)
    # VIEW 0 only for the position gradient (see test_rasterise_backward_at_512); both views for the values
    wgt_p = wgt.clone()
    wgt_p[1] = 0
    p_ref, c_ref = s['clip_c'].clone().requires_grad_(True), color.clone().requires_grad_(True)
    alpha_ref = ro.aa_alpha(rast_ref, p_ref, s['tri_c'], torch.as_tensor(opp_ref))
    out_ref = ro.aa_apply(c_ref, alpha_ref)
    alpha = dr.aa_analyze(rast_d, s['clip'], s['tri'], topo).cpu()
    ar = alpha_ref.detach()
    n_sil = int((ar != 0).sum())
    mism = (alpha - ar).abs() > 1e-4
    print(f"\n  tet-res{s['res']}: silhouette pairs {n_sil}, pairs whose blend factor differs by more than 1e-4: {int(mism.sum())}")
    for (bb, y, x, k) in torch.nonzero(mism)[:6].tolist():
        print(f"    view {bb} pixel ({y},{x}) pair {'right' if k == 0 else 'down'}: {float(alpha[bb, y, x, k]):.6f} vs {float(ar[bb, y, x, k]):.6f}")
    assert n_sil > 500
    assert int(mism.sum()) <= 2
    p, c = s['clip'].detach().clone().requires_grad_(True), color.to(DEV).requires_grad_(True)
    out = dr.antialias(c, rast_d, p, s['tri'])
    bad, _ = _outside(out, out_ref)
    print(f"  antialiased frame: pixels outside 1e-4: {int(bad.sum())}")
    assert int(bad.sum()) <= 2 + 2 * int(mism.sum())
    (out * wgt_p.to(DEV)).sum().backward()
    (out_ref * wgt_p).sum().backward()
    if not mism.any():
        e_c = float((c.grad.cpu() - c_ref.grad).abs().max() / c_ref.grad.abs().max())
        e_p = float((p.grad.cpu() - p_ref.grad).abs().max() / p_ref.grad.abs().max())
        print(f"  antialias gradients: colour {e_c:.2e}, position {e_p:.2e}")
        assert e_c <= 1e-4 and e_p <= 2e-4
    # the in-place form the training path uses
    frame = color.to(DEV).clone()
    out2 = dr.antialias_stacked([frame], rast_d, s['clip'].detach(), s['tri'], topo, inplace=True)[0]
    assert torch.equal(out2, out.detach())


def test_bilinear_taps_at_512(scene):
    from gshell_amd.render import rast as dr
    s = scene
    gen = torch.Generator().manual_seed(4)
    mask = (s['ids_ref'] >= 0)[..., None].float()
    gb = torch.rand(2, H, W, 3, generator=gen) * mask
    jitter = po.pixel_grid(W, H)[None] + torch.randn(2, H, W, 2, generator=gen) * 0.005
    t_ref = gb.clone().requires_grad_(True)
    ref = po.texture_linear_clamp(t_ref, jitter)
    wgt = torch.rand(2, H, W, 3, generator=gen)
    (ref * wgt).sum().backward()
    t = gb.to(DEV).requires_grad_(True)
    out = dr.texture(t, jitter.to(DEV), filter_mode='linear', boundary_mode='clamp')
    (out * wgt.to(DEV)).sum().backward()
    bad, _ = _outside(out, ref)
    assert int(bad.sum()) == 0
    assert float((t.grad.cpu() - t_ref.grad).abs().max()) <= 1e-4 * float(t_ref.grad.abs().max())
    m_ref = po.texture_linear_clamp(mask, jitter)
    m_out = dr.texture(mask.to(DEV), jitter.to(DEV), filter_mode='linear', boundary_mode='clamp')
    assert float((m_out.cpu() - m_ref).abs().max()) <= 1e-4        # u * 512 - 0.5 carries 3e-5 of float32 round-off at this frame size


def test_hash_grid_and_texture_field_on_the_covered_pixels_at_512(scene):
    """R5 on the real g-buffer: every covered pixel of both views (1.5 .. 3 10^5 surface points, clustered on a thin surface: the access
    pattern the binned table gradient was built for), reference configuration (16 levels, 2 features, 2^19 entries, 16 -> 4096)."""
    from gshell_amd.render.mlptexture import HashGridEncoding, _HashGridFn
    s = scene
    rast_ref, _ = ro.rast_from_ids(s['clip_c'], s['tri_c'], s['ids_ref'])
    gb_pos = ro.interpolate(s['mesh'].v_pos.detach().cpu()[None], rast_ref, s['tri_c'])
    pts = gb_pos[s['ids_ref'] >= 0]
    lo, hi = pts.min(0).values - 0.01, pts.max(0).values + 0.01
    x = ((pts - lo) / (hi - lo)).clamp(0, 1)
    cfg = (16, 2, 19, 16, float(np.exp(np.log(4096 / 16) / 15)))
    _, total = ho.level_meta(*cfg)
    gen = torch.Generator().manual_seed(5)
    params = torch.rand(total, generator=gen) - 0.5
    w = torch.randn(x.shape[0], 32, generator=gen)
    x_ref, p_ref = x.clone().requires_grad_(True), params.clone().requires_grad_(True)
    out_ref = ho.encode(x_ref, p_ref, *cfg)
    (out_ref * w).sum().backward()
    enc = HashGridEncoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16,
                               "per_level_scale": cfg[4]})
    xd, pd = x.to(DEV).requires_grad_(True), params.to(DEV).requires_grad_(True)
    out = _HashGridFn.apply(xd, pd, None, enc.cfg)
    (out * w.to(DEV)).sum().backward()
    d = (out.cpu() - out_ref.detach()).abs()
    bad = (d > 1e-4 * out_ref.detach().abs() + 1e-5).any(-1)
    # a point within float32 round-off of a cell face of a fine level lands in either cell: the encoding is continuous there, so the VALUE
    # still agrees -- no quota needed for the forward
    print(f"\n  tet-res{s['res']}: {x.shape[0]} surface points; rows outside 1e-4: {int(bad.sum())}; max abs err {float(d.max()):.2e}")
    assert int(bad.sum()) == 0
    e_p = float((pd.grad.cpu() - p_ref.grad).abs().max() / p_ref.grad.abs().max())
    print(f"  table gradient max err / max: {e_p:.2e}; position gradient rel L2: "
          f"{float((xd.grad.cpu() - x_ref.grad).norm() / x_ref.grad.norm()):.2e}")
    assert e_p <= 1e-5
    assert torch.equal(pd.grad.cpu() != 0, p_ref.grad != 0) or float(((pd.grad.cpu() != 0) != (p_ref.grad != 0)).float().sum()) <= 16
    # d/dx jumps at cell faces (piecewise-trilinear): a point ON a face may take either slope -- count and bound them
    gx, gr = xd.grad.cpu(), x_ref.grad
    badx = ((gx - gr).abs() > 1e-4 * gr.abs() + 1e-5 * float(gr.abs().max())).any(-1)
    print(f"  position-gradient rows outside: {int(badx.sum())} of {x.shape[0]} (points on a cell face of some level)")
    assert int(badx.sum()) <= max(4, x.shape[0] // 20000)


def test_fused_texture_field_training_path_against_the_oracle_at_512(scene):
    """the FUSED texture field of the training path -- MLPTexture3D.sample_many: AABB normalisation + clamp + level-major
    encoding of both coordinate sets in one pass, the texture MLP on the matrix cores, and in the backward the BINNED table gradient
    (per-bin records summed in 64-bit fixed point) -- directly against oracle/hashgrid_oracle + a plain torch MLP (pipeline_oracle.TextureOracle),
    not against the product's own operator-by-operator path.  Real g-buffer of the 512^2 frames, masked to the covered pixels, with the
    reference's two gradient hooks (mlptexture.py:72-77: x128 on the table, /128 x128 = 1 on the position)."""
    from gshell_amd.render.mlptexture import MLPTexture3D
    from oracle import pipeline_oracle as pl
    s = scene
    rast_ref, _ = ro.rast_from_ids(s['clip_c'], s['tri_c'], s['ids_ref'])
    gb_pos = ro.interpolate(s['mesh'].v_pos.detach().cpu()[None], rast_ref, s['tri_c'])
    mask = (s['ids_ref'] >= 0).float()
    gen = torch.Generator().manual_seed(6)
    noise = torch.randn(2, H, W, 3, generator=gen) * 0.01
    lo, hi = gb_pos[mask > 0].min(0).values - 0.02, gb_pos[mask > 0].max(0).values + 0.02
    aabb = (lo.to(DEV), hi.to(DEV))
    mn = torch.tensor([0, 0, 0, 0, 0.08, 0], dtype=torch.float32, device=DEV)
    mx = torch.tensor([1, 1, 1, 0.3, 1, 1], dtype=torch.float32, device=DEV)
    torch.manual_seed(7)
    tex = MLPTexture3D(aabb, channels=6, min_max=[mn, mx])
    with torch.no_grad():
        tex.encoder.params.mul_(3000.0)
    go = torch.randn(2, 2, H, W, 6, generator=gen) * mask[None, ..., None]
    p_d = gb_pos.to(DEV).requires_grad_(True)
    a, b = tex.sample_many([p_d + noise.to(DEV), p_d], mask.to(DEV))
    ((a * go[0].to(DEV)).sum() + (b * go[1].to(DEV)).sum()).backward()
    # oracle
    lin = [m for m in tex.net.net if isinstance(m, torch.nn.Linear)]
    weights = [m.weight.detach().cpu().clone().requires_grad_(True) for m in lin]
    params = tex.encoder.params.detach().cpu().clone().requires_grad_(True)
    tex_o = pl.TextureOracle((lo, hi), tex.encoder.cfg, params, weights, mn.cpu(), mx.cpu())
    p_c = gb_pos.clone().requires_grad_(True)
    cov = mask > 0
    # the oracle evaluates the covered rows only (the others are masked out of the loss on both sides)
    a_o = tex_o.sample((p_c + noise)[cov])
    b_o = tex_o.sample(p_c[cov])
    ((a_o * go[0][cov]).sum() + (b_o * go[1][cov]).sum()).backward()
    for name, x, y in (("jittered", a.detach().cpu()[cov], a_o.detach()), ("plain", b.detach().cpu()[cov], b_o.detach())):
        err = (x - y).abs().max(-1).values
        print(f"\n  tet-res{s['res']} fused field, {name}: {int(cov.sum())} rows, max abs err {float(err.max()):.2e}, rows outside 1e-4: {int((err > 1e-4).sum())}")
        assert int((err > 1e-4).sum()) == 0
    pairs = [("hash-grid table (x128 hook)", tex.encoder.params.grad.cpu(), params.grad * 128.0)]
    pairs += [(f"texture MLP weight {i}", m.weight.grad.cpu(), w.grad) for i, (m, w) in enumerate(zip(lin, weights))]
    for name, x, y in pairs:
        e = float((x - y).abs().max() / y.abs().max())
        print(f"  gradient {name}: max err / max {e:.2e}, rel L2 {float((x - y).norm() / y.norm()):.2e}")
        assert e <= 1e-4, (name, e)
    # d / d position: piecewise-trilinear features -> the slope JUMPS at the cell faces of every level (cells of 1/4096 of the box at the finest):
    # a surface point within float32 round-off of a face takes either slope.  Rows are compared where both sides sit in the same cells.
    gx, gr = p_d.grad.cpu()[cov], p_c.grad[cov]
    bad = ((gx - gr).abs() > 1e-4 * gr.abs() + 1e-5 * float(gr.abs().max())).any(-1)
    print(f"  position gradient: rows outside {int(bad.sum())} of {int(cov.sum())}; rel L2 over the rest {float((gx[~bad] - gr[~bad]).norm() / gr[~bad].norm()):.2e}")
    assert int(bad.sum()) <= max(8, int(cov.sum()) // 5000)
    assert float((gx[~bad] - gr[~bad]).norm() / gr[~bad].norm()) <= 1e-5
    assert float(p_d.grad.cpu()[~cov].abs().max()) == 0.0

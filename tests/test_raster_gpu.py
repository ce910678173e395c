"""Rasterise / interpolate / antialias: HIP path (through the C ABI) vs the CPU oracle.
Integer outputs (triangle ids, adjacency) bit-exact; floats within the tolerance written at each assert."""
import numpy as np
import pytest
import torch

from oracle import raster_oracle as ro
from oracle import scenes

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _scene(kind, seed):
    if kind == "soup":
        verts, tri = scenes.random_soup(600, seed)
    elif kind == "sheet":
        verts, tri = scenes.grid_sheet(14, seed)
    elif kind == "big":   # two screen-filling triangles behind a sheet: exercises the workgroup-per-triangle path
        v1, t1 = scenes.grid_sheet(10, seed)
        v2 = np.array([[-4, -4, -0.6], [4, -4, -0.6], [4, 4, -0.6], [-4, 4, -0.6]], dtype=np.float32)
        verts = np.concatenate([v1, v2])
        tri = np.concatenate([t1, np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32) + len(v1)])
    elif kind == "empty":
        verts, tri = np.zeros((3, 3), np.float32), np.zeros((0, 3), np.int32)
    return verts, tri


def _clip(verts, nviews, first=0):
    mvp, cam = scenes.orbit_views(nviews, first=first)
    pos = ro.xfm_points(torch.tensor(verts)[None], torch.tensor(mvp))
    return pos, mvp, cam


def test_xfm_points():
    from gshell_amd.render import renderutils as ru
    g = torch.Generator().manual_seed(0)
    for Bp in (1, 3):
        pts = torch.rand(Bp, 1000, 3, generator=g)
        mtx = torch.rand(3, 4, 4, generator=g)
        w = torch.rand(3, 1000, 4, generator=g)
        p_ref = pts.clone().requires_grad_(True)
        out_ref = ro.xfm_points(p_ref, mtx)
        (out_ref * w).sum().backward()
        p = pts.to(DEV).requires_grad_(True)
        out = ru.xfm_points(p, mtx.to(DEV))
        (out * w.to(DEV)).sum().backward()
        assert torch.allclose(out.cpu(), out_ref, rtol=1e-6, atol=1e-6)
        assert torch.allclose(p.grad.cpu(), p_ref.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("kind,res", [("soup", (64, 64)), ("sheet", (96, 80)), ("big", (128, 128)), ("empty", (16, 16))])
def test_rasterize_forward(kind, res):
    from gshell_amd.render import rast as dr
    verts, tri = _scene(kind, 1)
    pos, _, _ = _clip(verts, 2)
    H, W = res
    ids_ref = ro.rasterize_ids(pos.numpy(), tri, H, W)
    rast, db, vis = dr.rasterize(dr.RasterizeGLContext(), pos.to(DEV), torch.tensor(tri, device=DEV), (H, W), return_visible=True)
    ids = rast[..., 3].long().cpu().numpy() - 1
    np.testing.assert_array_equal(ids, ids_ref)                      # visibility: bit exact
    if tri.shape[0]:
        rast_ref, db_ref = ro.rast_from_ids(pos, torch.tensor(tri).long(), torch.tensor(ids_ref))
        assert torch.allclose(rast.cpu(), rast_ref, rtol=1e-4, atol=2e-5)   # u, v, z/w
        assert torch.allclose(db.cpu(), db_ref, rtol=1e-3, atol=1e-4)
        want = np.zeros(tri.shape[0], np.uint8)
        want[np.unique(ids_ref[ids_ref >= 0])] = 1
        np.testing.assert_array_equal(vis.cpu().numpy(), want)
    else:
        assert (rast == 0).all() and (db == 0).all()


@pytest.mark.parametrize("kind", ["soup", "sheet", "big"])
def test_rasterize_backward(kind):
    from gshell_amd.render import rast as dr
    verts, tri = _scene(kind, 2)
    pos, _, _ = _clip(verts, 2, first=3)
    H, W = 72, 72
    ids_ref = torch.tensor(ro.rasterize_ids(pos.numpy(), tri, H, W))
    wgt = torch.rand(2, H, W, 2, generator=torch.Generator().manual_seed(1))
    p_ref = pos.clone().requires_grad_(True)
    rast_ref, _ = ro.rast_from_ids(p_ref, torch.tensor(tri).long(), ids_ref)
    (rast_ref[..., :2] * wgt).sum().backward()
    p = pos.to(DEV).requires_grad_(True)
    rast, _ = dr.rasterize(None, p, torch.tensor(tri, device=DEV), (H, W))
    (rast[..., :2] * wgt.to(DEV)).sum().backward()
    g, g_ref = p.grad.cpu(), p_ref.grad
    assert g_ref.abs().max() > 0
    # float atomics: summation order differs -> relative to the gradient scale
    assert (g - g_ref).abs().max() <= 1e-4 * g_ref.abs().max() + 1e-5
    assert (g[..., 2] == 0).all()


@pytest.mark.parametrize("A,Ba", [(1, 1), (3, 1), (4, 2), (7, 1)])
def test_interpolate(A, Ba):
    from gshell_amd.render import rast as dr
    verts, tri = _scene("sheet", 3)
    pos, _, _ = _clip(verts, 2)
    H, W = 64, 64
    tri_l = torch.tensor(tri).long()
    ids = torch.tensor(ro.rasterize_ids(pos.numpy(), tri, H, W))
    rast_ref, db_ref = ro.rast_from_ids(pos, tri_l, ids)
    g = torch.Generator().manual_seed(5)
    attr = torch.rand(Ba, verts.shape[0], A, generator=g)
    wgt = torch.rand(2, H, W, A, generator=g)
    a_ref, r_ref = attr.clone().requires_grad_(True), rast_ref.clone().requires_grad_(True)
    out_ref, da_ref = ro.interpolate(a_ref, r_ref, tri_l, db_ref)
    (out_ref * wgt).sum().backward()
    a, r = attr.to(DEV).requires_grad_(True), rast_ref.to(DEV).requires_grad_(True)
    out, da = dr.interpolate(a, r, torch.tensor(tri, device=DEV), rast_db=db_ref.to(DEV), diff_attrs='all')
    (out * wgt.to(DEV)).sum().backward()
    assert torch.allclose(out.cpu(), out_ref, rtol=1e-5, atol=1e-6)
    assert torch.allclose(da.cpu(), da_ref, rtol=1e-4, atol=1e-5)
    assert (a.grad.cpu() - a_ref.grad).abs().max() <= 1e-4 * a_ref.grad.abs().max()
    assert torch.allclose(r.grad.cpu()[..., :2], r_ref.grad[..., :2], rtol=1e-4, atol=1e-5)
    out2, none = dr.interpolate(a, r, torch.tensor(tri, device=DEV))
    assert none is None and torch.equal(out2, out)
    # diff_attrs as a list of attribute indices (nvdiffrast's other form; the reference passes 'all'): the chosen pairs, in the order given
    pick = [A - 1, 0] if A > 1 else [0]
    a3, r3 = attr.to(DEV).requires_grad_(True), rast_ref.to(DEV).requires_grad_(True)
    out3, da3 = dr.interpolate(a3, r3, torch.tensor(tri, device=DEV), rast_db=db_ref.to(DEV), diff_attrs=pick)
    cols = [c for i in pick for c in (2 * i, 2 * i + 1)]
    assert torch.equal(out3, out) and da3.shape[-1] == 2 * len(pick) and torch.equal(da3, da[..., cols])
    assert torch.allclose(da3.cpu(), da_ref[..., cols], rtol=1e-4, atol=1e-5)
    (out3 * wgt.to(DEV)).sum().backward()                    # (the derivative output carries no gradient here, as on the reference's path: render.py:273-279 is no_grad)
    assert (a3.grad - a.grad).abs().max() <= 1e-5 * a.grad.abs().max()      # float atomics: order differs between the two launches
    with pytest.raises(ValueError):
        dr.interpolate(a, r, torch.tensor(tri, device=DEV), rast_db=db_ref.to(DEV), diff_attrs=[A])


@pytest.mark.parametrize("kind", ["soup", "sheet", "big"])
def test_antialias(kind):
    from gshell_amd.render import rast as dr
    verts, tri = _scene(kind, 4)
    pos, _, _ = _clip(verts, 2, first=1)
    H, W = 80, 80
    tri_l = torch.tensor(tri).long()
    tri_d = torch.tensor(tri, device=DEV)
    ids = torch.tensor(ro.rasterize_ids(pos.numpy(), tri, H, W))
    rast_ref, _ = ro.rast_from_ids(pos, tri_l, ids)
    opp_ref = ro.tri_adjacency(tri)
    topo = dr.AATopology(tri_d, verts.shape[0])
    np.testing.assert_array_equal(topo.opp.cpu().numpy(), opp_ref)                 # adjacency: bit exact
    g = torch.Generator().manual_seed(7)
    color = torch.rand(2, H, W, 5, generator=g)
    wgt = torch.rand(2, H, W, 5, generator=g)
    p_ref, c_ref = pos.clone().requires_grad_(True), color.clone().requires_grad_(True)
    alpha_ref = ro.aa_alpha(rast_ref, p_ref, tri_l, torch.tensor(opp_ref))
    out_ref = ro.aa_apply(c_ref, alpha_ref)
    (out_ref * wgt).sum().backward()
    assert (alpha_ref != 0).sum() > 20
    p, c = pos.to(DEV).requires_grad_(True), color.to(DEV).requires_grad_(True)
    rast_d = rast_ref.to(DEV)
    alpha = dr.aa_analyze(rast_d, p, tri_d, topo)
    # the silhouette decision is a float comparison: allow a handful of pairs to flip, none to drift
    a, ar = alpha.cpu(), alpha_ref.detach()
    mism = ((a - ar).abs() > 1e-4)
    assert mism.float().mean() < 1e-3, f"{int(mism.sum())} alpha entries differ"
    out = dr.antialias(c, rast_d, p, tri_d)
    (out * wgt.to(DEV)).sum().backward()
    ok = ~(mism.any(-1))
    ok = ok & torch.roll(ok, 1, 1) & torch.roll(ok, 1, 2)
    assert torch.allclose(out.cpu()[ok], out_ref.detach()[ok], rtol=1e-5, atol=1e-5)
    if not mism.any():
        assert torch.allclose(c.grad.cpu(), c_ref.grad, rtol=1e-4, atol=1e-5)
        assert (p.grad.cpu() - p_ref.grad).abs().max() <= 2e-4 * p_ref.grad.abs().max() + 1e-6
    # stacked form == per-buffer form
    o1, o2 = dr.antialias_stacked([c[..., :2], c[..., 2:]], rast_d, p, tri_d, topo)
    assert torch.equal(torch.cat([o1, o2], -1), out)


def test_face_normals():
    from gshell_amd.render import rast as dr
    verts, tri = _scene("sheet", 6)
    pos, _, _ = _clip(verts, 2)
    H, W = 56, 56
    tri_l = torch.tensor(tri).long()
    ids = torch.tensor(ro.rasterize_ids(pos.numpy(), tri, H, W))
    rast_ref, _ = ro.rast_from_ids(pos, tri_l, ids)
    w = torch.randn(2, H, W, 3, generator=torch.Generator().manual_seed(2))
    v_ref = torch.tensor(verts).requires_grad_(True)
    v0, v1, v2 = v_ref[tri_l[:, 0]], v_ref[tri_l[:, 1]], v_ref[tri_l[:, 2]]
    c = torch.cross(v1 - v0, v2 - v0, dim=-1)
    fn = c / torch.sqrt(torch.clamp((c * c).sum(-1, keepdim=True), min=1e-20))
    ref = torch.where((ids >= 0)[..., None], fn[ids.clamp(min=0)], torch.zeros(()))
    (ref * w).sum().backward()
    v = torch.tensor(verts, device=DEV).requires_grad_(True)
    out = dr.face_normals(v, torch.tensor(tri, device=DEV), rast_ref.to(DEV))
    (out * w.to(DEV)).sum().backward()
    assert torch.allclose(out.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    assert (v.grad.cpu() - v_ref.grad).abs().max() <= 1e-4 * v_ref.grad.abs().max()


def test_rasterize_near_plane_clipping_matches_oracle():
    """Triangles with a vertex behind the eye (camera inside / next to the surface): ids equal to the oracle's restatement of
    the view-volume clip, coverage non-empty, and the antialias cache must not serve a stale analysis for new tensors at
    recycled addresses."""
    from gshell_amd.render import rast as dr
    n, f = 0.1, 100.0
    proj = np.array([[1.5, 0, 0, 0], [0, 1.5, 0, 0], [0, 0, -(f + n) / (f - n), -2 * f * n / (f - n)], [0, 0, -1, 0]], dtype=np.float32)
    rng = np.random.default_rng(5)
    # a floor and a wall that both pass the eye, plus small triangles scattered around (and behind) it
    quad = np.array([[-1, -0.5, -4.0], [1, -0.5, -4.0], [1, -0.5, 3.0], [-1, -0.5, 3.0], [0.7, -1, -5.0], [0.7, 1, -5.0], [0.7, 1, 2.0], [0.7, -1, 2.0]],
                    dtype=np.float32)
    centres = rng.uniform(-1, 1, (20, 1, 3)).astype(np.float32) * np.array([0.6, 0.3, 2.0], np.float32) + np.array([0, 0.2, -1.5], np.float32)
    small = (centres + rng.uniform(-0.12, 0.12, (20, 3, 3)).astype(np.float32)).reshape(60, 3)      # some straddle the eye plane too
    verts = np.concatenate([quad, small])
    tri = np.concatenate([np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7]]), 8 + np.arange(60).reshape(20, 3)]).astype(np.int32)
    pos = (np.concatenate([verts, np.ones((len(verts), 1), np.float32)], 1) @ proj.T)[None].astype(np.float32)
    assert (pos[0, :, 3] <= 0).sum() >= 4
    H, W = 72, 96
    ids_ref = ro.rasterize_ids(pos, tri, H, W)
    rast, _ = dr.rasterize(None, torch.tensor(pos, device=DEV), torch.tensor(tri, device=DEV), (H, W))
    ids = rast[..., 3].long().cpu().numpy() - 1
    assert (ids_ref[0] == 0).sum() + (ids_ref[0] == 1).sum() > 300          # the near-clipped floor is visible
    assert np.array_equal(ids, ids_ref)


def test_antialias_cache_is_keyed_on_tensor_identity():
    """dr.antialias cached its analysis on (data_ptr, _version); kernel-written tensors keep _version 0 and the
    caching allocator recycles addresses, so the next iteration could be served the previous alpha.  The cache now holds the
    tensors themselves: new tensors at the same addresses are analysed afresh."""
    from gshell_amd.render import rast as dr
    verts, tri = _scene("sheet", 2)
    tri_d = torch.tensor(tri, device=DEV)
    outs = []
    for first in (0, 3):
        dr.antialias_cache_clear() if first == 0 else None
        pos, _, _ = _clip(verts, 1, first=first)
        pos_d = pos.to(DEV)
        rast, _ = dr.rasterize(None, pos_d, tri_d, (64, 64))
        col = (rast[..., 3:4] > 0).float().expand(-1, -1, -1, 3).contiguous()
        a = dr.antialias(col, rast, pos_d, tri_d)
        dr.antialias_cache_clear()
        b = dr.antialias(col, rast, pos_d, tri_d)
        assert torch.equal(a, b)
        ptrs = (rast.data_ptr(), pos_d.data_ptr())
        outs.append((a, ptrs))
        del rast, pos_d, col          # free the blocks so that the second pass may receive the same addresses
    assert not torch.equal(outs[0][0], outs[1][0])


@pytest.mark.parametrize("channels,skip", [((3, 3, 1), None), ((3, 3), None), ((3, 3, 1), 1), ((2,), None), ((1, 4, 3, 2), 2)])
def test_interpolate_groups_equals_the_stacked_interpolation(channels, skip):
    """dr.interpolate_groups (one launch, one contiguous output per attribute tensor: the g-buffer of render_layer) against
    dr.interpolate of torch.cat(attrs, -1) + channel slices: values and the barycentric gradient bit-identical, attribute gradients
    up to the float atomics' order; `skip`: an output nobody differentiates (NULL g_out, no gradient tensor for that attribute)."""
    from gshell_amd.render import rast as dr
    verts, tri = _scene("sheet", 3)
    pos, _, _ = _clip(verts, 2)
    H, W = 64, 64
    tri_l = torch.tensor(tri).long()
    ids = torch.tensor(ro.rasterize_ids(pos.numpy(), tri, H, W))
    rast_ref, _ = ro.rast_from_ids(pos, tri_l, ids)
    g = torch.Generator().manual_seed(15)
    attrs = [torch.rand(verts.shape[0], c, generator=g) for c in channels]
    wgts = [torch.rand(2, H, W, c, generator=g).to(DEV) for c in channels]
    tri_d = torch.tensor(tri, device=DEV)
    a1 = [a.to(DEV).requires_grad_(True) for a in attrs]
    r1 = rast_ref.to(DEV).requires_grad_(True)
    stacked, _ = dr.interpolate(torch.cat(a1, -1)[None], r1, tri_d)
    parts = torch.split(stacked, list(channels), dim=-1)
    sum((p * w).sum() for k, (p, w) in enumerate(zip(parts, wgts)) if k != skip).backward()
    a2 = [a.to(DEV).requires_grad_(True) for a in attrs]
    r2 = rast_ref.to(DEV).requires_grad_(True)
    outs = dr.interpolate_groups(a2, r2, tri_d)
    sum((o * w).sum() for k, (o, w) in enumerate(zip(outs, wgts)) if k != skip).backward()
    for o, p in zip(outs, parts):
        assert o.is_contiguous() and torch.equal(o, p)
    assert torch.equal(r1.grad, r2.grad)
    for k, (x, y) in enumerate(zip(a1, a2)):
        if k == skip:
            assert y.grad is None and float(x.grad.abs().max()) == 0
        else:
            assert float((x.grad - y.grad).abs().max()) <= 1e-5 * float(x.grad.abs().max())


@pytest.mark.parametrize("kind,C", [("soup", 5), ("sheet", 45), ("big", 3)])
def test_inplace_antialias_equals_the_streaming_one_bit_for_bit(kind, C):
    """dr.antialias_stacked([frame], inplace=True) (two sparse launches over the silhouette pixels, the frame and the incoming gradient updated
    in place) against the out-of-place kernels: output, colour gradient and position gradient... the first two bit-identical, the position
    gradient up to the float atomics' order (the in-place kernel sums a pixel's channels in one lane, the streaming one with atomics)."""
    from gshell_amd.render import rast as dr
    verts, tri = _scene(kind, 4)
    pos, _, _ = _clip(verts, 2, first=1)
    H, W = 80, 80
    tri_d = torch.tensor(tri, device=DEV)
    rast_d, _ = dr.rasterize(None, pos.to(DEV), tri_d, (H, W))
    g = torch.Generator().manual_seed(17)
    color = torch.rand(2, H, W, C, generator=g)
    wgt = torch.rand(2, H, W, C, generator=g).to(DEV)
    res = []
    for inplace in (False, True):
        c = color.to(DEV).requires_grad_(True)
        p = pos.to(DEV).requires_grad_(True)
        frame = c * 1.0                                   # a non-leaf the caller owns
        keep = frame.detach().clone()
        out = dr.antialias_stacked([frame], rast_d.detach(), p, tri_d, inplace=inplace)[0]
        assert (out.data_ptr() == frame.data_ptr()) == inplace
        (out * wgt).sum().backward()
        res.append((out.detach().clone(), c.grad.clone(), p.grad.clone(), keep))
    assert float((res[0][0] - res[0][3]).abs().max()) > 0, "the scene has no silhouette pixel: nothing tested"
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    scale = float(res[0][2].abs().max())
    assert scale > 0 and float((res[0][2] - res[1][2]).abs().max()) <= 1e-5 * scale


def test_inplace_antialias_backward_does_not_corrupt_a_gradient_someone_else_holds():
    """the in-place backward used to overwrite its incoming gradient tensor unconditionally.  A consumer may hand on a tensor that
    somebody else also holds (AddBackward gives ONE tensor to both operands; a tensor hook can keep what it is shown): unless the caller
    vouches for exclusivity (`grad_exclusive`, render_mesh does) the gradient is copied first.  Either way the results equal the
    out-of-place kernels'."""
    from gshell_amd.render import rast as dr
    verts, tri = _scene("sheet", 4)
    pos, _, _ = _clip(verts, 2, first=1)
    H, W, C = 80, 80, 7
    tri_d = torch.tensor(tri, device=DEV)
    rast_d, _ = dr.rasterize(None, pos.to(DEV), tri_d, (H, W))
    g = torch.Generator().manual_seed(21)
    color = torch.rand(2, H, W, C, generator=g)
    wgt = torch.rand(2, H, W, C, generator=g).to(DEV)
    res = {}
    for mode in ("streaming", "inplace", "inplace+exclusive", "inplace+shared"):
        c = color.to(DEV).requires_grad_(True)
        p = pos.to(DEV).requires_grad_(True)
        other = torch.zeros(2, H, W, C, device=DEV, requires_grad=True)
        frame = c * 1.0
        before = dr.INPLACE_GRAD_COPIES[0]
        out = dr.antialias_stacked([frame], rast_d.detach(), p, tri_d, inplace=mode != "streaming", grad_exclusive=mode == "inplace+exclusive")[0]
        kept = []
        if mode == "inplace+shared":          # AddBackward hands the SAME gradient tensor to `out` and to `other`; a hook keeps it as well
            out.register_hook(lambda t: kept.append(t))
            y = out + other
            y.backward(wgt)
        else:
            (out * wgt).sum().backward()
        res[mode] = (c.grad.clone(), p.grad.clone(), dr.INPLACE_GRAD_COPIES[0] - before, other.grad, kept)
    for mode in ("inplace", "inplace+exclusive", "inplace+shared"):
        assert torch.equal(res[mode][0], res["streaming"][0]), mode
        scale = float(res["streaming"][1].abs().max())
        assert float((res[mode][1] - res["streaming"][1]).abs().max()) <= 1e-5 * scale, mode
    assert res["inplace"][2] == 1 and res["inplace+shared"][2] == 1 and res["inplace+exclusive"][2] == 0
    assert torch.equal(res["inplace+shared"][3], wgt), "the gradient shared with the other operand of the addition was overwritten"
    assert torch.equal(res["inplace+shared"][4][0], wgt), "the gradient a tensor hook kept was overwritten"

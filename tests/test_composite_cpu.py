"""The stacked whole-frame composite of render.render_mesh (one cat + 0/1 alpha-pick matrix + where + lerp) against the
reference's per-buffer loop (render/render.py:417-433 of the reference: alpha = cover * buf[..., -1:], lerp(bg, [rgb, 1], alpha))."""
import torch

from gshell_amd.render import render


def test_stacked_composite_equals_per_buffer_loop():
    torch.manual_seed(0)
    B, H, W = 2, 5, 7
    keys, sizes = ['kd', 'shaded', 'msdf_image', 'wide', 'z_grad'], [4, 4, 1, 7, 4]
    bufs = {k: torch.rand(B, H, W, c, requires_grad=True) for k, c in zip(keys, sizes)}
    cover = (torch.rand(B, H, W, 1) > 0.4).float()
    background = torch.cat((torch.rand(B, H, W, 3), torch.zeros(B, H, W, 1)), -1)

    comps = []
    for k, b in bufs.items():
        a = cover * b[..., -1:]
        fg = torch.cat((b[..., :-1], torch.ones_like(b[..., -1:])), -1)
        bg = background if k == 'shaded' else torch.zeros_like(fg)
        comps.append(torch.lerp(bg, fg, a))
    ref = torch.cat(comps, -1)
    w = torch.rand_like(ref)
    g_ref = torch.autograd.grad((ref * w).sum(), list(bufs.values()))

    lay = render._composite_layout(tuple(sizes), torch.device("cpu"))
    stacked = torch.cat([bufs[k] for k in keys], -1)
    a = cover * torch.matmul(stacked, lay['pick_alpha'])
    fg = torch.where(lay['is_alpha'], lay['one'], stacked)
    bg = torch.zeros(B, H, W, sum(sizes))
    o = sum(sizes[:keys.index('shaded')])
    bg[..., o:o + 4] = background
    comp = torch.lerp(bg, fg, a)
    assert torch.equal(comp, ref)                       # the 0/1 matrix product is exact
    g = torch.autograd.grad((comp * w).sum(), list(bufs.values()))
    for x, y in zip(g, g_ref):
        assert torch.allclose(x, y, rtol=0, atol=1e-6)
    # the 1-channel buffer is its own alpha: composite = cover * value (lerp(0, 1, cover * v))
    assert torch.equal(comp[..., 8:9], cover * bufs['msdf_image'])


def test_visible_boundary_weight_equals_the_reference_mask():
    """mSDF 'close' regulariser (reference gshell_tets_geometry.py:344-353): the sync-free scatter-max weight must select exactly
    the boundary vertices the reference's unique -> gather -> boolean-mask chain selects, and the weighted Huber sum must equal
    the Huber sum over the compacted values."""
    import torch.nn.functional as F
    from gshell_amd.geometry.gshell_tets_geometry import visible_boundary_weight
    g = torch.Generator().manual_seed(0)
    nwt, nb, T = 40, 90, 200
    tri = torch.randint(0, nwt + nb, (T, 3), generator=g)
    flags = (torch.rand(T, generator=g) < 0.3).to(torch.uint8)
    msdf_boundary = torch.randn(nb, generator=g) * 0.01
    w = visible_boundary_weight(tri, flags, nwt, nb)
    vis_tris = torch.nonzero(flags).reshape(-1)
    vis_verts = tri[vis_tris].reshape(-1)
    mask = torch.zeros(nb, dtype=torch.bool)
    mask[vis_verts[vis_verts >= nwt] - nwt] = True
    assert torch.equal(w > 0, mask) and set(w.unique().tolist()) <= {0.0, 1.0}
    eps = torch.full((1,), 1e-3)
    bm = msdf_boundary[mask]
    ref = F.huber_loss(bm.clamp(max=eps), eps.expand(bm.size(0)), reduction='sum')
    new = (F.huber_loss(msdf_boundary.clamp(max=eps), eps.expand(nb), reduction='none') * w).sum()
    assert torch.allclose(ref, new, rtol=1e-6, atol=1e-12)
    assert float(visible_boundary_weight(tri, torch.zeros(T, dtype=torch.uint8), nwt, nb).sum()) == 0.0

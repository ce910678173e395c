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

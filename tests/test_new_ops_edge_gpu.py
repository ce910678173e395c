"""Empty / ragged inputs of the round-2 entry points (the reference's operators accept empty meshes and frames: an iteration
whose surface vanished must not crash)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_texture_field_with_no_rows_and_with_no_covered_rows():
    from gshell_amd.render.mlptexture import MLPTexture3D
    aabb = (torch.tensor([-1.0, -1, -1], device=DEV), torch.tensor([1.0, 1, 1], device=DEV))
    mn, mx = torch.zeros(6, device=DEV), torch.ones(6, device=DEV)
    tex = MLPTexture3D(aabb, channels=6, min_max=[mn, mx])
    out = tex.sample(torch.zeros(0, 3, device=DEV))
    assert out.shape == (0, 6)
    pos = torch.rand(1, 16, 16, 3, device=DEV, requires_grad=True)
    mask = torch.zeros(1, 16, 16, 1, device=DEV)
    a, b = tex.sample_many([pos, pos * 0.5], mask)
    assert torch.allclose(a, (0.5 * (mx - mn) + mn).expand_as(a)) and torch.equal(a, b)      # all-zero feature rows
    (a.sum() + b.sum()).backward()
    assert float(pos.grad.abs().max()) == 0.0 and float(tex.encoder.params.grad.abs().max()) == 0.0


def test_msdf_regularisers_and_visibility_weights_on_empty_inputs():
    from gshell_amd.geometry.gshell_tets_geometry import _MsdfRegFn, boundary_weight
    w = boundary_weight(torch.zeros(0, 3, dtype=torch.int32, device=DEV), torch.zeros(0, dtype=torch.uint8, device=DEV), 10, 5)
    assert w.shape == (5,) and float(w.abs().max()) == 0.0
    assert boundary_weight(torch.zeros(4, 3, dtype=torch.int32, device=DEV), torch.ones(4, dtype=torch.uint8, device=DEV), 1, 0).shape == (0,)
    m = torch.randn(100, device=DEV, requires_grad=True)
    two = _MsdfRegFn.apply(m, torch.zeros(0, 1, device=DEV), None, 1e-3, 1.0, 2.0)
    assert float(two[1]) == 0.0 and float(two[0]) > 0.0
    two.sum().backward()
    assert torch.isfinite(m.grad).all()
    e = torch.zeros(0, device=DEV, requires_grad=True)
    assert float(_MsdfRegFn.apply(e, torch.zeros(0, 1, device=DEV), None, 1e-3, 1.0, 2.0).abs().max()) == 0.0


def test_masked_bilateral_with_nothing_wanted_and_odd_sizes():
    from gshell_amd.render import optixutils as ou
    for (B, H, W) in ((1, 5, 7), (2, 33, 17)):
        col = torch.rand(B, H, W, 3, device=DEV, requires_grad=True)
        nrm = torch.nn.functional.normalize(torch.randn(B, H, W, 3, device=DEV), dim=-1)
        zdz = torch.rand(B, H, W, 2, device=DEV)
        out = ou.bilateral_denoiser_raw(col, nrm, zdz, 2.0, torch.zeros(B, H, W, device=DEV))
        assert torch.equal(out, torch.tensor([0, 0, 0, 1e-4], device=DEV).expand(B, H, W, 4))
        out.sum().backward()
        assert float(col.grad.abs().max()) == 0.0
        full = ou.bilateral_denoiser_raw(col.detach(), nrm, zdz, 2.0, torch.ones(B, H, W, device=DEV))
        assert torch.equal(full, ou.bilateral_denoiser_raw(col.detach(), nrm, zdz, 2.0))


def test_env_shade_saved_backward_with_no_covered_pixel():
    from gshell_amd.render import optixutils as ou
    from oracle import pixel_oracle as po
    B, H, W = 1, 8, 8
    verts = torch.tensor([[0.0, 0, 0], [1, 0, 0], [0, 1, 0]], device=DEV)
    tri = torch.tensor([[0, 1, 2]], dtype=torch.int32, device=DEV)
    ctx = ou.OptiXContext()
    ou.optix_build_bvh(ctx, verts, tri, rebuild=1)
    light = torch.rand(16, 32, 3) + 0.1
    pdf, rows, cols = po.update_pdf(light)
    leaves = [torch.rand(B, H, W, 3, device=DEV, requires_grad=True) for _ in range(4)]
    lgt = light.to(DEV).requires_grad_(True)
    d, s = ou.optix_env_shade(ctx, torch.zeros(B, H, W, device=DEV), leaves[0].detach(), leaves[0], leaves[1], torch.zeros(B, 1, 1, 3, device=DEV),
                              leaves[2], leaves[3], lgt, pdf.to(DEV), rows[:, 0].to(DEV), cols.to(DEV), BSDF='pbr', n_samples_x=2, rnd_seed=3)
    assert float(d.abs().max()) == 0.0 and float(s.abs().max()) == 0.0
    (d.sum() + s.sum()).backward()
    assert float(lgt.grad.abs().max()) == 0.0 and all(float(t.grad.abs().max()) == 0.0 for t in leaves)


def test_frame_sums_colour_term_and_eikonal_loss_on_tiny_inputs():
    from gshell_amd.geometry.mlp import MLP, eikonal_sq_sum
    from gshell_amd.render import regularizer as R
    st = torch.rand(1, 1, 1, 4, device=DEV, requires_grad=True)
    ref = torch.tensor([[[[0.2, 0.4, 0.6, 1.0]]]], device=DEV)
    fs = R.frame_sums((st, ['shaded'], [4]), ref, (0, 1))
    assert fs.shape == (10,) and torch.isfinite(fs).all()
    fs.sum().backward()
    assert torch.isfinite(st.grad).all()
    torch.manual_seed(0)
    net = MLP(n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3]).to(DEV)
    l1 = eikonal_sq_sum(net, torch.rand(1, 3, device=DEV) - 0.5)           # one sample = 4 virtual rows of a 128-row pad
    l1.backward()
    assert torch.isfinite(l1) and all(torch.isfinite(p.grad).all() for p in net.parameters())

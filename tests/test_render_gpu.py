"""End-to-end parity of gshell_amd.render.render.render_mesh (HIP ops, fwd + bwd) against the composed CPU oracle
(oracle/pipeline_oracle.py), on the mesh of a real extraction, with the reference's noise tensors injected on both sides."""
import numpy as np
import pytest
import torch

from oracle import fields, mtets_oracle, pipeline_oracle as pl, pixel_oracle as po, scenes

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close_frac(a, b, rtol):
    scale = b.abs().max().clamp(min=1e-12)
    return float(((a - b).abs() <= rtol * b.abs() + rtol * scale).float().mean())


@pytest.mark.parametrize("denoise", [False, True])
def test_render_mesh_matches_pipeline_oracle(denoise):
    from gshell_amd import grid
    from gshell_amd.denoiser.denoiser import BilateralDenoiser
    from gshell_amd.render import light, mesh, mlptexture, optixutils as ou, render
    from gshell_amd.train import default_flags
    B, H, W, n = 2, 40, 40, 2
    verts, tets = grid.bcc_grid(10)
    vn = verts.numpy()
    ex = mtets_oracle.extract(torch.tensor(vn), torch.tensor(fields.make_sdf(vn, "skirt", 3)), torch.tensor(fields.make_msdf(vn, "wavy", 3)), tets,
                              with_tangents=False)
    v_pos, faces, msdf = (ex["verts_aug"] * 2.2).detach(), ex["faces_aug"], ex["msdf"].detach()
    mvp, cam = scenes.orbit_views(B, first=2)
    gen = torch.Generator().manual_seed(0)
    noise = {'jitter': torch.randn(B, H, W, 2, generator=gen) * 0.005, 'texture': torch.randn(B, H, W, 3, generator=gen) * 0.01,
             'tangent': torch.randn(B, H, W, 3, generator=gen)}
    light_base = torch.rand(16, 32, 3, generator=gen) + 0.2
    bg = torch.rand(B, H, W, 3, generator=gen)
    perms = torch.argsort(torch.rand(ou.PERM_ROWS, n * n, generator=gen), dim=-1).int()
    seed, shadow = 17, 0.8
    sigma = 1.2

    # ---- product path
    aabb = (torch.tensor([-1.2, -1.2, -1.2], device=DEV), torch.tensor([1.2, 1.2, 1.2], device=DEV))
    mn = torch.tensor([0, 0, 0, 0, 0.001, 0], dtype=torch.float32, device=DEV)
    mx = torch.tensor([1, 1, 1, 0, 1.0, 1], dtype=torch.float32, device=DEV)
    torch.manual_seed(3)
    tex = mlptexture.MLPTexture3D(aabb, channels=6, min_max=[mn, mx])
    with torch.no_grad():
        tex.encoder.params.mul_(3000.0)        # make the texture vary visibly over the object
    FLAGS = default_flags(n_samples=n)
    vd = v_pos.to(DEV).requires_grad_(True)
    md = msdf.to(DEV).requires_grad_(True)
    lgt = light.EnvironmentLight(light_base.to(DEV).requires_grad_(True))
    imesh = mesh.auto_normals(mesh.Mesh(vd, faces.to(DEV), material={'kd_ks': tex, 'bsdf': 'pbr', 'no_perturbed_nrm': False}))
    ctx = ou.OptiXContext()
    ou.optix_build_bvh(ctx, vd.detach(), imesh.faces_i32(), 1)
    ou.set_random_perm(n, perms.to(DEV))
    render.noise_override = {k: t.to(DEV) for k, t in noise.items()}
    render.rnd_seed = seed
    den = BilateralDenoiser()
    den.set_influence(sigma / 2)
    try:
        out = render.render_mesh(FLAGS, None, imesh, torch.tensor(mvp, device=DEV), torch.tensor(cam, device=DEV), lgt, [H, W], spp=1, msaa=True,
                                 background=bg.to(DEV), optix_ctx=ctx, denoiser=den if denoise else None, shadow_scale=shadow, use_uv=False,
                                 extra_dict={'msdf': md})
    finally:
        render.noise_override = None

    # ---- oracle
    weights = [m.weight.detach().cpu().clone().requires_grad_(True) for m in tex.net.net if isinstance(m, torch.nn.Linear)]
    params = tex.encoder.params.detach().cpu().clone().requires_grad_(True)
    tex_o = pl.TextureOracle((aabb[0].cpu(), aabb[1].cpu()), tex.encoder.cfg, params, weights, mn.cpu(), mx.cpu())
    v_ref, m_ref, l_ref = v_pos.clone().requires_grad_(True), msdf.clone().requires_grad_(True), light_base.clone().requires_grad_(True)
    ref = pl.render_mesh(v_ref, faces, po.auto_normals(v_ref, faces), m_ref, torch.tensor(mvp), torch.tensor(cam), l_ref, bg, noise, tex_o, n, seed, shadow,
                         perms.numpy(), denoise_sigma=sigma if denoise else None, resolution=(H, W))

    assert set(out.keys()) == set(ref.keys())
    np.testing.assert_array_equal(out['visible_triangles'].cpu().numpy(), ref['visible_triangles'].numpy())    # integer: bit exact
    for key in ref:
        if key == 'visible_triangles':
            continue
        frac = _close_frac(out[key].detach().cpu(), ref[key].detach(), 1e-4)
        floor = 0.999       # measured on MI355X (r02): 1.0000 for every buffer of every case (MC sample-placement flips: see test_shade_gpu)
        print(f"  buffer {key}: pixels within 1e-4: {frac:.4f}")
        assert frac >= floor, (key, frac)

    gen2 = torch.Generator().manual_seed(9)
    w_sh, w_ms = torch.rand(B, H, W, 4, generator=gen2), torch.rand(B, H, W, 1, generator=gen2)

    def loss(o, dev):
        return (o['shaded'] * w_sh.to(dev)).sum() + (o['msdf_image'] * w_ms.to(dev)).sum() + o['kd_grad'].sum() * 0.1 + o['normal'].sum() * 0.05
    loss(out, DEV).backward()
    loss(ref, "cpu").backward()
    pairs = [("v_pos", vd.grad.cpu(), v_ref.grad), ("msdf", md.grad.cpu(), m_ref.grad), ("light", lgt.base.grad.cpu(), l_ref.grad),
             ("hash grid", tex.encoder.params.grad.cpu(), params.grad * 128.0)]       # product keeps the reference's x128 gradient hook
    lin = [m for m in tex.net.net if isinstance(m, torch.nn.Linear)]
    pairs += [(f"mlp{i}", m.weight.grad.cpu(), w.grad) for i, (m, w) in enumerate(zip(lin, weights))]
    for name, a, b in pairs:
        assert torch.isfinite(a).all(), name
        assert b.abs().max() > 0, name
        # float atomics + a few MC placement flips: compare in aggregate (relative L2) and element-wise coverage
        rel = float((a - b).norm() / b.norm())
        print(f"  end-to-end gradient {name}: relative L2 error {rel:.2e}")
        # measured (r02): v_pos 1.4e-4 .. 1.7e-4 (float atomics through the silhouette antialiasing), everything else <= 8e-5
        assert rel < (5e-4 if name == "v_pos" else 2e-4), (name, rel)

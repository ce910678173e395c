"""End-to-end parity of gshell_amd.render.render.render_mesh (HIP ops, fwd + bwd) against the composed CPU oracle
(oracle/pipeline_oracle.py), on the mesh of a real extraction, with the reference's noise tensors injected on both sides."""
import numpy as np
import pytest
import torch

from oracle import fields, mtets_oracle, pipeline_oracle as pl, pixel_oracle as po, scenes

pytestmark = pytest.mark.gpu
DEV = "cuda"


def render_both(denoise, tex_levels=16):
    """-> product outputs (HIP), oracle outputs (CPU), and the leaf tensors of both sides (also used by tools/render_grad_diag.py).
    tex_levels: hash-grid levels that carry texture (the finer ones are zeroed) -- see test_render_mesh_matches_pipeline_oracle."""
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
        from oracle import hashgrid_oracle as ho
        metas, _ = ho.level_meta(*tex.encoder.cfg)
        if tex_levels < len(metas):
            tex.encoder.params[metas[tex_levels][2] * tex.encoder.cfg[1]:] = 0.0
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

    lin = [m for m in tex.net.net if isinstance(m, torch.nn.Linear)]
    leaves = dict(v_pos=(vd, v_ref), msdf=(md, m_ref), light=(lgt.base, l_ref), hash_grid=(tex.encoder.params, params),
                  **{f"mlp{i}": (m.weight, w) for i, (m, w) in enumerate(zip(lin, weights))})
    return out, ref, leaves, (B, H, W)


@pytest.mark.parametrize("denoise,tex_levels", [(False, 16), (True, 16), (False, 6), (True, 6)])
def test_render_mesh_matches_pipeline_oracle(denoise, tex_levels):
    """tex_levels = 16: all hash-grid levels carry texture (amplitude x 3000).  The texture is then piecewise trilinear with cells of
    1/4096 of the box: d texture / d position JUMPS at every cell face, and a surface point within float32 round-off of a face takes
    either side's slope -- the float32 and float64 runs of the CPU oracle itself differ by 4e-3 (relative L2) in the v_pos
    gradient (measured), the HIP path by 1.3e-4 .. 1.7e-4 from the float32 oracle, concentrated in ~10 vertices
    (tools/render_grad_diag.py).  tex_levels = 6: the fine levels are zeroed (cells >= 1/100 of the box, slope jumps 40 x
    smaller): there the end-to-end position gradient must meet the north-star 1e-4."""
    out, ref, leaves, (B, H, W) = render_both(denoise, tex_levels)
    assert set(out.keys()) == set(ref.keys())
    np.testing.assert_array_equal(out['visible_triangles'].cpu().numpy(), ref['visible_triangles'].numpy())    # integer: bit exact
    n = 2                                                # n_samples of render_both: 2 n^2 = 8 Monte-Carlo samples per pixel
    for key in ref:
        if key == 'visible_triangles':
            continue
        a, b = out[key].detach().cpu(), ref[key].detach()
        scale = b.abs().max().clamp(min=1e-12)
        dev = ((a - b).abs() - 1e-4 * b.abs()).amax(dim=-1) / scale
        bad = dev > 1e-4
        # Every pixel within 1e-4 (measured on MI355X, r02 / r03: all of them, every buffer, every case).  The vertex normals that feed
        # the sampler are float-atomic sums on the GPU (as in the reference), so an ulp of run-to-run noise can move ONE of a pixel's
        # 2 n^2 samples across a CDF cell / probe texel (tests/test_ref_parity_gpu.py lists such samples against the reference
        # kernel): tolerate at most two such pixels per buffer, each off by no more than two samples' weight -- not a percentage.
        print(f"  buffer {key}: pixels outside 1e-4: {int(bad.sum())} of {bad.numel()}")
        assert int(bad.sum()) <= 2, (key, int(bad.sum()))
        assert float(dev.max()) <= 2.0 / (2 * n * n) * 4.0, (key, float(dev.max()))

    gen2 = torch.Generator().manual_seed(9)
    w_sh, w_ms = torch.rand(B, H, W, 4, generator=gen2), torch.rand(B, H, W, 1, generator=gen2)

    def loss(o, dev):
        return (o['shaded'] * w_sh.to(dev)).sum() + (o['msdf_image'] * w_ms.to(dev)).sum() + o['kd_grad'].sum() * 0.1 + o['normal'].sum() * 0.05
    loss(out, DEV).backward()
    loss(ref, "cpu").backward()
    # the product keeps the reference's x128 gradient hook on the hash-grid parameters
    pairs = [(name, a.grad.cpu(), b.grad * (128.0 if name == "hash_grid" else 1.0)) for name, (a, b) in leaves.items()]
    for name, a, b in pairs:
        assert torch.isfinite(a).all(), name
        assert b.abs().max() > 0, name
        # float atomics + a few MC placement flips: compare in aggregate (relative L2) and element-wise coverage
        rel = float((a - b).norm() / b.norm())
        print(f"  end-to-end gradient {name}: relative L2 error {rel:.2e}")
        assert rel < (5e-4 if (name == "v_pos" and tex_levels == 16) else 1e-4 if name == "v_pos" else 2e-4), (name, rel)

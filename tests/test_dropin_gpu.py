"""Drop-in use exactly as the reference's train script does it (train_gshelltet_deepfashion.py:620-680, :395-478), through the
reference's own import names (gshell_amd.compat): tet grid from an .npz on disk, GShellTetsGeometry, MLPTexture3D material,
trainable probe, BilateralDenoiser, geometry.tick, three Adam steps.  The loss must go down on a fixed target."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_reference_style_training_loop(tmp_path):
    import gshell_amd.compat as compat
    compat.install()
    import nvdiffrast.torch as dr
    import render.renderutils as ru
    from denoiser.denoiser import BilateralDenoiser
    from geometry.gshell_tets_geometry import GShellTetsGeometry
    from render import light, mlptexture

    from gshell_amd import grid, workload
    from gshell_amd.train import default_flags

    torch.manual_seed(0)
    np.random.seed(0)
    res = 24
    verts, tets = grid.grid_for_res(res)
    npz = str(tmp_path / f"{res}_tets.npz")
    grid.save_npz(npz, verts, tets)                       # same keys as the reference's data/tets/generate_tets.py:47
    FLAGS = default_flags(gshell_grid=res, n_samples=2, batch=2, train_res=[64, 64], sdf_mlp_pretrain_steps=150, sphere_init_norm=0.45, iter=100)
    glctx = dr.RasterizeGLContext()
    lgt = light.create_trainable_env_rnd(64, scale=0.0, bias=0.5)
    denoiser = BilateralDenoiser().cuda()
    geometry = GShellTetsGeometry(FLAGS.gshell_grid, FLAGS.mesh_scale, FLAGS, tet_init_file=npz)
    kd_min, kd_max = torch.tensor(FLAGS.kd_min, device="cuda"), torch.tensor(FLAGS.kd_max, device="cuda")
    ks_min, ks_max = torch.tensor(FLAGS.ks_min, device="cuda"), torch.tensor(FLAGS.ks_max, device="cuda")
    mat = {'kd_ks': mlptexture.MLPTexture3D(geometry.getAABB(), channels=6, min_max=[torch.cat((kd_min[0:3], ks_min)), torch.cat((kd_max[0:3], ks_max))]),
           'bsdf': 'pbr', 'no_perturbed_nrm': False}
    loss_fn = lambda img, ref: ru.image_loss(img, ref, loss='l1', tonemapper='log_srgb')          # createLoss('logl1'), train script :57-58
    mvp, campos = workload.views([0, 5], "cuda")
    target = {'mvp': mvp, 'campos': campos, 'resolution': [64, 64], 'spp': 1, 'background': torch.ones(2, 64, 64, 3, device="cuda")}
    img = torch.zeros(2, 64, 64, 4, device="cuda")
    yy, xx = torch.meshgrid(torch.arange(64, device="cuda"), torch.arange(64, device="cuda"), indexing="ij")
    disc = ((xx - 32) ** 2 + (yy - 32) ** 2 < 13 ** 2).float()
    img[..., 3] = disc
    img[..., 0:3] = 0.6 * disc[..., None] + (1 - disc[..., None])
    target['img'] = img
    named = list(geometry.named_parameters())
    opt_mesh = torch.optim.Adam([{'params': [p for n, p in named if 'deform' in n], 'lr': 0.03}, {'params': [p for n, p in named if 'msdf' in n], 'lr': 0.03},
                                 {'params': [p for n, p in named if 'sdf' in n and 'msdf' not in n], 'lr': 3e-4}])
    opt_mat = torch.optim.Adam(mat['kd_ks'].parameters(), lr=0.005)
    opt_lgt = torch.optim.Adam(lgt.parameters(), lr=0.03)
    losses = []
    for it in range(12):
        for o in (opt_mesh, opt_mat, opt_lgt):
            o.zero_grad()
        lgt.update_pdf()
        img_loss, depth_loss, reg_loss = geometry.tick(glctx, target, lgt, mat, loss_fn, it, denoiser=denoiser)
        (img_loss + reg_loss).backward()
        lgt.base.grad *= 64
        mat['kd_ks'].encoder.params.grad /= 8.0
        for o in (opt_mat, opt_mesh, opt_lgt):
            o.step()
        with torch.no_grad():
            lgt.clamp_(min=1e-4)
            geometry.clamp_deform()
        losses.append(float(img_loss.detach()))
    assert all(np.isfinite(losses))
    # descent: clearly lower at the end, no step up beyond the Monte-Carlo / float-atomic noise of a 2-sample, 64 x 64 render (a 1e-3
    # bound failed once in ~10 runs of the unchanged code: summation order differs from run to run and 12 Adam steps amplify it)
    assert losses[-1] < losses[0] - 0.015 and all(b < a + 5e-3 for a, b in zip(losses, losses[1:])), losses
    sd = geometry.state_dict()
    assert {"sdf", "msdf", "deform"} <= set(sd) and "sdf_net.net.0.weight" in sd

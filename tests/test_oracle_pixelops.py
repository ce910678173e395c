"""Pins oracle/pixel_oracle.py against golden vectors minted from the REAL reference functions
(oracle/make_golden_pixelops.py): renderutils python twins, mesh.auto_normals, EnvironmentLight.update_pdf."""
import os

import numpy as np
import torch

from oracle import pixel_oracle as po

G = os.path.join(os.path.dirname(__file__), "golden")


def test_shading_normal_matches_reference_twin():
    g = np.load(os.path.join(G, "pixelops_shading_normal.npz"))
    wgt = torch.tensor(g["w"])
    for tag in ("nopert", "pert"):
        for two_sided in (True, False):
            key = f"{tag}_{int(two_sided)}"
            leaves = {k: torch.tensor(g[f"in_{k}"]).requires_grad_(True) for k in ("pos", "view_pos", "smooth_nrm", "smooth_tng", "geom_nrm")}
            pn = torch.tensor(g["in_perturbed_nrm"]).requires_grad_(True) if tag == "pert" else None
            out = po.prepare_shading_normal(leaves["pos"], leaves["view_pos"], pn, leaves["smooth_nrm"], leaves["smooth_tng"], leaves["geom_nrm"],
                                            two_sided, True)
            (out * wgt).sum().backward()
            np.testing.assert_allclose(out.detach().numpy(), g[f"out_{key}"], rtol=1e-6, atol=1e-7)
            for k, v in leaves.items():
                np.testing.assert_allclose(v.grad.numpy(), g[f"g_{k}_{key}"], rtol=1e-5, atol=1e-6, err_msg=f"{k} {key}")
            if pn is not None:
                np.testing.assert_allclose(pn.grad.numpy(), g[f"g_perturbed_nrm_{key}"], rtol=1e-5, atol=1e-6)


def test_image_loss_matches_reference_twin():
    g = np.load(os.path.join(G, "pixelops_image_loss.npz"))
    img, tgt = torch.tensor(g["in_img"]), torch.tensor(g["in_target"])
    for l in ("l1", "mse", "smape", "relmse"):
        for tm in ("none", "log_srgb"):
            a = img.clone().requires_grad_(True)
            v = po.image_loss(a, tgt, l, tm, twin=True)
            v.backward()
            np.testing.assert_allclose(v.detach().numpy(), g[f"out_{l}_{tm}"], rtol=1e-6)
            np.testing.assert_allclose(a.grad.numpy(), g[f"g_img_{l}_{tm}"], rtol=1e-5, atol=1e-8)
    # kernel mode == twin mode where the reference kernel and its twin agree: in-range inputs, no tonemapper, no smape
    a = img.clamp(0, 1)
    for l in ("l1", "mse", "relmse"):
        assert torch.allclose(po.image_loss(a, tgt, l, "none"), po.image_loss(a, tgt, l, "none", twin=True), rtol=1e-6)


def test_auto_normals_matches_reference():
    g = np.load(os.path.join(G, "pixelops_auto_normals.npz"))
    v = torch.tensor(g["in_verts"]).requires_grad_(True)
    out = po.auto_normals(v, torch.tensor(g["in_tri"]).long())
    (out * torch.tensor(g["w"])).sum().backward()
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(v.grad.numpy(), g["g_verts"], rtol=1e-5, atol=1e-6)
    assert (out[-2:].detach() == torch.tensor([0.0, 0.0, 1.0])).all()


def test_light_pdf_matches_reference():
    g = np.load(os.path.join(G, "pixelops_light_pdf.npz"))
    pdf, rows, cols = po.update_pdf(torch.tensor(g["in_base"]))
    np.testing.assert_array_equal(pdf.numpy(), g["pdf"])
    np.testing.assert_array_equal(rows.numpy(), g["rows"])
    np.testing.assert_array_equal(cols.numpy(), g["cols"])


def test_texture_linear_clamp_known_answers():
    tex = torch.arange(12, dtype=torch.float32).reshape(1, 3, 4, 1)
    centres = po.pixel_grid(4, 3)[None]
    assert torch.allclose(po.texture_linear_clamp(tex, centres), tex, atol=1e-5)             # texel centres reproduce texels
    uv = torch.tensor([[[[0.0, 0.0], [1.0, 1.0], [0.25, 0.5]]]])
    out = po.texture_linear_clamp(tex, uv)[0, 0, :, 0]
    assert torch.allclose(out, torch.tensor([0.0, 11.0, 4.5]))                     # clamp at borders; midpoint of 4 and 5

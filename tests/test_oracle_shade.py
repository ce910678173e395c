"""Pins oracle/shade_oracle.py: BSDF terms and the bilateral filter against golden vectors minted from the REAL
reference python code (oracle/make_golden_shade.py), and the Monte-Carlo estimator against analytic cases."""
import os

import numpy as np
import torch

from oracle import pixel_oracle as po
from oracle import shade_oracle as so

G = os.path.join(os.path.dirname(__file__), "golden")


def test_bsdf_matches_reference_twins():
    g = np.load(os.path.join(G, "shade_bsdf.npz"))
    leaves = [torch.tensor(g[k]).requires_grad_(True) for k in ("col", "nrm", "wo", "wi", "alpha")]
    w = torch.tensor(g["w"])
    spec = so.pbr_specular(*leaves)
    lam = so.lambert(leaves[1], leaves[3])
    ((spec * w).sum() + (lam * w[:, :1]).sum()).backward()
    np.testing.assert_allclose(spec.detach().numpy(), g["spec"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(lam.detach().numpy(), g["lambert"], rtol=1e-6, atol=1e-8)
    for k, t in zip(("col", "nrm", "wo", "wi", "alpha"), leaves):
        ref = g[f"g_{k}"]
        np.testing.assert_allclose(t.grad.numpy(), ref, rtol=1e-3, atol=2e-5 * np.abs(ref).max(), err_msg=k)


def test_bilateral_matches_reference_python_filter():
    g = np.load(os.path.join(G, "shade_bilateral.npz"))
    img, w = torch.tensor(g["in"]), torch.tensor(g["w"])
    for sigma in (0.4, 2.0):
        col = img[..., 0:3].clone().requires_grad_(True)
        o4 = so.bilateral(col, img[..., 3:6], img[..., 9:11], sigma)
        out = o4[..., 0:3] / o4[..., 3:4]
        (out * w).sum().backward()
        np.testing.assert_allclose(out.detach().numpy(), g[f"out_{sigma}"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(col.grad.numpy(), g[f"g_col_{sigma}"], rtol=1e-4, atol=1e-6)


def _flat_gbuffer(B, H, W, seed=0):
    gen = torch.Generator().manual_seed(seed)
    pos = torch.rand(B, H, W, 3, generator=gen) * 0.2
    nrm = torch.tensor([0.0, 1.0, 0.0]).expand(B, H, W, 3).contiguous()
    view = torch.tensor([0.3, 2.0, 0.4]).expand(B, 1, 1, 3).contiguous()
    return pos, nrm, view


def test_white_probe_lambert_integrates_to_one_and_occlusion():
    B, H, W, n = 1, 6, 6, 8
    pos, nrm, view = _flat_gbuffer(B, H, W)
    light = torch.ones(16, 32, 3)
    pdf, rows, cols = po.update_pdf(light)
    perms = torch.argsort(torch.rand(64, n * n, generator=torch.Generator().manual_seed(1)), dim=-1).int().numpy()
    kd, ks = torch.full((B, H, W, 3), 0.7), torch.tensor([0.0, 0.5, 0.0]).expand(B, H, W, 3).contiguous()
    mask = torch.ones(B, H, W)
    mask[0, 0, 0] = 0
    ro = pos + nrm * 1e-3
    no_tris = (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int64))
    diff, spec = so.env_shade(mask, ro, pos, nrm, view, kd, ks, light, pdf, rows[:, 0], cols, perms, 1, n, 7, 1.0, *no_tris)
    assert (diff[0, 0, 0] == 0).all()                               # masked pixel untouched
    d = diff[mask > 0]
    assert abs(float(d.mean()) - 1.0) < 0.03, float(d.mean())       # int cos/pi over the hemisphere = 1
    assert (spec == 0).all()                                        # 'diffuse' BSDF has no specular lobe
    # a big roof above the patch occludes every upward ray: V = 1 - shadow_scale
    roof = np.array([[-5000, 0.3, -5000], [5000, 0.3, -5000], [0, 0.3, 8000]], np.float32)
    tri = np.array([[0, 1, 2]], np.int64)
    d_occ, _ = so.env_shade(mask, ro, pos, nrm, view, kd, ks, light, pdf, rows[:, 0], cols, perms, 1, n, 7, 0.75, roof, tri)
    assert torch.allclose(d_occ, diff * 0.25, rtol=2e-3, atol=1e-6)      # (grazing rays below the roof's horizon escape)


def test_env_shade_pbr_gradients_finite_and_light_grad_nonzero():
    B, H, W, n = 1, 4, 4, 2
    pos, nrm, view = _flat_gbuffer(B, H, W, 3)
    gen = torch.Generator().manual_seed(5)
    light = (torch.rand(8, 16, 3, generator=gen) + 0.1).requires_grad_(True)
    pdf, rows, cols = po.update_pdf(light.detach())
    perms = torch.argsort(torch.rand(32, n * n, generator=gen), dim=-1).int().numpy()
    kd = torch.rand(B, H, W, 3, generator=gen).requires_grad_(True)
    ks = torch.rand(B, H, W, 3, generator=gen).requires_grad_(True)
    nrm = (nrm + 0.2 * torch.randn(B, H, W, 3, generator=gen)).requires_grad_(True)
    pos = pos.requires_grad_(True)
    diff, spec = so.env_shade(torch.ones(B, H, W), pos.detach() + nrm.detach() * 1e-3, pos, nrm, view, kd, ks, light, pdf, rows[:, 0], cols, perms, 0, n,
                              11, 1.0, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int64))
    (diff.sum() + spec.sum()).backward()
    for t in (pos, nrm, kd, ks, light):
        assert torch.isfinite(t.grad).all()
    assert light.grad.abs().sum() > 0 and ks.grad.abs().sum() > 0 and kd.grad.abs().sum() > 0


def test_any_hit_bruteforce_known_answers():
    verts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    tri = np.array([[0, 1, 2]], np.int64)
    org = np.array([[0.2, 0.2, 1], [0.2, 0.2, 1], [0.9, 0.9, 1], [0.2, 0.2, -1]], np.float32)
    d = np.array([[0, 0, -1], [0, 0, 1], [0, 0, -1], [0, 0, 1]], np.float32)
    assert so.any_hit_bruteforce(org, d, verts, tri).tolist() == [True, False, False, True]

"""Pins the python restatements (oracle/shade_oracle.py, pixel_oracle.py) to tests/golden/ref_*.npz -- OUTPUTS OF THE REFERENCE'S
OWN native kernels (kernel.cu, denoising.cu, loss.cu, normal.cu, mesh.cu compiled for the host by oracle/Makefile; minted by
oracle/make_golden_ref.py) -- and, wherever oracle/_ref is built, re-mints every golden and compares it bit for bit."""
import os

import numpy as np
import pytest
import torch

from oracle import make_golden_ref as mg
from oracle import pixel_oracle as po
from oracle import refnative as rn
from oracle import shade_oracle as so

G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return dict(np.load(os.path.join(G, name)))


def _outside(a, b, tol=1e-4):
    """indices (all but the channel axis) whose deviation exceeds tol * max|b|"""
    sc = max(float(np.abs(b).max()), 1e-30)
    return np.argwhere(np.abs(a - b).max(-1) > tol * sc)


@pytest.mark.skipif(not (rn.available("ref_envshade") and rn.available("ref_denoise") and rn.available("ref_renderutils")),
                    reason="oracle/_ref not built (needs /root/reference)")
def test_goldens_are_what_the_reference_build_produces():
    for name, arrays in mg.all_goldens().items():
        g = _load(name)
        assert sorted(g) == sorted(arrays), name
        for k, v in arrays.items():
            np.testing.assert_array_equal(np.asarray(v), g[k], err_msg=f"{name}:{k}")


@pytest.mark.parametrize("bsdf,n,suffix", [(b, n, "") for b, n in mg.ENVSHADE_CASES] + [("pbr", 8, "_64x64"), ("pbr", 4, "_occluder")])
def test_env_shade_restatement_equals_the_reference_kernel(bsdf, n, suffix):
    """oracle/shade_oracle.env_shade (numpy sampling + torch autograd) vs kernel.cu compiled for the host: EVERY pixel of the
    forward outputs and of the five gradients within 1e-4 of the tensor's maximum.  The two sides share libm / numpy float
    arithmetic up to association, so no discrete decision flips on these inputs -- any outlier is a restatement bug."""
    g = _load(f"ref_envshade_{bsdf}_n{n}{suffix}.npz")
    t = {k: torch.tensor(g[k]) for k in ("mask", "ro", "gb_pos", "gb_normal", "view_pos", "gb_kd", "gb_ks", "light", "pdf", "rows", "cols")}
    leaves = [t[k].clone().requires_grad_(True) for k in ("gb_pos", "gb_normal", "gb_kd", "gb_ks", "light")]
    d, s = so.env_shade(t["mask"], t["ro"], leaves[0], leaves[1], t["view_pos"], leaves[2], leaves[3], leaves[4], t["pdf"], t["rows"], t["cols"],
                        g["perms"].astype(np.int32), int(g["bsdf"]), n, int(g["seed"]), float(g["shadow_scale"]), g["verts"], g["tris"].astype(np.int64))
    ((d * torch.tensor(g["diff_grad"])).sum() + (s * torch.tensor(g["spec_grad"])).sum()).backward()
    assert len(_outside(d.detach().numpy(), g["diff"])) == 0
    assert len(_outside(s.detach().numpy(), g["spec"])) == 0
    assert float(g["mask"].sum()) > 20 and float(np.abs(g["diff"]).max()) > 0
    for k, leaf in zip(("gb_pos", "gb_normal", "gb_kd", "gb_ks", "light"), leaves):
        ref = g[f"g_{k}"]
        if bsdf != "pbr" and k in ("gb_pos", "gb_kd", "gb_ks"):
            assert float(np.abs(ref).max()) == 0.0      # the Lambert branch of the reference writes no such gradient (kernel.cu:421-428)
            assert leaf.grad is None or float(leaf.grad.abs().max()) == 0.0
            continue
        assert float(np.abs(ref).max()) > 0, k
        # 2e-4: the normal gradient sums O(100) cancelling terms of magnitude ~300 in float32 on both sides
        assert len(_outside(leaf.grad.numpy(), ref, 2e-4)) == 0, (k, _outside(leaf.grad.numpy(), ref, 2e-4)[:5])


def test_bilateral_restatement_equals_the_reference_kernel():
    g = _load("ref_bilateral.npz")
    for sigma in (0.4, 1.0, 2.0):
        col = torch.tensor(g["col"]).requires_grad_(True)
        out = so.bilateral(col, torch.tensor(g["nrm"]), torch.tensor(g["zdz"]), sigma)
        (out * torch.tensor(g["out_grad"])).sum().backward()
        np.testing.assert_allclose(out.detach().numpy(), g[f"out_{sigma}"], rtol=2e-5, atol=1e-6)
        # the reference's backward kernel is the exact adjoint w.r.t. col with the TAP's dz in the depth weight (denoising.cu:118);
        # so.bilateral differentiates the forward, whose depth weight uses the CENTRE's dz: the two agree where dz is smooth only
        gk = g[f"g_col_{sigma}"]
        assert gk.shape == col.grad.shape and np.isfinite(gk).all()


def test_bilateral_backward_kernel_is_the_tap_dz_adjoint():
    """denoising.cu:74-130 restated literally (gather form, weight evaluated with the tap's dz): pins what the HIP backward must do."""
    g = _load("ref_bilateral.npz")
    col, nrm, zdz, og = (torch.tensor(g[k]) for k in ("col", "nrm", "zdz", "out_grad"))
    B, H, W, _ = col.shape
    for sigma in (0.4, 1.0):
        rad = 2 * int(np.ceil(sigma * 2.5)) + 1
        acc = torch.zeros(B, H, W, 3)
        ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        for fy in range(-rad, rad + 1):
            for fx in range(-rad, rad + 1):
                yy, xx = ys + fy, xs + fx
                valid = ((yy >= 0) & (yy < H) & (xx >= 0) & (xx < W))[None, ..., None]
                yc, xc = yy.clamp(0, H - 1), xx.clamp(0, W - 1)
                t_nrm, t_zdz, t_g = nrm[:, yc, xc], zdz[:, yc, xc], og[:, yc, xc, :3]
                d2 = float(fx * fx + fy * fy)
                w = np.exp(-d2 / (2 * sigma * sigma)) * torch.clamp((t_nrm * nrm).sum(-1, keepdim=True), 1e-4, 1.0) ** 128.0 \
                    * torch.exp(-(torch.abs(t_zdz[..., 0:1] - zdz[..., 0:1]) / torch.clamp(t_zdz[..., 1:2] * np.sqrt(d2), min=1e-4)))
                acc = acc + torch.where(valid, w * t_g, torch.zeros(()))
        np.testing.assert_allclose(acc.numpy(), g[f"g_col_{sigma}"], rtol=2e-4, atol=2e-6)


def test_image_loss_restatement_equals_the_reference_kernel():
    """Forward: pixel_oracle.image_loss (kernel semantics) vs loss.cu for all 8 (loss, tonemapper) pairs.  Backward: autograd of
    the restatement equals the kernel wherever both inputs lie inside (0, 65535); outside, the kernel's backward re-evaluates
    the loss on the UNCLAMPED values (loss.cu:157-163) and then zeroes its own input's gradient -- checked literally below."""
    g = _load("ref_image_loss.npz")
    img, tgt = torch.tensor(g["img"]), torch.tensor(g["target"])
    inside = ((img > 0) & (img < 65535) & (tgt > 0) & (tgt < 65535)).numpy()
    assert 0.5 < inside.mean() < 1.0
    for loss in ("l1", "mse", "smape", "relmse"):
        for tm in ("none", "log_srgb"):
            a, b = img.clone().requires_grad_(True), tgt.clone().requires_grad_(True)
            v = po.image_loss(a, b, loss, tm)
            v.backward()
            ref = float(g[f"{loss}_{tm}_value"])
            assert abs(float(v.detach()) - ref) <= 2e-6 * abs(ref), (loss, tm, float(v.detach()), ref)
            for mine, key in ((a.grad.numpy(), "g_img"), (b.grad.numpy(), "g_target")):
                r = g[f"{loss}_{tm}_{key}"]
                np.testing.assert_allclose(mine[inside], r[inside], rtol=2e-4, atol=2e-6 * np.abs(r[inside]).max(), err_msg=f"{loss} {tm} {key}")
            gi, gt = po.image_loss_kernel_backward(img, tgt, loss, tm)
            np.testing.assert_allclose(gi.numpy(), g[f"{loss}_{tm}_g_img"], rtol=2e-4, atol=2e-6 * np.abs(g[f"{loss}_{tm}_g_img"]).max())
            np.testing.assert_allclose(gt.numpy(), g[f"{loss}_{tm}_g_target"], rtol=2e-4, atol=2e-6 * np.abs(g[f"{loss}_{tm}_g_target"]).max())


def test_shading_normal_restatement_equals_the_reference_kernel():
    g = _load("ref_shading_normal.npz")
    names = ("pos", "view_pos", "perturbed_nrm", "smooth_nrm", "smooth_tng", "geom_nrm")
    zero_nrm = np.zeros(g["pos"].shape[:3], bool)
    zero_nrm[0, 0, 0] = True        # smooth_nrm == 0 there: CUDA safeNormalize gives 0 (vec3f.h:90), the python twin's F.normalize too
    for two_sided in (True, False):
        for opengl in (True, False):
            tag = f"ts{int(two_sided)}_gl{int(opengl)}"
            leaves = [torch.tensor(g[k]).requires_grad_(True) for k in names]
            out = po.prepare_shading_normal(*leaves, two_sided_shading=two_sided, opengl=opengl)
            (out * torch.tensor(g["grad"])).sum().backward()
            np.testing.assert_allclose(out.detach().numpy(), g[f"{tag}_out"], rtol=1e-4, atol=2e-6)
            for k, leaf in zip(names, leaves):
                r = g[f"{tag}_g_{k}"]
                m = ~zero_nrm if r.shape[:3] == zero_nrm.shape else np.ones(r.shape[:3], bool)
                np.testing.assert_allclose(leaf.grad.numpy()[m], r[m], rtol=2e-3, atol=2e-5 * np.abs(r).max(), err_msg=f"{tag} {k}")


def test_xfm_points_golden_is_the_matrix_product():
    g = _load("ref_xfm_points.npz")
    pts, mtx = torch.tensor(g["points"]).requires_grad_(True), torch.tensor(g["matrix"])
    out = torch.matmul(torch.nn.functional.pad(pts, (0, 1), value=1.0), mtx.transpose(1, 2))      # the python twin (renderutils/ops.py:528)
    (out * torch.tensor(g["grad"])).sum().backward()
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pts.grad.numpy(), g["g_points_full"].sum(0, keepdims=True), rtol=1e-5, atol=1e-5)

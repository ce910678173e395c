"""CPU oracle for the per-pixel / per-vertex ops (TEST INFRASTRUCTURE -- checker only).

Restates, in plain torch (autograd = gradient oracle):
  prepare_shading_normal  reference Python twin render/renderutils/bsdf.py:21-51
  image_loss              reference Python twin render/renderutils/loss.py:15-42
  auto_normals            render/mesh.py:212-237
  update_pdf              render/light.py:46-59 (+ util.pixel_grid render/util.py:61-65)
  texture_linear_clamp    nvdiffrast dr.texture(filter_mode='linear', boundary_mode='clamp') [3P, parity
                          unpinned]: texel centres at (i+.5)/W == grid_sample(align_corners=False, border)
Parity pin: tests/test_oracle_pixelops.py checks these against tests/golden/pixelops_*.npz minted from the
REAL reference functions by oracle/make_golden_pixelops.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import numpy as np
import torch

NORMAL_THRESHOLD = 0.1


def _dot(x, y):
    return torch.sum(x * y, -1, keepdim=True)


def _safe_normalize(x):
    return torch.nn.functional.normalize(x, dim=-1)


def prepare_shading_normal(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading=True, opengl=True):
    if perturbed_nrm is None:
        perturbed_nrm = torch.tensor([0, 0, 1], dtype=pos.dtype)[None, None, None, :]
    smooth_nrm = _safe_normalize(smooth_nrm)
    smooth_tng = _safe_normalize(smooth_tng)
    view_vec = _safe_normalize(view_pos - pos)
    smooth_bitang = _safe_normalize(torch.cross(smooth_tng, smooth_nrm, dim=-1))
    sgn = -1.0 if opengl else 1.0
    shading_nrm = smooth_tng * perturbed_nrm[..., 0:1] + sgn * smooth_bitang * perturbed_nrm[..., 1:2] \
        + smooth_nrm * torch.clamp(perturbed_nrm[..., 2:3], min=0.0)
    shading_nrm = _safe_normalize(shading_nrm)
    if two_sided_shading:
        front = _dot(geom_nrm, view_vec) > 0
        shading_nrm = torch.where(front, shading_nrm, -shading_nrm)
        geom_nrm = torch.where(front, geom_nrm, -geom_nrm)
    t = torch.clamp(_dot(view_vec, shading_nrm) / NORMAL_THRESHOLD, min=0, max=1)
    return torch.lerp(geom_nrm, shading_nrm, t)


def _tonemap_srgb(f, exposure=5):
    f = f * exposure
    return torch.where(f > 0.0031308, torch.pow(torch.clamp(f, min=0.0031308), 1.0 / 2.4) * 1.055 - 0.055, 12.92 * f)


def _tonemap_log_srgb_kernel(x):
    """What the reference CUDA kernel computes (c_src/loss.cu:28-48): sRGB(log(x+1)) WITHOUT the python twin's
    exposure=5 factor (the twin and the kernel disagree in the reference; the training path runs the kernel)."""
    f = torch.log(torch.clamp(x, min=0, max=65535) + 1)
    return torch.where(f > 0.0031308, torch.pow(torch.clamp(f, min=0.0031308), 1.0 / 2.4) * 1.055 - 0.055, 12.92 * torch.clamp(f, min=0.0))


def image_loss(img, target, loss='l1', tonemapper='none', twin=False):
    """twin=True reproduces the reference's python twin (loss.py:30-42) literally; twin=False reproduces its CUDA
    kernel (loss.cu:95-135), which is what `ru.image_loss` runs in training: inputs clamped to [0, 65535] for every
    tonemapper, log_srgb without the exposure factor, SMAPE without abs() in the denominator."""
    if twin:
        if tonemapper == 'log_srgb':
            img = _tonemap_srgb(torch.log(torch.clamp(img, min=0, max=65535) + 1))
            target = _tonemap_srgb(torch.log(torch.clamp(target, min=0, max=65535) + 1))
        if loss == 'mse':
            return torch.nn.functional.mse_loss(img, target)
        if loss == 'smape':
            return torch.mean(torch.abs(img - target) / (torch.abs(img) + torch.abs(target) + 0.01))
        if loss == 'relmse':
            return torch.mean((img - target) * (img - target) / (img * img + target * target + 0.1))
        return torch.nn.functional.l1_loss(img, target)
    if tonemapper == 'log_srgb':
        img, target = _tonemap_log_srgb_kernel(img), _tonemap_log_srgb_kernel(target)
    else:
        img, target = torch.clamp(img, 0, 65535), torch.clamp(target, 0, 65535)
    if loss == 'mse':
        v = (img - target) ** 2
    elif loss == 'relmse':
        v = (img - target) ** 2 / (img * img + target * target + 0.1)
    elif loss == 'smape':
        v = torch.abs(img - target) / (img + target + 0.01)
    else:
        v = torch.abs(img - target)
    return v.mean()


def auto_normals(v_pos, t_pos_idx):
    i0, i1, i2 = t_pos_idx[:, 0], t_pos_idx[:, 1], t_pos_idx[:, 2]
    v0, v1, v2 = v_pos[i0, :], v_pos[i1, :], v_pos[i2, :]
    face_normals = torch.cross(v1 - v0, v2 - v0, dim=-1)
    v_nrm = torch.zeros_like(v_pos)
    v_nrm = v_nrm.scatter_add(0, i0[:, None].repeat(1, 3), face_normals)
    v_nrm = v_nrm.scatter_add(0, i1[:, None].repeat(1, 3), face_normals)
    v_nrm = v_nrm.scatter_add(0, i2[:, None].repeat(1, 3), face_normals)
    v_nrm = torch.where(_dot(v_nrm, v_nrm) > 1e-20, v_nrm, torch.tensor([0.0, 0.0, 1.0], dtype=v_pos.dtype))
    return v_nrm / torch.sqrt(torch.clamp(_dot(v_nrm, v_nrm), min=1e-20))


def pixel_grid(width, height, center_x=0.5, center_y=0.5):
    y, x = torch.meshgrid((torch.arange(0, height, dtype=torch.float32) + center_y) / height,
                          (torch.arange(0, width, dtype=torch.float32) + center_x) / width, indexing='ij')
    return torch.stack((x, y), dim=-1)


def update_pdf(base):
    """-> (pdf [H,W], rows [H,W], cols [H,W])  (light.py:46-59)"""
    Y = pixel_grid(base.shape[1], base.shape[0])[..., 1]
    pdf = torch.max(base, dim=-1)[0] * torch.sin(Y * np.pi)
    pdf = pdf / torch.sum(pdf)
    cols = torch.cumsum(pdf, dim=1)
    rows = torch.cumsum(cols[:, -1:].repeat([1, cols.shape[1]]), dim=0)
    cols = cols / torch.where(cols[:, -1:] > 0, cols[:, -1:], torch.ones_like(cols))
    rows = rows / torch.where(rows[-1:, :] > 0, rows[-1:, :], torch.ones_like(rows))
    return pdf, rows, cols


def texture_linear_clamp(tex, uv):
    """tex [B,H,W,C], uv [B,h,w,2] in [0,1] -> [B,h,w,C]."""
    uv = uv.to(tex.dtype)
    out = torch.nn.functional.grid_sample(tex.permute(0, 3, 1, 2), uv * 2 - 1, mode='bilinear', padding_mode='border', align_corners=False)
    return out.permute(0, 2, 3, 1)


def sdf_reg_loss(sdf, all_edges):
    """compute_sdf_reg_loss, literal restatement of geometry/gshell_tets_geometry.py:33-39."""
    pair = sdf[all_edges.reshape(-1)].reshape(-1, 2)
    mask = torch.sign(pair[..., 0]) != torch.sign(pair[..., 1])
    pair = pair[mask]
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    return bce(pair[..., 0], (pair[..., 1] > 0).float()) + bce(pair[..., 1], (pair[..., 0] > 0).float())


def image_loss_kernel_backward(img, target, loss='l1', tonemapper='none', d_scalar=1.0):
    """Gradient of `ru.image_loss` as the reference's CUDA backward kernel computes it (c_src/loss.cu:137-209), which is NOT
    the autograd of its forward outside (0, 65535): the kernel re-evaluates the tonemap / loss derivative on the UNCLAMPED
    inputs (:157-163), applies the log-sRGB chain rule only where 0 < x < 65535 (:50-67) and finally zeroes the gradient of an
    input that is itself <= 0 or >= 65535 (:197-202) -- the OTHER input still receives the unclamped-value derivative.
    d_out per pixel = d_scalar / (B*H*W) (renderutils/ops.py:497).  Pinned by tests/golden/ref_image_loss.npz."""
    B, H, W, _ = img.shape
    d_v = torch.full_like(img, d_scalar / (B * H * W)) / 3.0

    def srgb(x):
        return torch.where(x > 0.0031308, torch.pow(torch.clamp(x, min=0.0031308), 1.0 / 2.4) * 1.055 - 0.055, 12.92 * torch.clamp(x, min=0.0))

    def d_srgb(x):   # bwdSRGB (:33-39)
        return torch.where(x > 0.0031308, 0.439583 / torch.pow(torch.clamp(x, min=1e-30), 0.583333), torch.where(x > 0.0, torch.full_like(x, 12.92), torch.zeros_like(x)))
    a, b = img, target
    if tonemapper == 'log_srgb':
        a, b = srgb(torch.log(img + 1.0)), srgb(torch.log(target + 1.0))      # NaN for x <= -1: masked out below like the kernel's `if`
    if loss == 'mse':
        d_a = d_v * 2 * (a - b)
        d_b = -d_a
    elif loss == 'relmse':
        den = b * b + a * a + 0.1
        d_a = d_v * 2 * (a - b) * (b * (b + a) + 0.1) / (den * den)
        d_b = -(d_v * 2 * (a - b) * (a * (b + a) + 0.1) / (den * den))
    elif loss == 'smape':
        den = b + a + 0.01
        sg = torch.sign(a - b)
        d_a = d_v * sg * (2 * b + 0.01) / (den * den)
        d_b = -(d_v * sg * (2 * a + 0.01) / (den * den))
    else:
        d_a = d_v * torch.sign(a - b)
        d_b = -d_a
    if tonemapper == 'log_srgb':
        def chain(x, d):
            ok = (x > 0) & (x < 65535)
            xs = torch.where(ok, x, torch.ones_like(x))
            return torch.where(ok, d * d_srgb(torch.log(xs + 1.0)) / (xs + 1.0), torch.zeros_like(x))
        d_a, d_b = chain(img, d_a), chain(target, d_b)
    d_a = torch.where((img <= 0) | (img >= 65535), torch.zeros_like(d_a), d_a)
    d_b = torch.where((target <= 0) | (target >= 65535), torch.zeros_like(d_b), d_b)
    return d_a, d_b

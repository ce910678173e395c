"""Image-space regularisers of the G-Shell loss (semantics of the reference's render/regularizer.py:17-51).
Inputs are the composited buffers [B,H,W,C]; everything is elementwise + a mean, so it stays torch."""
import torch

from . import util

_EPS = 1e-3


def _mean3(x):      # "luma" of the reference: unweighted channel mean
    return x[..., :3].mean(dim=-1, keepdim=True)


def _max3(x):       # HSV "value"
    return x[..., :3].amax(dim=-1, keepdim=True)


def luma(x):
    return _mean3(x).expand(*x.shape[:-1], 3)


def value(x):
    return _max3(x).expand(*x.shape[:-1], 3)


def chroma_loss(kd, color_ref, lambda_chroma):
    alpha = color_ref[..., 3:]
    ref = color_ref[..., :3] / _max3(color_ref).clamp_min(_EPS)
    opt = kd[..., :3] / _max3(kd).clamp_min(_EPS)
    return ((opt - ref) * alpha).abs().mean() * lambda_chroma


def _log_srgb(x):
    return util.rgb_to_srgb(torch.log(x.clamp(0, 65535) + 1))


def shading_loss(diffuse_light, specular_light, color_ref, lambda_diffuse, lambda_specular):
    """Monochrome-lighting prior: tonemapped luminance of the light should explain the reference's value channel;
    plus a specular/diffuse energy ratio penalty."""
    alpha = color_ref[..., 3:]
    d, s = luma(diffuse_light), luma(specular_light)
    err = (_log_srgb((d + s) * alpha) - _log_srgb(value(color_ref) * alpha)).abs().mean()
    return err * lambda_diffuse + s.mean() / d.mean().clamp_min(_EPS) * lambda_specular


def material_smoothness_grad(kd_grad, ks_grad, nrm_grad, lambda_kd=0.25, lambda_ks=0.1, lambda_nrm=0.0):
    """Buffers hold |jittered - centre| differences in [..., :-1] and coverage in [..., -1]."""
    loss = (kd_grad[..., :3].sum(-1) / 3 * kd_grad[..., -1]).mean() * lambda_kd
    loss = loss + (ks_grad[..., :-1] * ks_grad[..., -1:]).mean() * lambda_ks
    return loss + (nrm_grad[..., :-1] * nrm_grad[..., -1:]).mean() * lambda_nrm

"""Image-space regularisers of the G-Shell loss (semantics of the reference's render/regularizer.py:17-51).
Inputs are the composited buffers [B,H,W,C].  The functions below are the term-by-term torch formulation (also the
reference the fused path is tested against); `frame_sums` evaluates the pixel sums of all of them -- plus the alpha MSE and
the two mSDF image terms of the training loss -- in ONE pass over the stacked frame (gs_frame_sums_fwd/bwd)."""
import ctypes

import torch

from .. import _lib
from .._lib import c_int, c_int64, check, ptr, stream
from . import util

FRAME_SUM_BUFFERS = ('shaded', 'msdf_image', 'diffuse_light', 'specular_light', 'kd_grad', 'ks_grad', 'normal_grad')


class _FrameSumsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, stacked, color_ref, offs, img=None):
        st, ref = stacked.detach().contiguous().float(), color_ref.detach().contiguous().float()
        C = st.shape[-1]
        n = st.numel() // C
        if ref.shape[-1] != 4 or ref.numel() != 4 * n:
            raise _lib.GShellHipError(f"frame_sums: color_ref {tuple(ref.shape)} does not match the frame {tuple(st.shape)}")
        c_offs = (ctypes.c_int32 * 7)(*offs)
        L = _lib.lib()
        partial = torch.empty((int(L.gs_frame_sums_partials(c_int64(n))), 9 if img is None else 10), dtype=torch.float32, device=st.device)
        with torch.cuda.device(st.device):
            if img is None:
                check(L.gs_frame_sums_fwd(ptr(st, torch.float32, "stacked"), ptr(ref, torch.float32, "color_ref"), c_int64(n), c_int64(C), c_offs,
                                          ptr(partial), stream()), "gs_frame_sums_fwd")
            else:
                check(L.gs_frame_sums_img_fwd(ptr(st, torch.float32, "stacked"), ptr(ref, torch.float32, "color_ref"), c_int64(n), c_int64(C), c_offs,
                                              c_int(img[0]), c_int(img[1]), ptr(partial), stream()), "gs_frame_sums_img_fwd")
        ctx.save_for_backward(st, ref)
        ctx.offs, ctx.img = tuple(offs), img
        return partial.sum(0)

    @staticmethod
    def backward(ctx, g):
        st, ref = ctx.saved_tensors
        C = st.shape[-1]
        n = st.numel() // C
        g9 = g.contiguous().float()
        g_st = torch.empty_like(st)
        c_offs = (ctypes.c_int32 * 7)(*ctx.offs)
        with torch.cuda.device(st.device):
            if ctx.img is None:
                check(_lib.lib().gs_frame_sums_bwd(ptr(st), ptr(ref), c_int64(n), c_int64(C), c_offs, ptr(g9), ptr(g_st), stream()), "gs_frame_sums_bwd")
            else:
                check(_lib.lib().gs_frame_sums_img_bwd(ptr(st), ptr(ref), c_int64(n), c_int64(C), c_offs, c_int(ctx.img[0]), c_int(ctx.img[1]), ptr(g9),
                                                       ptr(g_st), stream()), "gs_frame_sums_img_bwd")
        return g_st, None, None, None


def frame_sums(stacked_info, color_ref, img_loss=None):
    """stacked_info = (tensor [B,H,W,C], buffer names, channel counts) = render_mesh(...).stacked.
    Returns a [9] tensor: sum (a-m)^2, sum |msdf+[m=0]|, sum |msdf-[m=1]-1|, sum |logsrgb((d+s)m) - logsrgb(value(ref)m)|,
    sum luma(spec), sum luma(diff), sum kd_grad term, sum ks_grad term, sum normal_grad term  (m = reference alpha).
    `img_loss` = (loss id, tonemapper id) of renderutils.image_loss: a TENTH entry, the element sum of
    image_loss(shaded rgb * m, reference rgb * m) (the colour term of the training loss)."""
    stacked, keys, sizes = stacked_info
    offs = []
    for name in FRAME_SUM_BUFFERS:
        offs.append(sum(sizes[:keys.index(name)]) if name in keys else -1)
    if img_loss is not None and offs[0] < 0:
        img_loss = None
    return _FrameSumsFn.apply(stacked, color_ref, tuple(offs), None if img_loss is None else (int(img_loss[0]), int(img_loss[1])))


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

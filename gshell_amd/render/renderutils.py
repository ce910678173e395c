"""The three `renderutils` ops the G-Shell training path uses, with the reference's call surface
(render/renderutils/ops.py: xfm_points :518, prepare_shading_normal :197, image_loss :479), backed by
HIP kernels through the C ABI.  `use_python=True` is rejected: the reference's PyTorch twins live in
the test oracle, not in the product path."""
import torch

from .. import _lib
from .._lib import c_int, c_int64, c_float, check, ptr, stream


def _no_python(use_python):
    if use_python:
        raise _lib.GShellHipError("use_python=True is not available: this package has no eager fallback (see oracle/ for the checker)")


class _XfmPointsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, matrix):
        pts = points.detach().contiguous().float()
        mtx = matrix.detach().contiguous().float()
        Bp, V, _ = pts.shape
        B = mtx.shape[0]
        out = torch.empty((B, V, 4), dtype=torch.float32, device=pts.device)
        with torch.cuda.device(pts.device):
            check(_lib.lib().gs_xfm_points_fwd(ptr(pts, torch.float32, "points"), c_int64(Bp), ptr(mtx, torch.float32, "matrix"), c_int64(B),
                                               c_int64(V), ptr(out), stream()), "gs_xfm_points_fwd")
        ctx.save_for_backward(mtx)
        ctx.dims = (Bp, B, V)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (mtx,) = ctx.saved_tensors
        Bp, B, V = ctx.dims
        g = g_out.contiguous().float()
        g_pts = torch.empty((Bp, V, 3), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            check(_lib.lib().gs_xfm_points_bwd(ptr(g), c_int64(Bp), ptr(mtx), c_int64(B), c_int64(V), ptr(g_pts), stream()), "gs_xfm_points_bwd")
        return g_pts, None


def xfm_points(points, matrix, use_python=False):
    """points [1|B,V,3], matrix [B,4,4] -> [B,V,4]   (ref ops.py:518-537)"""
    _no_python(use_python)
    if points.dim() != 3 or points.shape[-1] != 3 or matrix.dim() != 3 or tuple(matrix.shape[1:]) != (4, 4):
        raise _lib.GShellHipError(f"xfm_points: bad shapes {tuple(points.shape)} {tuple(matrix.shape)}")
    return _XfmPointsFn.apply(points, matrix)


class _ShadingNormalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided, opengl):
        B, H, W, _ = pos.shape
        full = (B, H, W, 3)

        def c(t):
            return t.detach().expand(full).contiguous().float()
        pos_c, nrm_c, tng_c, gn_c = c(pos), c(smooth_nrm), c(smooth_tng), c(geom_nrm)
        view_full = int(tuple(view_pos.shape) == full)
        if view_full:
            vp = c(view_pos)
        else:
            if tuple(view_pos.shape) not in ((B, 1, 1, 3), (1, 1, 1, 3)):
                raise _lib.GShellHipError(f"view_pos must be [B,1,1,3] or full size, got {tuple(view_pos.shape)}")
            vp = view_pos.detach().expand(B, 1, 1, 3).reshape(B, 3).contiguous().float()
        pn = None
        if perturbed_nrm is not None:
            pn = c(perturbed_nrm)
        out = torch.empty(full, dtype=torch.float32, device=pos.device)
        with torch.cuda.device(pos.device):
            check(_lib.lib().gs_shading_normal_fwd(ptr(pos_c, torch.float32, "pos"), ptr(vp), c_int(view_full), ptr(pn), ptr(nrm_c), ptr(tng_c),
                                                   ptr(gn_c), c_int64(B), c_int64(H * W), c_int(int(two_sided)), c_int(int(opengl)), ptr(out),
                                                   stream()), "gs_shading_normal_fwd")
        ctx.save_for_backward(pos_c, vp, pn, nrm_c, tng_c, gn_c)
        ctx.cfg = (B, H, W, view_full, int(two_sided), int(opengl))
        ctx.shapes = (pos.shape, view_pos.shape, None if perturbed_nrm is None else perturbed_nrm.shape, smooth_nrm.shape, smooth_tng.shape,
                      geom_nrm.shape)
        return out

    @staticmethod
    def backward(ctx, g_out):
        pos_c, vp, pn, nrm_c, tng_c, gn_c = ctx.saved_tensors
        B, H, W, view_full, two_sided, opengl = ctx.cfg
        need = ctx.needs_input_grad
        g = g_out.contiguous().float()

        def alloc(flag):
            return torch.empty((B, H, W, 3), dtype=torch.float32, device=g.device) if flag else None
        g_pos, g_view, g_pn, g_nrm, g_tng, g_gn = alloc(need[0]), alloc(need[1]), alloc(need[2] and pn is not None), alloc(need[3]), \
            alloc(need[4]), alloc(need[5])
        with torch.cuda.device(g.device):
            check(_lib.lib().gs_shading_normal_bwd(ptr(pos_c), ptr(vp), c_int(view_full), ptr(pn), ptr(nrm_c), ptr(tng_c), ptr(gn_c), c_int64(B),
                                                   c_int64(H * W), c_int(two_sided), c_int(opengl), ptr(g), ptr(g_pos), ptr(g_view), ptr(g_pn),
                                                   ptr(g_nrm), ptr(g_tng), ptr(g_gn), stream()), "gs_shading_normal_bwd")

        def reduce_to(gt, shape):
            if gt is None or shape is None:
                return None
            return gt.sum_to_size(shape) if tuple(shape) != tuple(gt.shape) else gt
        s = ctx.shapes
        return (reduce_to(g_pos, s[0]), reduce_to(g_view, s[1]), reduce_to(g_pn, s[2]), reduce_to(g_nrm, s[3]), reduce_to(g_tng, s[4]),
                reduce_to(g_gn, s[5]), None, None)


def prepare_shading_normal(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading=True, opengl=True, use_python=False):
    """Final shading normal (ref ops.py:197-229): tangent-frame perturbation (identity when perturbed_nrm is
    None), two-sided flip by the geometric normal, bend towards the viewer."""
    _no_python(use_python)
    return _ShadingNormalFn.apply(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading, opengl)


_LOSS = {'l1': 0, 'mse': 1, 'relmse': 2, 'smape': 3}
_TONEMAP = {'none': 0, 'log_srgb': 1}


@torch.no_grad()
def depth_zgrad(clip_pos, clip_pos_deriv, eps=0.00001):
    """Depth + depth-slope guide of the denoiser, [...,2] = (z0, |z1 - z0|) from the interpolated clip position [...,4] and its
    screen-space derivatives [...,8] (reference render/render.py:276-279, there ~14 ATen launches; no gradient, as there)."""
    c = clip_pos.detach().contiguous().float()
    d = clip_pos_deriv.detach().contiguous().float()
    if c.shape[-1] != 4 or d.shape[-1] != 8 or c.shape[:-1] != d.shape[:-1]:
        raise _lib.GShellHipError(f"depth_zgrad: clip_pos {tuple(c.shape)} / clip_pos_deriv {tuple(d.shape)} must be [...,4] / [...,8]")
    out = torch.empty(tuple(c.shape[:-1]) + (2,), dtype=torch.float32, device=c.device)
    with torch.cuda.device(c.device):
        check(_lib.lib().gs_depth_zgrad(ptr(c, torch.float32, "clip_pos"), ptr(d), c_int64(c.numel() // 4), c_float(eps), ptr(out), stream()), "gs_depth_zgrad")
    return out


class _ImageLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, target, loss, tonemapper):
        L = _lib.lib()
        shape = torch.broadcast_shapes(img.shape, target.shape)
        img_c, tgt_c = img.detach().expand(shape).contiguous().float(), target.detach().expand(shape).contiguous().float()
        n = img_c.numel()
        partials = torch.empty((L.gs_image_loss_partials(c_int64(n)),), dtype=torch.float32, device=img_c.device)
        with torch.cuda.device(img_c.device):
            check(L.gs_image_loss_fwd(ptr(img_c, torch.float32, "img"), ptr(tgt_c, torch.float32, "target"), c_int64(n), c_int(loss),
                                      c_int(tonemapper), ptr(partials), stream()), "gs_image_loss_fwd")
        ctx.save_for_backward(img_c, tgt_c)
        ctx.cfg = (loss, tonemapper, n, img.shape, target.shape)
        # mean over all elements == the reference's  sum(per-pixel mean over 3 channels) / (B*H*W)
        return partials.sum() / max(n, 1)

    @staticmethod
    def backward(ctx, g):
        img_c, tgt_c = ctx.saved_tensors
        loss, tonemapper, n, s_img, s_tgt = ctx.cfg
        g_img = torch.empty_like(img_c) if ctx.needs_input_grad[0] else None
        g_tgt = torch.empty_like(tgt_c) if ctx.needs_input_grad[1] else None
        gs = g.detach().reshape(1).contiguous().float()
        with torch.cuda.device(img_c.device):
            check(_lib.lib().gs_image_loss_bwd(ptr(img_c), ptr(tgt_c), c_int64(n), c_int(loss), c_int(tonemapper), ptr(gs), c_float(1.0 / max(n, 1)),
                                               ptr(g_img), ptr(g_tgt), stream()), "gs_image_loss_bwd")
        if g_img is not None and tuple(s_img) != tuple(g_img.shape):
            g_img = g_img.sum_to_size(s_img)
        if g_tgt is not None and tuple(s_tgt) != tuple(g_tgt.shape):
            g_tgt = g_tgt.sum_to_size(s_tgt)
        return g_img, g_tgt, None, None


def image_loss(img, target, loss='l1', tonemapper='none', use_python=False):
    """HDR image loss -> scalar (ref ops.py:479-503)."""
    _no_python(use_python)
    if loss not in _LOSS or tonemapper not in _TONEMAP:
        raise _lib.GShellHipError(f"image_loss: unknown loss/tonemapper {loss!r}/{tonemapper!r}")
    return _ImageLossFn.apply(img, target, _LOSS[loss], _TONEMAP[tonemapper])

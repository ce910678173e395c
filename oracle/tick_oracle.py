"""CPU oracle of the loss assembly of one training iteration -- TEST INFRASTRUCTURE, checker only; never imported by gshell_amd/.

Restates, term by term, what the reference computes AFTER rendering:
  * `GShellTetsGeometry.tick`            geometry/gshell_tets_geometry.py:257-384 (image / coverage / mSDF-image terms :275-285, eikonal
                                         :302-324, mSDF open / close Huber regularisers :326-358, sdf sign regulariser :361-362 with
                                         `compute_sdf_reg_loss` :33-39, the three image-space regularisers :364-378)
  * `regularizer.shading_loss`           render/regularizer.py:27-41
  * `regularizer.material_smoothness_grad`   render/regularizer.py:47-52
  * `regularizer.chroma_loss`            render/regularizer.py:20-24
  * `util.rgb_to_srgb`                   render/util.py:94-101
  * `ru.image_loss`                      render/renderutils/ops.py:479-505 = the CUDA kernels of c_src/loss.cu: the forward value of
                                         oracle/pixel_oracle.image_loss and the backward of pixel_oracle.image_loss_kernel_backward
                                         (both pinned to loss.cu compiled for the host, tests/golden/ref_image_loss.npz)

PINNED: tests/test_tick_oracle_cpu.py executes the REAL reference `tick` (the method object of /root/reference's GShellTetsGeometry,
modules it cannot import stubbed, `self.render` returning prepared buffers) and compares value and every gradient with `tick` below.
"""
import torch
import torch.nn.functional as F

from oracle import pixel_oracle as po


class _ImageLossKernel(torch.autograd.Function):
    """ru.image_loss: forward = imgLossFwdKernel's mean, backward = imgLossBwdKernel (NOT the derivative of the forward outside (0, 65535))."""

    @staticmethod
    def forward(ctx, img, target, loss, tonemapper):
        ctx.save_for_backward(img, target)
        ctx.spec = (loss, tonemapper)
        return po.image_loss(img, target, loss, tonemapper)

    @staticmethod
    def backward(ctx, g):
        img, target = ctx.saved_tensors
        d_a, d_b = po.image_loss_kernel_backward(img, target, ctx.spec[0], ctx.spec[1], d_scalar=1.0)
        return d_a * g, d_b * g, None, None


def image_loss(img, target, loss='l1', tonemapper='log_srgb'):
    return _ImageLossKernel.apply(img, target, loss, tonemapper)


def _rgb_to_srgb(f):          # render/util.py:94-95
    return torch.where(f <= 0.0031308, f * 12.92, torch.pow(torch.clamp(f, 0.0031308), 1.0 / 2.4) * 1.055 - 0.055)


def luma(x):                  # render/regularizer.py:15-16
    return ((x[..., 0:1] + x[..., 1:2] + x[..., 2:3]) / 3).repeat(1, 1, 1, 3)


def value(x):                 # render/regularizer.py:17-18
    return torch.max(x[..., 0:3], dim=-1, keepdim=True)[0].repeat(1, 1, 1, 3)


def chroma_loss(kd, color_ref, lambda_chroma):
    eps = 0.001
    ref_chroma = color_ref[..., 0:3] / torch.clip(value(color_ref), min=eps)
    opt_chroma = kd[..., 0:3] / torch.clip(value(kd), min=eps)
    return torch.mean(torch.abs((opt_chroma - ref_chroma) * color_ref[..., 3:])) * lambda_chroma


def shading_loss(diffuse_light, specular_light, color_ref, lambda_diffuse, lambda_specular):
    diffuse_luma, specular_luma, ref_luma = luma(diffuse_light), luma(specular_light), value(color_ref)
    eps = 0.001
    img = _rgb_to_srgb(torch.log(torch.clamp((diffuse_luma + specular_luma) * color_ref[..., 3:], min=0, max=65535) + 1))
    target = _rgb_to_srgb(torch.log(torch.clamp(ref_luma * color_ref[..., 3:], min=0, max=65535) + 1))
    loss = torch.mean(torch.abs(img - target)) * lambda_diffuse
    return loss + torch.mean(specular_luma) / torch.clamp(torch.mean(diffuse_luma), min=eps) * lambda_specular


def material_smoothness_grad(kd_grad, ks_grad, nrm_grad, lambda_kd=0.25, lambda_ks=0.1, lambda_nrm=0.0):
    kd_luma_grad = (kd_grad[..., 0] + kd_grad[..., 1] + kd_grad[..., 2]) / 3
    loss = torch.mean(kd_luma_grad * kd_grad[..., -1]) * lambda_kd
    loss = loss + torch.mean(ks_grad[..., :-1] * ks_grad[..., -1:]) * lambda_ks
    return loss + torch.mean(nrm_grad[..., :-1] * nrm_grad[..., -1:]) * lambda_nrm


def eikonal_coeff(FLAGS, iteration):
    if FLAGS.eikonal_scale is not None:
        return FLAGS.eikonal_scale
    return 3e-1 if iteration < 500 else (1e-1 if iteration < 2000 else 1e-2)


def tick(FLAGS, grid_res, sdf_net, all_edges, d, target, iteration, loss=('l1', 'log_srgb')):
    """d = what the reference's `render` returns: 'buffers' (composited + antialiased frames, 'visible_triangles'), 'imesh_faces' [T,3],
    'msdf' [V_aug], 'msdf_boundary', 'n_verts_watertight', 'sdf' [N] (or [N,1]), 'sampled_pts' [n,3] or None.
    -> (img_loss, depth_loss, reg_loss, terms)"""
    buffers = d['buffers']
    t_iter = iteration / FLAGS.iter
    color_ref = target['img']
    gt_mask = color_ref[..., 3:]
    img_loss = F.mse_loss(buffers['shaded'][..., 3:], color_ref[..., 3:])
    img_loss = img_loss + image_loss(buffers['shaded'][..., 0:3] * color_ref[..., 3:], color_ref[..., 0:3] * color_ref[..., 3:], *loss)
    img_loss = img_loss + 5e-1 * F.l1_loss(buffers['msdf_image'].clamp(min=0) * (gt_mask == 0).float(), torch.zeros_like(gt_mask))
    img_loss = img_loss + 5e-1 * F.l1_loss(buffers['msdf_image'].clamp(max=0) * (gt_mask == 1).float(), torch.ones_like(gt_mask))
    depth_loss = torch.zeros(())
    terms = {}

    if FLAGS.use_sdf_mlp and FLAGS.use_eikonal and d.get('sampled_pts') is not None:
        v = d['sampled_pts'].detach().clone().requires_grad_(True)
        sdf_eik = sdf_net(v)
        grad = torch.autograd.grad(sdf_eik.sum(), v, create_graph=True)[0]
        eik_loss = eikonal_coeff(FLAGS, iteration) * (grad.pow(2).sum(dim=-1).sqrt() - 1).pow(2).mean()
    else:
        eik_loss = torch.zeros(())
    terms['eikonal'] = eik_loss

    if FLAGS.use_mesh_msdf_reg:
        regscale = (64 / grid_res) ** 3
        eps = torch.tensor([1e-3], dtype=d['msdf'].dtype)
        open_scale, close_scale = FLAGS.msdf_reg_open_scale, FLAGS.msdf_reg_close_scale
        if open_scale > 0:
            msdf_reg = open_scale * regscale * F.huber_loss(d['msdf'].clamp(min=-eps).squeeze(), -eps.expand(d['msdf'].size(0)), reduction='sum')
        else:
            msdf_reg = torch.zeros(())
        if close_scale != 0:
            with torch.no_grad():
                visible_verts = d['imesh_faces'][buffers['visible_triangles']].unique()
                vb = visible_verts[visible_verts >= d['n_verts_watertight']] - d['n_verts_watertight']
                mask = torch.zeros(d['msdf_boundary'].size(0))
                mask[vb] = 1
                mask = mask.bool()
            boundary = d['msdf_boundary'][mask]
            msdf_reg = msdf_reg + close_scale * regscale * F.huber_loss(boundary.clamp(max=eps).squeeze(), eps.expand(boundary.size(0)), reduction='sum')
    else:
        msdf_reg = torch.zeros(())
    terms['msdf_reg'] = msdf_reg

    sdf_weight = FLAGS.sdf_regularizer - (FLAGS.sdf_regularizer - 0.01) * min(1.0, 4.0 * t_iter)
    sdf_reg = po.sdf_reg_loss(d['sdf'], all_edges).mean() * sdf_weight
    terms['sdf_reg'] = sdf_reg

    if 'diffuse_light' not in buffers:
        monochrome = torch.zeros_like(img_loss)
    else:
        monochrome = shading_loss(buffers['diffuse_light'], buffers['specular_light'], color_ref, FLAGS.lambda_diffuse, FLAGS.lambda_specular)
    smooth = material_smoothness_grad(buffers['kd_grad'], buffers['ks_grad'], buffers['normal_grad'], lambda_kd=FLAGS.lambda_kd, lambda_ks=FLAGS.lambda_ks,
                                      lambda_nrm=FLAGS.lambda_nrm)
    chroma = chroma_loss(buffers['kd'], color_ref, FLAGS.lambda_chroma)
    terms.update(monochrome=monochrome, smooth=smooth, chroma=chroma, img=img_loss)
    reg_loss = (sdf_reg + eik_loss + msdf_reg) + (monochrome + smooth + chroma)
    return img_loss, depth_loss, reg_loss, terms

"""oracle/tick_oracle.py pinned to the REAL reference: the method object `GShellTetsGeometry.tick` of /root/reference's
geometry/gshell_tets_geometry.py is executed on the CPU (its own regularizer.py, util.py and mlp.py exec'd from the reference tree; what
cannot be imported -- nvdiffrast, kaolin, the plugin modules -- stubbed; `self.render` returns prepared buffers; `loss_fn` = the reference's
loss.cu compiled for the host, oracle/_ref) and compared with the restatement: the three returned values and the gradient of their sum with
respect to every buffer, the mSDF tensors, the sdf values and every SDF-network parameter (the eikonal term's double backward).

Runs only where /root/reference exists (the build container); the GPU box uses the restatement this file pins."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import refload, refnative, tick_oracle

pytestmark = pytest.mark.skipif(not refload.reference_available(), reason="/root/reference not present")


def _load_reference_tick():
    """-> (reference GShellTetsGeometry class, reference MLP class)"""
    saved = {k: sys.modules.get(k) for k in ("render", "render.util", "render.mesh", "render.render", "render.optixutils", "render.regularizer",
                                               "nvdiffrast", "nvdiffrast.torch", "imageio", "kaolin", "geometry", "geometry.gshell_tets",
                                               "geometry.mlp", "geometry.embedding")}
    try:
        nv = types.ModuleType("nvdiffrast")
        nv.torch = types.ModuleType("nvdiffrast.torch")
        sys.modules.update({"nvdiffrast": nv, "nvdiffrast.torch": nv.torch, "imageio": types.ModuleType("imageio"), "kaolin": types.ModuleType("kaolin")})
        render_pkg = types.ModuleType("render")
        render_pkg.__path__ = []
        sys.modules["render"] = render_pkg
        with refload.CudaToCpu():
            util = refload._exec_module("render.util", "render/util.py")
            sys.modules["render.util"] = util
            render_pkg.util = util
            for name in ("mesh", "render", "optixutils"):
                m = types.ModuleType("render." + name)
                sys.modules["render." + name] = m
                setattr(render_pkg, name, m)
            reg = refload._exec_module("render.regularizer", "render/regularizer.py", {"__package__": "render"})
            sys.modules["render.regularizer"] = reg
            render_pkg.regularizer = reg
            geo_pkg = types.ModuleType("geometry")
            geo_pkg.__path__ = []
            sys.modules["geometry"] = geo_pkg
            emb = refload._exec_module("geometry.embedding", "geometry/embedding.py", {"__package__": "geometry"})
            sys.modules["geometry.embedding"] = emb
            mlp = refload._exec_module("geometry.mlp", "geometry/mlp.py", {"__package__": "geometry"})
            sys.modules["geometry.mlp"] = mlp
            gt = types.ModuleType("geometry.gshell_tets")
            gt.GShell_Tets = object
            sys.modules["geometry.gshell_tets"] = gt
            geo = refload._exec_module("geometry.gshell_tets_geometry", "geometry/gshell_tets_geometry.py", {"__package__": "geometry"})
        return geo.GShellTetsGeometry, mlp.MLP, geo
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


class _RefKernelLoss(torch.autograd.Function):
    """ru.image_loss (renderutils/ops.py:479-501) on the reference's own loss.cu compiled for the host."""

    @staticmethod
    def forward(ctx, img, target):
        ctx.save_for_backward(img, target)
        v, _ = refnative.image_loss_fwd(img.detach().numpy(), target.detach().numpy(), 'l1', 'log_srgb')
        return torch.tensor(float(v))

    @staticmethod
    def backward(ctx, g):
        img, target = ctx.saved_tensors
        gi, gt = refnative.image_loss_bwd(img.detach().numpy(), target.detach().numpy(), 'l1', 'log_srgb', 1.0)
        return torch.tensor(gi) * g, torch.tensor(gt) * g


def _case(seed, B=2, H=24, W=20, N=400, T=300, Vw=120, Vb=60, n_pts=500):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)
    alpha = (r(B, H, W, 1) > 0.4).float()

    def with_alpha(x):
        return torch.cat((x, alpha), -1)
    leaf = {
        'shaded': torch.cat((r(B, H, W, 3) * 1.5 - 0.1, r(B, H, W, 1)), -1),
        'msdf_image': r(B, H, W, 1) * 2 - 1,
        'diffuse_light': with_alpha(r(B, H, W, 3) * 2), 'specular_light': with_alpha(r(B, H, W, 3) * 0.5),
        'kd_grad': with_alpha(r(B, H, W, 3) * 0.1), 'ks_grad': with_alpha(r(B, H, W, 3) * 0.1), 'normal_grad': with_alpha(r(B, H, W, 3) * 0.1),
        'kd': with_alpha(r(B, H, W, 3)),
        'msdf': r(Vw + Vb) * 0.02 - 0.005, 'msdf_boundary': r(Vb) * 0.004 - 0.001, 'sdf': (r(N, 1) - 0.5) * 0.3,
    }
    mask = (r(B, H, W, 1) > 0.5).float()
    target = {'img': torch.cat((r(B, H, W, 3) * 1.2, mask), -1)}
    const = {'faces': torch.randint(0, Vw + Vb, (T, 3), generator=g), 'visible': torch.unique(torch.randint(0, T, (T // 3,), generator=g)),
             'all_edges': torch.randint(0, N, (900, 2), generator=g), 'pts': r(n_pts, 3) - 0.5, 'n_wt': Vw}
    return leaf, target, const


@pytest.mark.parametrize("seed,iteration,close,chroma", [(0, 0, 3e-6, 0.0), (1, 700, 3e-6, 0.1), (2, 2500, 0.0, 0.0), (3, 4000, -2e-6, 0.05)])
def test_restated_tick_equals_the_reference_tick(seed, iteration, close, chroma):
    RefGeom, RefMLP, geo_mod = _load_reference_tick()
    if not refnative.available("ref_renderutils"):
        pytest.skip("oracle/_ref/ref_renderutils.so not built")
    FLAGS = types.SimpleNamespace(iter=5000, use_img_2nd_layer=False, use_depth=False, use_sdf_mlp=True, use_eikonal=True, eikonal_scale=None,
                                  use_mesh_msdf_reg=True, msdf_reg_open_scale=1e-6, msdf_reg_close_scale=close, sdf_regularizer=0.2,
                                  lambda_diffuse=0.15, lambda_specular=0.0025, lambda_kd=0.1, lambda_ks=0.05, lambda_nrm=0.025, lambda_chroma=chroma)
    leaf0, target, const = _case(seed)
    torch.manual_seed(seed)
    with refload.CudaToCpu():
        net = RefMLP(n_freq=6, d_hidden=64, n_hidden=3, skip_in=[2])
    grid_res = 128
    results = []
    for which in ("reference", "oracle"):
        leaf = {k: v.clone().requires_grad_(True) for k, v in leaf0.items()}
        net.zero_grad()
        buffers = {k: leaf[k] for k in ('shaded', 'msdf_image', 'diffuse_light', 'specular_light', 'kd_grad', 'ks_grad', 'normal_grad', 'kd')}
        buffers['visible_triangles'] = const['visible']
        if which == "reference":
            imesh = types.SimpleNamespace(t_pos_idx=const['faces'])
            d = {'buffers': buffers, 'imesh': imesh, 'msdf': leaf['msdf'], 'msdf_boundary': leaf['msdf_boundary'], 'n_verts_watertight': const['n_wt'],
                 'sdf': leaf['sdf'], 'sampled_pts': const['pts']}
            fake = types.SimpleNamespace(FLAGS=FLAGS, grid_res=grid_res, sdf_net=net, all_edges=const['all_edges'], render=lambda *a, **k: d)
            with refload.CudaToCpu():
                img, depth, reg = RefGeom.tick(fake, None, target, None, None, _RefKernelLoss.apply, iteration, None)
        else:
            d = {'buffers': buffers, 'imesh_faces': const['faces'], 'msdf': leaf['msdf'], 'msdf_boundary': leaf['msdf_boundary'],
                 'n_verts_watertight': const['n_wt'], 'sdf': leaf['sdf'], 'sampled_pts': const['pts']}
            img, depth, reg, _ = tick_oracle.tick(FLAGS, grid_res, net, const['all_edges'], d, target, iteration)
        (img + depth + reg).backward()
        grads = {k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
        grads.update({f"net.{n}": (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in net.named_parameters()})
        results.append((float(img), float(depth), float(reg), grads))
    (i0, d0, r0, g0), (i1, d1, r1, g1) = results
    assert abs(i0 - i1) <= 1e-6 * abs(i0) and d0 == d1 == 0.0 and abs(r0 - r1) <= 1e-6 * abs(r0), (i0, i1, r0, r1)
    assert r0 != 0 and i0 != 0
    for k in g0:
        a, b = g0[k], g1[k]
        last_bias = k.startswith("net.") and k.endswith(".bias") and a.numel() == 1      # the eikonal term does not see the output bias
        if (k != 'kd' or chroma != 0) and (k != 'msdf_boundary' or close != 0) and not last_bias:
            assert float(a.abs().max()) > 0, f"{k}: the reference gradient is zero -- the case does not exercise this input"
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7 * float(a.abs().max())), (k, float((a - b).abs().max()), float(a.abs().max()))

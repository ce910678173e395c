"""Mint golden vectors for the shading stage from the REAL reference python code (build container only).

    python -m oracle.make_golden_shade

  * render/renderutils/bsdf.py : bsdf_lambert, bsdf_pbr_specular (the python twins of c_src/bsdf.h)
  * render/optixutils/tests/filter_test.py : class BilateralDenoiser (python reference of denoising.cu)
Writes tests/golden/shade_bsdf.npz and tests/golden/shade_bilateral.npz."""
import ast
import math
import os

import numpy as np
import torch

from oracle import refload

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    g = torch.Generator().manual_seed(0)
    with refload.CudaToCpu():
        bsdf = refload.load_simple("render/renderutils/bsdf.py", "ref_ru_bsdf")
        n = 512
        nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
        wo = torch.nn.functional.normalize(nrm + 0.8 * torch.randn(n, 3, generator=g), dim=-1)
        wi = torch.nn.functional.normalize(nrm + 0.8 * torch.randn(n, 3, generator=g), dim=-1)
        col = torch.rand(n, 3, generator=g)
        alpha = torch.rand(n, 1, generator=g) ** 2
        w = torch.randn(n, 3, generator=g)
        leaves = [t.clone().requires_grad_(True) for t in (col, nrm, wo, wi, alpha)]
        spec = bsdf.bsdf_pbr_specular(*leaves, min_roughness=0.08)
        lam = bsdf.bsdf_lambert(leaves[1], leaves[3])
        ((spec * w).sum() + (lam * w[:, :1]).sum()).backward()
        np.savez_compressed(os.path.join(OUT, "shade_bsdf.npz"), col=col.numpy(), nrm=nrm.numpy(), wo=wo.numpy(), wi=wi.numpy(), alpha=alpha.numpy(),
                            w=w.numpy(), spec=spec.detach().numpy(), lambert=lam.detach().numpy(),
                            **{f"g_{k}": t.grad.numpy() for k, t in zip(("col", "nrm", "wo", "wi", "alpha"), leaves)})

        # bilateral: exec the reference's python class (filter_test.py:31-74)
        src = open(os.path.join(refload.REF_ROOT, "render/optixutils/tests/filter_test.py")).read()
        tree = ast.parse(src)
        parts = [ast.get_source_segment(src, nd) for nd in tree.body if getattr(nd, "name", None) in ("length", "safe_normalize", "dot", "BilateralDenoiser")]
        ns = {"torch": torch, "np": np, "math": math}
        exec("\n\n".join(parts), ns)
        B, H, W = 1, 14, 11
        img = torch.rand(B, H, W, 11, generator=g)
        img[..., 3:6] = ns["safe_normalize"](img[..., 3:6] - 0.3)
        img[..., 9] = img[..., 9] * 0.2 + 0.5            # depth
        img[..., 10] = img[..., 10] * 0.02 + 0.001       # |dz|
        wgt = torch.randn(B, H, W, 3, generator=g)
        rec = {"in": img.numpy(), "w": wgt.numpy()}
        for sigma in (0.4, 2.0):
            x = img.clone().requires_grad_(True)
            out = ns["BilateralDenoiser"](sigma=sigma).forward(x)
            (out * wgt).sum().backward()
            rec[f"out_{sigma}"] = out.detach().numpy()
            rec[f"g_col_{sigma}"] = x.grad[..., 0:3].numpy()
        np.savez_compressed(os.path.join(OUT, "shade_bilateral.npz"), **rec)
    print("wrote shade goldens")


if __name__ == "__main__":
    main()

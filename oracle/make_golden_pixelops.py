"""Mint golden vectors for the per-pixel ops from the REAL reference code (run in the build container only).

    python -m oracle.make_golden_pixelops

Imports the reference's own Python twins (render/renderutils/bsdf.py, loss.py), and exec's the source of
render/mesh.py:auto_normals and render/light.py:EnvironmentLight.update_pdf with minimal stubs, on CPU.
Writes tests/golden/pixelops_*.npz (inputs + outputs + input gradients of a fixed weighted sum)."""
import ast
import os
import types

import numpy as np
import torch

from oracle import refload

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _function_source(relpath, names):
    src = open(os.path.join(refload.REF_ROOT, relpath)).read()
    tree = ast.parse(src)
    parts = [ast.get_source_segment(src, n) for n in tree.body if getattr(n, "name", None) in names]
    assert len(parts) == len(names), (relpath, names)
    return "\n\n".join(parts)


def _util_ns():
    util = refload._util_stub()

    def pixel_grid(width, height, center_x=0.5, center_y=0.5):   # render/util.py:61-65 (device kwarg removed)
        y, x = torch.meshgrid((torch.arange(0, height, dtype=torch.float32) + center_y) / height,
                              (torch.arange(0, width, dtype=torch.float32) + center_x) / width, indexing='ij')
        return torch.stack((x, y), dim=-1)
    util.pixel_grid = pixel_grid
    return util


def main():
    g = torch.Generator().manual_seed(0)
    with refload.CudaToCpu():
        bsdf = refload.load_simple("render/renderutils/bsdf.py", "ref_ru_bsdf")
        loss = refload.load_simple("render/renderutils/loss.py", "ref_ru_loss")

        # ---- prepare_shading_normal
        B, H, W = 2, 6, 5
        ins = {k: torch.randn(B, H, W, 3, generator=g) for k in ("pos", "smooth_nrm", "smooth_tng", "geom_nrm", "perturbed_nrm")}
        ins["geom_nrm"] = torch.nn.functional.normalize(ins["geom_nrm"], dim=-1)
        ins["view_pos"] = torch.randn(B, 1, 1, 3, generator=g) * 3
        wgt = torch.randn(B, H, W, 3, generator=g)
        rec = {"w": wgt.numpy()}
        for tag, pn in (("nopert", torch.tensor([0., 0., 1.])[None, None, None, :]), ("pert", ins["perturbed_nrm"])):
            for two_sided in (True, False):
                leaves = {k: v.clone().requires_grad_(True) for k, v in ins.items() if k != "perturbed_nrm"}
                pnl = pn.clone().requires_grad_(True)
                out = bsdf.bsdf_prepare_shading_normal(leaves["pos"], leaves["view_pos"], pnl, leaves["smooth_nrm"], leaves["smooth_tng"],
                                                       leaves["geom_nrm"], two_sided, True)
                (out * wgt).sum().backward()
                key = f"{tag}_{int(two_sided)}"
                rec[f"out_{key}"] = out.detach().numpy()
                for k, v in leaves.items():
                    rec[f"g_{k}_{key}"] = v.grad.numpy()
                if tag == "pert":
                    rec[f"g_perturbed_nrm_{key}"] = pnl.grad.numpy()
        for k, v in ins.items():
            rec[f"in_{k}"] = v.numpy()
        np.savez_compressed(os.path.join(OUT, "pixelops_shading_normal.npz"), **rec)

        # ---- image_loss (python twin)
        img = torch.rand(2, 7, 9, 3, generator=g) * 1.5 - 0.1
        tgt = torch.rand(2, 7, 9, 3, generator=g) * 1.5
        rec = {"in_img": img.numpy(), "in_target": tgt.numpy()}
        for l in ("l1", "mse", "smape", "relmse"):
            for tm in ("none", "log_srgb"):
                a = img.clone().requires_grad_(True)
                v = loss.image_loss_fn(a, tgt, l, tm)
                v.backward()
                rec[f"out_{l}_{tm}"] = v.detach().numpy()
                rec[f"g_img_{l}_{tm}"] = a.grad.numpy()
        np.savez_compressed(os.path.join(OUT, "pixelops_image_loss.npz"), **rec)

        # ---- auto_normals (render/mesh.py:212-237) with stub Mesh / util
        ns = {"torch": torch, "util": _util_ns()}

        class Mesh:
            def __init__(self, v_pos=None, t_pos_idx=None, v_nrm=None, t_nrm_idx=None, base=None, **kw):
                self.v_pos, self.t_pos_idx, self.v_nrm, self.t_nrm_idx = v_pos, t_pos_idx, v_nrm, t_nrm_idx
        ns["Mesh"] = Mesh
        exec(_function_source("render/mesh.py", ["auto_normals"]), ns)
        from oracle import scenes
        verts, tri = scenes.grid_sheet(5, seed=3)
        verts = np.concatenate([verts, np.zeros((2, 3), np.float32)])      # two unreferenced vertices -> (0,0,1)
        v = torch.tensor(verts).requires_grad_(True)
        wv = torch.randn(v.shape, generator=g)
        out = ns["auto_normals"](Mesh(v, torch.tensor(tri).long())).v_nrm
        (out * wv).sum().backward()
        np.savez_compressed(os.path.join(OUT, "pixelops_auto_normals.npz"), in_verts=verts, in_tri=tri, w=wv.numpy(), out=out.detach().numpy(),
                            g_verts=v.grad.numpy())

        # ---- EnvironmentLight.update_pdf (render/light.py:21-59)
        ns = {"torch": torch, "np": np, "util": _util_ns()}
        exec(_function_source("render/light.py", ["EnvironmentLight"]), ns)
        base = torch.rand(16, 32, 3, generator=g) * 0.5 + 0.25
        lgt = ns["EnvironmentLight"](base)
        np.savez_compressed(os.path.join(OUT, "pixelops_light_pdf.npz"), in_base=base.numpy(), pdf=lgt._pdf.numpy(), rows=lgt.rows.numpy(),
                            cols=lgt.cols.numpy())
    print("wrote goldens to", OUT)


if __name__ == "__main__":
    main()
